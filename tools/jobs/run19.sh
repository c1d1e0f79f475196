set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r02/gputest_h.log 2>&1 || (tail -40 gpurun_out/r02/gputest_h.log; exit 1)
tail -3 gpurun_out/r02/gputest_h.log
python tools/hostprof_setup.py > gpurun_out/r02/hostprof_setup2.txt 2>&1
grep "setup_s" gpurun_out/r02/hostprof_setup2.txt | cut -c1-300
grep -E "lu_factor|_factor|_get|result" gpurun_out/r02/hostprof_setup2.txt | head
python bench.py --no-cpu-baseline --steps 5 --warmup 1 --no-fft 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print(json.dumps(b['full_poisson_solve']))"
echo done
