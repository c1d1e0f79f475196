set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r02/gputest_e.log 2>&1 || true
tail -12 gpurun_out/r02/gputest_e.log
python tools/run_sharded_solve.py --problem modhelm --nb 8192 --M 20 --k 10 --ng 4096 > gpurun_out/r02/config3_single.json 2> gpurun_out/r02/config3_single.err
cat gpurun_out/r02/config3_single.json
echo done
