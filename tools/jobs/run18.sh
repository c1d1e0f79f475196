set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02/gputest_g.log 2>&1 || true
tail -6 gpurun_out/r02/gputest_g.log
python tools/profile_solve.py | tail -1
echo done
