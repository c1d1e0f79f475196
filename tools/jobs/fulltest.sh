# the whole GPU test-suite and the two warm-solve timings (one MI355X)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -6
timeout -k 10 300 python3 tools/profile_solve.py 2>&1 | tail -1
timeout -k 10 300 python3 tools/profile_stokes_solve.py 2>&1 | tail -1
