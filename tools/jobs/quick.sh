# scratch job: edit for the experiment at hand (gpurun -- 'bash tools/jobs/quick.sh')
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_annular_gpu.py tests/test_solver_gpu.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -5
timeout -k 10 500 python3 tools/ab_gmres_lookahead.py 2>&1 | grep -v amdgpu.ids | tail -4
