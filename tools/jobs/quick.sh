set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/hostprof_ewald_setup.py 2>&1 | grep -v amdgpu.ids | cut -c1-180 > gpurun_out/ewald_setup_prof.txt
head -50 gpurun_out/ewald_setup_prof.txt
