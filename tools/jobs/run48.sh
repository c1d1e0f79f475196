set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 300 python3 tools/hostprof_setup.py > gpurun_out/r02/hostprof_setup3.txt 2>&1
head -50 gpurun_out/r02/hostprof_setup3.txt | cut -c1-150
