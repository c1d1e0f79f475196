set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIPDE_LU_STAMPS tools/lu_persist_probe.hip -o /tmp/lu_probe 2>/dev/null
timeout -k 10 60 /tmp/lu_probe 6400 1 > gpurun_out/r02/lu_probe_1.txt
timeout -k 10 60 /tmp/lu_probe 4096 2 > gpurun_out/r02/lu_probe_2.txt
head -30 gpurun_out/r02/lu_probe_1.txt
