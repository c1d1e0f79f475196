set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIPDE_LU_STAMPS tools/lu_panel_probe.hip -o /tmp/lu_panel_probe 2>/dev/null
timeout -k 10 60 /tmp/lu_panel_probe 4096 0 > gpurun_out/r02/lu_panel_probe_0.txt
timeout -k 10 60 /tmp/lu_panel_probe 4096 60 > gpurun_out/r02/lu_panel_probe_60.txt
cat gpurun_out/r02/lu_panel_probe_0.txt gpurun_out/r02/lu_panel_probe_60.txt
