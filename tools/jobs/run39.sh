set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/stokes_trace3
IPDE_PROFILE_STOP_AFTER_WARM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/stokes_trace3 -- python3 tools/profile_stokes_solve.py > gpurun_out/r02/stokes_trace3.log 2>&1
grep "warm" gpurun_out/r02/stokes_trace3.log
