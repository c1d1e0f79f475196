set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q --durations=10 --deselect tests/test_configs_gpu.py::test_config4_multi_stokes_three_bodies_4096_grid > gpurun_out/r02/gputest_c.log 2>&1 || true
tail -30 gpurun_out/r02/gputest_c.log
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/segprobe tools/segment_bw_probe.hip && /tmp/segprobe > gpurun_out/r02/segprobe.txt 2>&1
cat gpurun_out/r02/segprobe.txt
python tools/diag_stokes.py > gpurun_out/r02/diag_stokes.log 2>&1 || true
grep "^{" gpurun_out/r02/diag_stokes.log || tail -5 gpurun_out/r02/diag_stokes.log
python bench.py > gpurun_out/r02/bench_b.json 2> gpurun_out/r02/bench_b.err
cat gpurun_out/r02/bench_b.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu-baseline --no-full-solve --no-fft > gpurun_out/r02/bench_b_torchrun.json 2> gpurun_out/r02/bench_b_torchrun.err
cat gpurun_out/r02/bench_b_torchrun.json
echo done
