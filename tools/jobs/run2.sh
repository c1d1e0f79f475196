set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02/gputest_b.log 2>&1 || (tail -60 gpurun_out/r02/gputest_b.log; exit 1)
tail -25 gpurun_out/r02/gputest_b.log
python bench.py > gpurun_out/r02/bench_b.json 2> gpurun_out/r02/bench_b.err
cat gpurun_out/r02/bench_b.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu-baseline --no-full-solve --no-fft > gpurun_out/r02/bench_b_torchrun.json 2> gpurun_out/r02/bench_b_torchrun.err
cat gpurun_out/r02/bench_b_torchrun.json
echo done
