set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 5 --warmup 2 --rehearse-shared-gpu > gpurun_out/bench_rehearse2.json 2> gpurun_out/bench_rehearse2.err
python3 -c "
import json; d=json.load(open('gpurun_out/bench_rehearse2.json')); print(d['n_gpus'], d['ms_per_step'], d['parity_max_rel_err_vs_oracle'], d['config']['target_patches'], d['config']['n_targets_per_gpu'])"
timeout -k 10 900 python3 -m pytest tests/test_sharding.py tests/test_spectral_gpu.py tests/test_ewald_gpu.py -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -5
