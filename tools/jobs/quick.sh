set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --no-full-solve --no-fft --no-cpu-baseline > gpurun_out/b1.json 2> gpurun_out/b1.err
python3 -c "
import json; d=json.load(open('gpurun_out/b1.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r['list_kernel'], d['parity_max_rel_err_vs_oracle'])"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 5 --warmup 2 --rehearse-shared-gpu --no-cpu-baseline --no-full-solve --no-fft > gpurun_out/b2.json 2> gpurun_out/b2.err
python3 -c "
import json; d=json.load(open('gpurun_out/b2.json')); print(d['n_gpus'], d['ms_per_step'], d['parity_max_rel_err_vs_oracle'], d['roofline']['list_kernel'])"
