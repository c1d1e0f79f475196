# BASELINE configs[3] on one GPU, fresh processes, rocSOLVER / own LU alternating (the first process of a
# job also pays the box's cold file cache): gpurun -- 'bash tools/jobs/configs.sh'
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for v in 0 1 0 1; do
IPDE_OWN_LU=$v timeout -k 10 500 python3 tools/run_sharded_solve.py --problem modhelm --nb 8192 --M 20 --k 10 --ng 4096 > gpurun_out/r02/config3_ownlu$v.json 2> gpurun_out/r02/config3_final.err
python3 -c "
import json; d=json.load(open('gpurun_out/r02/config3_ownlu$v.json')); print('own_lu=$v', 'setup %.2f s' % d['timings']['setup_s'], 'first solve %.3f s' % d['timings']['inhomogeneous_solve_s'], 'warm %.1f ms' % (1e3*d['warm_inhomogeneous_solve_s']), 'wall %.2f' % d['wall_s'], 'err %.1e' % d['error'])"
done
# configs[4]: 3-body Stokes, 4096^2 grid (n_b = 2400)
timeout -k 10 500 python3 - > gpurun_out/r02/config4_final.json 2> gpurun_out/r02/config4_final.err <<'PY'
import json, sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "examples"))
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import multi_stokes
ue, ve, pe, scale, T = multi_stokes.run(nb=2400, M=14, ng=4096, warm=True)
print(json.dumps({"u_err": ue, "v_err": ve, "p_err": pe, "scale": scale, "timings": T}))
PY
cat gpurun_out/r02/config4_final.json
