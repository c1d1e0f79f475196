#!/usr/bin/env python3
"""One rank of a target-sharded solve: every rank holds the replicated solver state and
evaluates its slice of grid_pnai; the slices are all-gathered (ipde_amd/sharding.py).

On an 8-GPU node:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/run_sharded_solve.py --problem poisson --nb 4096 --M 20
On a one-GPU box (rehearsal: all ranks share cuda:0, collectives over gloo):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
        tools/run_sharded_solve.py --backend gloo --share-gpu --problem poisson --nb 800
Prints one JSON line from rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", choices=["poisson", "modhelm", "multi_modhelm", "stokes"], default="poisson")
    ap.add_argument("--nb", type=int, default=800)
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--k", type=float, default=10.0)
    ap.add_argument("--ng", type=int, default=None, help="force an ng x ng grid (scalar problems)")
    ap.add_argument("--grid-backend", default=None, help="'ewald' for the split grid evaluator")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0 (one-GPU rehearsal)")
    ap.add_argument("--sharded-result", action="store_true",
                    help="scalar problems: the answer stays sharded through the solve and the correction "
                         "(only the interface values of the sum onto grid_pnai are exchanged)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        kw = {} if a.backend == "gloo" else {"device_id": torch.device("cuda", local)}
        dist.init_process_group(a.backend, **kw)
    from ipde_amd.device import get_context
    get_context(local)
    Ns = None if a.ng is None else [a.ng, a.ng]
    t0 = time.perf_counter()
    if a.problem == "poisson":
        import interior_poisson
        err, scale, solver, ue, T = interior_poisson.run(nb=a.nb, M=a.M, Ns=Ns, grid_backend=a.grid_backend,
                                                         sharded_result=a.sharded_result)
        res = {"error": err / scale}
    elif a.problem == "modhelm":
        import interior_modified_helmholtz as imh
        err, scale, solver, ue, T = imh.run(nb=a.nb, M=a.M, helmholtz_k=a.k, Ns=Ns, grid_backend=a.grid_backend,
                                            sharded_result=a.sharded_result)
        res = {"error": err / scale}
    elif a.problem == "multi_modhelm":
        # three boundaries (one outer, two holes): with world > 3 some ranks own no boundary at all
        import multi_modified_helmholtz as mmh
        err, scale, T = mmh.run(nb=a.nb, M=a.M, helmholtz_k=a.k)
        res = {"error": err / scale}
    else:
        import multi_stokes
        ue, ve, pe, scale, T = multi_stokes.run(nb=a.nb, M=a.M)
        res = {"error": max(ue, ve) / scale, "p_error": pe}
    if a.problem in ("poisson", "modhelm"):
        # a second, warm solve of the same problem
        import numpy as np
        from ipde_amd.embedded_function import EmbeddedFunction
        f = EmbeddedFunction(solver.ebdyc)
        f.define_via_function(lambda x, y: np.sin(x) * np.cos(y))
        from ipde_amd import sharding
        torch.cuda.synchronize()
        sharding.reset_stats()
        t1 = time.perf_counter()
        full = solver(f, tol=1e-12, maxiter=100, restart=20)
        torch.cuda.synchronize()
        res["warm_inhomogeneous_solve_s"] = time.perf_counter() - t1
        # logical payload this rank received in the solve's collectives (ipde_amd/sharding.py STATS)
        res["collectives_per_solve"] = dict(sharding.STATS)
        if a.sharded_result:
            sharding.reset_stats()
            t1 = time.perf_counter()
            part = solver(f, tol=1e-12, maxiter=100, restart=20, sharded_result=True)
            torch.cuda.synchronize()
            res["warm_inhomogeneous_solve_sharded_result_s"] = time.perf_counter() - t1
            res["collectives_per_solve_sharded_result"] = dict(sharding.STATS)
            own = part.owned
            # this rank's entries are BITWISE those of the replicated answer, and the masks partition
            same = bool(np.array_equal(np.asarray(part)[own], np.asarray(full)[own]))
            counts = sharding.gather_owned(np.ones(own.shape[0]), own)
            whole = sharding.gather_owned(np.asarray(part), own)
            res["sharded_result_bitwise_equal_on_owned"] = same
            res["owned_masks_partition_the_answer"] = bool(np.all(counts == 1.0))
            res["gathered_sharded_result_equals_replicated"] = bool(np.array_equal(whole, np.asarray(full)))
    res.update({"problem": a.problem, "world": world, "nb": a.nb, "M": a.M,
                "grid_backend": a.grid_backend, "wall_s": time.perf_counter() - t0,
                "timings": {k: v for k, v in T.items() if isinstance(v, (int, float, list))}})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
