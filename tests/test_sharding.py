"""The N > 1 path on CPU: world_size-2 (and 3, ragged) gloo processes exercise the
target partition, the density all-gather and the result assembly of
ipde_amd/sharding.py.  The compute function is the numpy oracle here (the HIP
library cannot run without a GPU); on the GPU box bench.py runs the same host logic
with the HIP kernels over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ipde_amd.sharding import ShardedLayerApply, allgather_density, shard_sizes, target_slice  # noqa: E402


def test_target_slices_partition_everything():
    for nt in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                s = target_slice(nt, r, world)
                cover.extend(range(s.start, s.stop) if nt < 100 else [s.start, s.stop])
            if nt < 100:
                assert cover == list(range(nt))
            sizes = shard_sizes(nt, world)
            assert sum(sizes) == nt and max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import layer_potentials as olp
        from util import Curve
        c = Curve(101, a=0.2, f=5)          # 101 nodes: ragged density shards
        rng = np.random.default_rng(0)
        sigma = rng.standard_normal(c.N)
        tx = rng.uniform(-0.5, 0.5, 1003)   # ragged target shards
        ty = rng.uniform(-0.5, 0.5, 1003)

        def apply_fn(src, x, y, dens):
            return olp.laplace_layer_apply(src.x, src.y, x, y, charge=dens, weights=src.weights)

        sh = ShardedLayerApply(apply_fn, tx, ty)
        ds = target_slice(c.N, rank, world)
        full = sh(c, sigma[ds], c.N, gather_result=True)
        local = sh(c, sigma[ds], c.N)
        ref = olp.laplace_layer_apply(c.x, c.y, tx, ty, charge=sigma, weights=c.weights)
        ok = np.allclose(full, ref, rtol=0, atol=1e-13) and \
            np.allclose(local, ref[sh.slice], rtol=0, atol=1e-13)
        # Stokes-style tuple results and 2-row densities
        f = rng.standard_normal((2, c.N))
        dens = allgather_density(f[:, ds], c.N)
        ok = ok and np.array_equal(dens, f)

        def apply3(src, x, y, d):
            return olp.stokes_layer_apply(src.x, src.y, x, y, force=d, weights=src.weights)
        sh3 = ShardedLayerApply(apply3, tx, ty)
        u, v, p = sh3(c, f[:, ds], c.N, gather_result=True)
        ur, vr, pr = olp.stokes_layer_apply(c.x, c.y, tx, ty, force=f, weights=c.weights)
        ok = ok and np.allclose(u, ur, atol=1e-13) and np.allclose(p, pr, atol=1e-13)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok)))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_apply_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)], results


def _worker_pnai(rank, world, port, q):
    """make_pnai_evaluator: the solvers' sharded Grid_Evaluator (replicated density,
    sharded targets, all-gathered result), scalar and Stokes-style tuple results."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import layer_potentials as olp
        from util import Curve, Points
        from ipde_amd.sharding import make_pnai_evaluator
        c = Curve(64, a=0.2, f=5)
        rng = np.random.default_rng(0)
        trg = Points(rng.uniform(-0.5, 0.5, 1001), rng.uniform(-0.5, 0.5, 1001))
        sigma = rng.standard_normal(c.N)
        f = rng.standard_normal((2, c.N))
        wrap = lambda x, y: Points(x, y)

        def la(src, t, d):
            return torch.as_tensor(olp.laplace_layer_apply(src.x, src.y, t.x, t.y, charge=d,
                                                           weights=src.weights))

        def la3(src, t, d):
            return tuple(torch.as_tensor(a) for a in
                         olp.stokes_layer_apply(src.x, src.y, t.x, t.y, force=d, weights=src.weights))
        got = make_pnai_evaluator(la, c, trg, wrap)(sigma).numpy()
        ref = olp.laplace_layer_apply(c.x, c.y, trg.x, trg.y, charge=sigma, weights=c.weights)
        ok = got.shape == ref.shape and np.allclose(got, ref, rtol=0, atol=1e-13)
        got3 = make_pnai_evaluator(la3, c, trg, wrap)(f)
        ref3 = olp.stokes_layer_apply(c.x, c.y, trg.x, trg.y, force=f, weights=c.weights)
        ok = ok and all(np.allclose(g.numpy(), r, rtol=0, atol=1e-13) for g, r in zip(got3, ref3))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok)))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_pnai_evaluator_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pnai, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)], results
