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
        # prepare(): the rank's slice is wrapped ahead of the first evaluation (the solvers' set-up makes
        # the list resident — and cut into patches — there), exactly once, and is what the evaluation uses
        from ipde_amd.sharding import target_slice
        wrapped = []
        ev = make_pnai_evaluator(la, c, trg, lambda x, y: wrapped.append(Points(x, y)) or wrapped[-1])
        mine = ev.prepare()
        sl = target_slice(trg.N, rank, world)
        ok = ok and len(wrapped) == 1 and mine is wrapped[0] and np.array_equal(mine.x, trg.x[sl])
        ok = ok and np.allclose(ev(sigma).numpy(), ref, rtol=0, atol=1e-13) and len(wrapped) == 1
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


def _worker_new_paths(rank, world, port, q):
    """ResultGather (preallocated all_gather_into_tensor, one collective per tuple),
    make_sharded_evaluator with its small-sum bypass, exchange_owned, and the solvers'
    boundary-ownership helpers (_owned / _run_owned) on stub helpers."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from ipde_amd import sharding
        from ipde_amd.sharding import ResultGather, make_sharded_evaluator, exchange_owned, owner_of
        from oracle import layer_potentials as olp
        from util import Curve, Points
        ok = True
        # ragged (1001) and even (world * 50) lengths, 1 and 3 components, buffers reused
        for nt in (1001, world * 50, 2, 0):
            for ncomp in (1, 3):
                g = ResultGather(nt, ncomp)
                sl = target_slice(nt, rank, world)
                for rep in range(2):
                    full = [torch.arange(nt, dtype=torch.float64) * (c + 1) + rep for c in range(ncomp)]
                    got = g(tuple(f[sl] for f in full))
                    ok = ok and all(torch.equal(a, b) for a, b in zip(got, full))
                    first = got if rep == 0 else first
                # results of the first call must not alias the reused receive buffer
                ok = ok and all(torch.equal(a, torch.arange(nt, dtype=torch.float64) * (c + 1))
                                for c, a in enumerate(first))
        # sharded evaluator: sharded path and the min_pairs bypass give the same numbers
        c = Curve(64, a=0.2, f=5)
        rng = np.random.default_rng(0)
        trg = Points(rng.uniform(-0.5, 0.5, 1001), rng.uniform(-0.5, 0.5, 1001))
        sigma = rng.standard_normal(c.N)
        calls = []

        def la(src, t, d):
            calls.append(t.N)
            return torch.as_tensor(olp.laplace_layer_apply(src.x, src.y, t.x, t.y, charge=d,
                                                           weights=src.weights))
        ref = olp.laplace_layer_apply(c.x, c.y, trg.x, trg.y, charge=sigma, weights=c.weights)
        ev = make_sharded_evaluator(la, trg, lambda x, y: Points(x, y))
        ok = ok and np.allclose(ev(c, sigma).numpy(), ref, rtol=0, atol=1e-13) and calls[-1] < 1001
        ev = make_sharded_evaluator(la, trg, lambda x, y: Points(x, y), min_pairs=1e9)
        ok = ok and np.allclose(ev(c, sigma).numpy(), ref, rtol=0, atol=1e-13) and calls[-1] == 1001
        # exchange_owned: ragged per-boundary arrays + a ride-along vector
        nb = 5
        shapes = [(3 + i, 2) for i in range(nb)]
        truth = [np.random.default_rng(i).standard_normal(s) for i, s in enumerate(shapes)]
        vals = [t if owner_of(i, world) == rank else None for i, t in enumerate(truth)]
        extra = [float(10 + i) if owner_of(i, world) == rank else 0.0 for i in range(nb)]
        got, ex = exchange_owned(vals, shapes, extra=extra)
        ok = ok and all(np.array_equal(a, b) for a, b in zip(got, truth))
        ok = ok and np.array_equal(ex, 10.0 + np.arange(nb))
        # the solvers' ownership helpers on stub helpers
        from ipde_amd.solvers.multi_boundary.scalar import _owned, _run_owned

        class Helper:
            def __init__(self, i):
                self.i = i
                self.started = 0

            def start_call(self, a, b, scale=1.0):
                self.started += 1
                return [("req", self.i, a + b, scale)]

            def finish_call(self, res):
                return np.full(4, res)

        class Solver:
            DISTRIBUTE_BOUNDARIES = True
            helpers = [Helper(i) for i in range(nb)]

            def _concurrent_helpers(self):
                return False
        s = Solver()
        mine, distributed = _owned(s)
        ok = ok and distributed and mine == [i for i in range(nb) if i % world == rank]
        solve_many = lambda reqs: [r[1] * 100 + r[2] * r[3] for r in reqs]
        out = _run_owned(s, mine, 'start_call', 'finish_call', [(i, 1) for i in range(nb)], solve_many,
                         scale=2.0)
        ok = ok and all((out[i] is None) == (i not in mine) for i in range(nb))
        ok = ok and all(h.started == (1 if i in mine else 0) for i, h in enumerate(s.helpers))
        full = exchange_owned(out, [(4,)] * nb)
        ok = ok and all(np.array_equal(full[i], np.full(4, i * 100 + (i + 1) * 2.0)) for i in range(nb))
        one = Solver()
        one.helpers = one.helpers[:1]
        ok = ok and _owned(one) == ([0], False)       # a single boundary stays replicated
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_gather_exchange_and_ownership_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_new_paths, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)], results


def _worker_sharded_through(rank, world, port, q):
    """The sum onto grid_pnai with its grid part LEFT sharded (sharding.gather_tail, the evaluators'
    .sharded form): local slice + gathered tail reproduce the full sum, scalar and tuple results,
    tails shorter and longer than the last rank's slice; the exchange is n_tail numbers, not the list;
    exchange_owned on tensors returns tensors (the device-resident flow's form)."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import layer_potentials as olp
        from util import Curve, Points
        from ipde_amd import sharding
        from ipde_amd.sharding import make_pnai_evaluator, target_slice, exchange_owned, owner_of
        c = Curve(64, a=0.2, f=5)
        rng = np.random.default_rng(0)
        nt = 1001
        trg = Points(rng.uniform(-0.5, 0.5, nt), rng.uniform(-0.5, 0.5, nt))
        sigma = rng.standard_normal(c.N)
        f = rng.standard_normal((2, c.N))
        wrap = lambda x, y: Points(x, y)

        def la(src, t, d):
            return torch.as_tensor(olp.laplace_layer_apply(src.x, src.y, t.x, t.y, charge=d,
                                                           weights=src.weights))

        def la3(src, t, d):
            return tuple(torch.as_tensor(a) for a in
                         olp.stokes_layer_apply(src.x, src.y, t.x, t.y, force=d, weights=src.weights))
        ref = olp.laplace_layer_apply(c.x, c.y, trg.x, trg.y, charge=sigma, weights=c.weights)
        ref3 = olp.stokes_layer_apply(c.x, c.y, trg.x, trg.y, force=f, weights=c.weights)
        sl = target_slice(nt, rank, world)
        ok = True
        for n_tail in (0, 37, 400, 700):       # 700 > a slice at world 2 and 3: the tail spans ranks
            sharding.reset_stats()
            so = make_pnai_evaluator(la, c, trg, wrap).sharded(sigma, n_tail)
            ok = ok and (so.slice.start, so.slice.stop, so.nt) == (sl.start, sl.stop, nt)
            ok = ok and np.allclose(so.local.numpy(), ref[sl], rtol=0, atol=1e-13)
            ok = ok and so.tail.shape[0] == n_tail and np.allclose(so.tail.numpy(), ref[nt - n_tail:], rtol=0, atol=1e-13)
            ok = ok and sharding.STATS == {"collectives": 1, "bytes": 8 * n_tail}
            sharding.reset_stats()
            so3 = make_pnai_evaluator(la3, c, trg, wrap).sharded(f, n_tail)
            ok = ok and all(np.allclose(l.numpy(), r[sl], rtol=0, atol=1e-13) for l, r in zip(so3.local, ref3))
            ok = ok and all(np.allclose(t.numpy(), r[nt - n_tail:], rtol=0, atol=1e-13) for t, r in zip(so3.tail, ref3))
            ok = ok and sharding.STATS == {"collectives": 1, "bytes": 3 * 8 * n_tail}   # one collective per tuple
        # the gathered form of the same evaluator moves the whole (padded) list
        sharding.reset_stats()
        make_pnai_evaluator(la, c, trg, wrap)(sigma)
        ok = ok and sharding.STATS["bytes"] >= 8 * nt
        # exchange_owned with tensors in, tensors out (every position written by its owner only)
        nb = 4
        shapes = [(3, 5), (7,), (2, 2), (6,)]
        vals = [torch.full(s, float(i + 1), dtype=torch.float64) if owner_of(i, world) == rank else None
                for i, s in enumerate(shapes)]
        its = [float(10 + i) if owner_of(i, world) == rank else 0.0 for i in range(nb)]
        full, its_all = exchange_owned(vals, shapes, extra=its)
        ok = ok and all(isinstance(v, torch.Tensor) and tuple(v.shape) == s and bool((v == i + 1).all())
                        for i, (v, s) in enumerate(zip(full, shapes)))
        ok = ok and np.array_equal(its_all, [10.0, 11.0, 12.0, 13.0])
        # world > number of boundaries: a rank that owns nothing passes only None; with the kind of the
        # result STATED (as_tensors=True, the solvers' device flow) it gets tensors like everybody else —
        # inferred, it took the numpy path and gridops.concat on its list raised (round-3 advisor)
        from ipde_amd import gridops
        few = [(5,), (3,)][:max(1, world - 1)]
        vals = [torch.full(s, float(i + 1), dtype=torch.float64) if owner_of(i, world) == rank else None
                for i, s in enumerate(few)]
        its = [float(20 + i) if owner_of(i, world) == rank else 0.0 for i in range(len(few))]
        if world > len(few):
            ok = ok and (rank < len(few) or all(v is None for v in vals))
        full, its_all = exchange_owned(vals, few, device="cpu", extra=its, as_tensors=True)
        ok = ok and all(isinstance(v, torch.Tensor) for v in full)
        cat = gridops.concat(list(full))
        ok = ok and cat.tolist() == [float(i + 1) for i, s in enumerate(few) for _ in range(s[0])]
        ok = ok and np.array_equal(its_all, [20.0 + i for i in range(len(few))])
        host = exchange_owned([None if v is None else v.numpy() for v in vals], few, as_tensors=False)
        ok = ok and all(isinstance(v, np.ndarray) for v in host) and np.array_equal(np.concatenate(host), cat.numpy())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_through_evaluation_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded_through, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)], results


def test_sharded_form_in_a_single_process():
    from oracle import layer_potentials as olp
    from util import Curve, Points
    from ipde_amd.sharding import make_pnai_evaluator
    c = Curve(32, a=0.2, f=5)
    rng = np.random.default_rng(1)
    trg = Points(rng.uniform(-0.5, 0.5, 200), rng.uniform(-0.5, 0.5, 200))
    sigma = rng.standard_normal(c.N)
    la = lambda src, t, d: torch.as_tensor(olp.laplace_layer_apply(src.x, src.y, t.x, t.y, charge=d,
                                                                   weights=src.weights))
    so = make_pnai_evaluator(la, c, trg, lambda x, y: Points(x, y)).sharded(sigma, 30)
    ref = olp.laplace_layer_apply(c.x, c.y, trg.x, trg.y, charge=sigma, weights=c.weights)
    assert (so.slice.start, so.slice.stop) == (0, 200)
    assert np.allclose(so.local.numpy(), ref, rtol=0, atol=1e-13) and np.array_equal(so.tail.numpy(), so.local.numpy()[170:])
