"""ScalarSolver — the coupled grid + annuli driver, mirrors
ipde/solvers/multi_boundary/scalar.py:7-119 step for step:
  FFT grid solve -> values/gradient on the interfaces -> per-boundary annular solves
  and QFS densities -> Grid_Evaluator (dense GPU sum) -> correct() -> radial->grid.

The grid-sized arrays never leave HBM during a solve: the forcing's physical values go
up once (pinned), the masks / cut-offs / index sets are resident, every grid update is
a device gather/scatter, and only the physical values of the answer come back.  Small
per-boundary vectors (N or M*N doubles) travel as numpy, as in the reference.
"""
import os

import numpy as np
import torch

from ...derivatives import fd_x_4, fd_y_4
from ...embedded_function import EmbeddedFunction, BoundaryFunction
from ...interp import periodic_interp2d, periodic_interp2d_gradient, radial_to_grid
from ...qfs import call_many, u2s_many
from ...layer_potentials import DeviceTargets
from ...pybie2d_compat import BoundaryCollection
from ...device import prewarm_wait, prewarm_submit
from ... import hostio, gridops
from ...sharding import make_pnai_evaluator, exchange_owned, owner_of, is_distributed, _dist_state
from ...spectral import get_plan


def _finish_all(helpers, method, reqs, solve_many):
    """run the QFS solves requested by all helpers together and hand each helper its own"""
    flat = [r for req in reqs for r in req]
    res = solve_many(flat) if flat else []
    out, k = [], 0
    for helper, req in zip(helpers, reqs):
        out.append(getattr(helper, method)(*res[k:k + len(req)]))
        k += len(req)
    return out


def _concurrent_helpers(solver):
    """True (and a thread pool on solver._pool) if the annular solvers of the helpers can be
    driven concurrently: several of them, all distinct, each on a library context of its own"""
    if not solver.CONCURRENT_ANNULAR:
        return False
    ok = getattr(solver, '_concurrent', None)
    if ok is None:
        solvers = [h.annular_solver for h in solver.helpers]
        ctxs = [getattr(a, 'ctx', None) for a in solvers]
        ok = (len(solvers) > 1 and len({id(a) for a in solvers}) == len(solvers)
              and all(c is not None for c in ctxs) and len({id(c) for c in ctxs}) == len(ctxs)
              and all(c is not solver.plan.ctx for c in ctxs))
        if ok:
            from concurrent.futures import ThreadPoolExecutor
            solver._pool = ThreadPoolExecutor(len(solvers), thread_name_prefix="ipde-annular")
        solver._concurrent = ok
    return ok


def _owned(solver):
    """Indices of the boundaries whose annular / QFS work this rank does, and whether that
    work is distributed at all.  One process, or a single boundary: every rank does
    everything (replicated, no exchange).  Several boundaries under torch.distributed:
    boundary i belongs to rank i mod world (reference multi_boundary/scalar.py:98-101 loops
    over independent helpers); the densities and the annular solutions are exchanged with
    sharding.exchange_owned."""
    n = len(solver.helpers)
    _, rank, world = _dist_state()
    if world == 1 or n == 1 or not solver.DISTRIBUTE_BOUNDARIES:
        return list(range(n)), False
    return [i for i in range(n) if owner_of(i, world) == rank], True


def _run_owned(solver, mine, method_start, method_finish, args, solve_many, **kwargs):
    """start_* on the owned helpers (concurrently when they have contexts of their own),
    the QFS solves of all of them in one batched substitution, finish_* — results by
    boundary index (None for the boundaries of other ranks)"""
    helpers = [solver.helpers[i] for i in mine]
    margs = [(solver.helpers[i],) + tuple(args[i]) for i in mine]
    start = lambda a: getattr(a[0], method_start)(*a[1:], **kwargs)
    if method_start == 'start_call' and len(mine) > 1 and solver._concurrent_helpers():
        reqs = list(solver._pool.map(start, margs))
    else:
        reqs = [start(a) for a in margs]
    res = _finish_all(helpers, method_finish, reqs, solve_many)
    out = [None] * len(solver.helpers)
    for i, r in zip(mine, res):
        out[i] = r
    return out


class ScalarSolver(object):
    def __init__(self, ebdyc, solver_type='spectral', helpers=None, grid_backend=None):
        self.ebdyc = ebdyc
        self.solver_type = solver_type
        self.grid_backend = grid_backend
        if helpers is None:
            helpers = [None, ] * self.ebdyc.N
        self._start_pnai_targets_early()
        self.helpers = [self._get_helper(ebdy, helper) for ebdy, helper in zip(self.ebdyc, helpers)]
        self.AS_list = self.helpers
        self.grid = self.ebdyc.grid
        self.kx, self.ky = self.ebdyc.kx, self.ebdyc.ky
        self.ikx, self.iky = self.ebdyc.ikx, self.ebdyc.iky
        self.lap = -self.kx * self.kx - self.ky * self.ky
        self.plan = get_plan(self.grid.Nx, self.grid.Ny, self.grid.xh, self.grid.yh)
        self._get_specific_operators()
        self._set_derivative_method()
        self.interpolation_order = 3 if self.solver_type == 'fourth' else np.inf
        # values and gradient of the grid solution on the interfaces: through the library's
        # oversampled-FFT interpolation when the plan has it (power-of-two grids), else from the
        # full spectrum by the dense Fourier sums of ipde_amd.interp
        self._fast_interp = self.USE_FAST_INTERP and self.interpolation_order == np.inf \
            and self.plan.keep_spectrum(True)
        if self._fast_interp:      # its one-time state in the background of the set-up, not in the first solve
            prewarm_submit(("interp", self.grid.Nx, self.grid.Ny), self.plan.prepare_interp)
        self.grid_step = self.ebdyc.grid_step
        self._define_layer_apply()
        self._collect_grid_sources()
        self._make_device_state()
        self._resolve_grid_backend()
        self._define_grid_evaluator()
        if os.environ.get("IPDE_PATCH_TARGETS", "1") == "0":
            self.PATCH_TARGETS = False      # A/B switch: the list kernel
        if os.environ.get("IPDE_FAR_EXPANSION", "1") == "0":
            self.FAR_EXPANSION = False      # A/B switch: every pair directly
        if self.PATCH_TARGETS and hasattr(self.Grid_Evaluator, "prepare"):
            # the grid_pnai list into HBM and into patches now, in the background of the set-up,
            # instead of inside the first solve
            self.Grid_Evaluator.prepare()

    # the dense sum onto grid_pnai through the 4 x 4 patch kernel (kernels that have one: Laplace)
    PATCH_TARGETS = False
    # ... with the far sources of every 8 x 8 block of patches in a local expansion (Laplace:
    # ipde_laplace_apply_patches_far; to rounding, 2048^2 x 4096: 3.3 -> 0.3 ms; IPDE_FAR_EXPANSION=0
    # or this attribute False: every pair directly)
    FAR_EXPANSION = False

    # grid_backend None / 'auto': the package's choice — the dense sum; for a solver WITHOUT the far-field
    # form of it, beyond this many source-target pairs per solve and in a single process, the Ewald-type
    # split of the reference's grid evaluators (1e-13; configs[3], 8.5e10 pairs: 46 -> 16 ms for +0.3 s of
    # set-up; with the far-field form the dense sum takes 14-16 ms there).  Under torch.distributed the
    # dense sum, sharded over the ranks.
    AUTO_EWALD_MIN_PAIRS = 5.0e10

    def _resolve_grid_backend(self):
        if self.grid_backend in (None, 'auto'):
            pairs = float(self.grid_sources.N) * float(self.ebdyc.grid_pnai.N)
            # (solvers whose dense sum takes the far sources through local expansions — Poisson,
            # modified Helmholtz — keep it at every size: configs[3] 14-16 ms either way, and the
            # split costs 0.3 s of set-up)
            far = self.PATCH_TARGETS and self.FAR_EXPANSION and os.environ.get("IPDE_FAR_EXPANSION", "1") != "0"
            big = pairs >= self.AUTO_EWALD_MIN_PAIRS and not is_distributed() and not far
            self.grid_backend = 'ewald' if big else 'hip'

    CONCURRENT_ANNULAR = True     # False: annular solves one boundary after the other
    DISTRIBUTE_BOUNDARIES = True  # under torch.distributed: boundary i on rank i mod world
    USE_FAST_INTERP = True        # False: always the dense Fourier sums (the checker)
    DEVICE_FLOW = True            # False: per-boundary vectors travel as numpy between the stages (the round-1 flow)

    def _concurrent_helpers(self):
        return _concurrent_helpers(self)

    def _collect_grid_sources(self):
        # with a single boundary every rank runs the whole helper flow: its radial sums can
        # then be sharded collectively (helpers evaluate them through sharded evaluators)
        for helper in self.helpers:
            helper.shard_radial_sums = len(self.helpers) == 1 and is_distributed()
        self.grid_sources = BoundaryCollection()
        for helper in self.helpers:
            self.grid_sources.add(helper.interface_qfs_g.source, 'i' if helper.interior else 'e')
        self.grid_sources.amass_information()

    def _make_device_state(self):
        """Resident copies of everything grid-sized the solve touches."""
        e = self.ebdyc
        dev = self.plan.ctx.torch_device()
        self._dev = dev
        flat = lambda mask: torch.as_tensor(np.flatnonzero(mask.ravel()), device=dev)
        self._phys_idx = flat(e.phys)
        self._pna_idx = flat(e.phys_not_in_annulus)
        self._grid_step_d = torch.as_tensor(np.ascontiguousarray(e.grid_step), device=dev)
        self._phys_d = torch.as_tensor(e.phys.astype(float), device=dev)
        self._ikx_d = torch.as_tensor(np.ascontiguousarray(self.ikx), device=dev)
        self._iky_d = torch.as_tensor(np.ascontiguousarray(self.iky), device=dev)
        self._ifx_d = torch.as_tensor(e.interfaces_x_transf, device=dev)
        self._ify_d = torch.as_tensor(e.interfaces_y_transf, device=dev)
        self._ia = []
        for ebdy in e:
            idx = torch.as_tensor(ebdy.grid_ia_xind * self.grid.Ny + ebdy.grid_ia_yind, device=dev)
            self._ia.append((idx, torch.as_tensor(ebdy.grid_ia_xi, device=dev),
                             torch.as_tensor(ebdy.grid_ia_t, device=dev)))
        self._pin_in = torch.empty(e.grid_phys.N, dtype=torch.float64, pin_memory=True)

    def _start_pnai_targets_early(self):
        """The grid_pnai list into HBM and into its patch plan from the first line of the set-up: the
        plan is host work in a thread of its own (0.3 s at 4096^2), and started where the evaluator is
        defined — after the helpers' annular solvers and QFS factorisations — the first solve waited
        for it (configs[3]: first solve 0.26 s).  Single process, planned dense sums only; the
        evaluator's target factory hands this object out if its flags still match."""
        self._early_pnai = None
        gb = self.grid_backend
        if is_distributed() or not self.PATCH_TARGETS or os.environ.get("IPDE_PATCH_TARGETS", "1") == "0":
            return
        if not (gb is None or gb in ('auto', 'hip', 'pybie2d', 'fmm2d', 'flexmm')):
            return
        far = self.FAR_EXPANSION and os.environ.get("IPDE_FAR_EXPANSION", "1") != "0"
        if gb == 'pybie2d':
            far = False
        elif gb in ('fmm2d', 'flexmm'):
            far = True
        if gb in (None, 'auto') and not far:
            return                      # (may resolve to the split evaluator)
        from ...pybie2d_compat import PointSet
        pn = self.ebdyc.grid_pnai
        self._early_pnai = (pn.x, pn.y, bool(far), DeviceTargets(PointSet(x=pn.x, y=pn.y), plan=True, far=far))

    def _pnai_evaluator(self):
        """density -> potential on grid_pnai.  The solver evaluates onto the same target
        set in every solve: it stays in HBM; under torch.distributed the targets are
        sharded over the ranks (ipde_amd.sharding.make_pnai_evaluator)."""
        from ...pybie2d_compat import PointSet

        def targets(x, y):
            far = bool(self.PATCH_TARGETS and self.FAR_EXPANSION)
            early = getattr(self, "_early_pnai", None)
            if early is not None and self.PATCH_TARGETS and x is early[0] and y is early[1] and far == early[2]:
                return early[3]
            return DeviceTargets(PointSet(x=x, y=y), plan=self.PATCH_TARGETS, far=far)
        return make_pnai_evaluator(lambda src, trg, ch: self.Layer_Apply(src, trg, ch),
                                   self.grid_sources, self.ebdyc.grid_pnai, targets)

    def _get_helper(self, ebdy, helper):
        raise NotImplementedError

    def _grid_solve(self, fc):
        raise NotImplementedError

    def _get_specific_operators(self):
        pass

    def _set_derivative_method(self):
        if self.solver_type == 'spectral':
            self.dx = lambda x: self.plan.dx(x)
            self.dy = lambda x: self.plan.dy(x)
        else:
            self.dx = lambda x: fd_x_4(x, self.ebdyc.grid.xh)
            self.dy = lambda x: fd_y_4(x, self.ebdyc.grid.yh)

    def get_boundary_values(self, u):
        bv = BoundaryFunction(self.ebdyc)
        bv.load_data([helper.get_boundary_values(ur) for ur, helper in zip(u, self.helpers)])
        return bv

    def get_boundary_normal_derivatives(self, u):
        bv = BoundaryFunction(self.ebdyc)
        bv.load_data([helper.get_boundary_normal_derivatives(ur) for ur, helper in zip(u, self.helpers)])
        return bv

    def evaluate_to_grid_pnai(self, sigmag):
        """(reference :63-71) -> device tensor over grid_pnai"""
        if self.split_grid_evaluation:
            grid_out = self.Grid_Evaluator(sigmag)
            grid_out = grid_out if isinstance(grid_out, torch.Tensor) \
                else torch.as_tensor(grid_out, device=self._dev)
            grid_pna = gridops.gather(self._pna_idx, grid_out.reshape(-1).contiguous())
            if isinstance(sigmag, torch.Tensor):     # device flow: the interface nodes resident too
                if getattr(self, '_all_iv_dev', None) is None:
                    self._all_iv_dev = DeviceTargets(self.ebdyc.all_iv)
                interface_out = self.Layer_Apply(self.grid_sources, self._all_iv_dev, sigmag)
            else:
                interface_out = self.Layer_Apply(self.grid_sources, self.ebdyc.all_iv, sigmag)
            return torch.cat([grid_pna, torch.as_tensor(interface_out, device=self._dev)])
        return self.Grid_Evaluator(sigmag)

    def __call__(self, f, sharded_result=False, **kwargs):
        """f: EmbeddedFunction -> EmbeddedFunction (reference :72-117).

        sharded_result (under torch.distributed): the dense sum onto grid_pnai is left sharded through
        the masked add — every rank returns an answer whose grid values outside the annuli are
        complete only on its own slice of grid_pnai, flagged by the boolean `owned` attribute of the
        result (everything else is complete everywhere); the per-solve exchange of that sum drops from
        the whole list to the interface values.  Default: the answer replicated on every rank."""
        prewarm_wait()      # FFT plans below: never concurrently with the warm-up thread
        e = self.ebdyc
        Nx, Ny = self.grid.shape
        fr_list = f.get_radial_value_list()
        # fc = (grid values) * grid_step on the full grid  (embedded_function.py:135-138)
        resident = isinstance(f, hostio.DeviceFunction)      # right-hand side (and answer) in HBM
        if resident:
            if f.ebdyc is not e or f.data.device != self._dev:
                raise ValueError("DeviceFunction of another collection or device")
            fp = f.grid_values
        else:
            fp = torch.empty(e.grid_phys.N, dtype=torch.float64, device=self._dev)
            hostio.upload(fp, f['grid'], self._pin_in)
        fc = gridops.scatter(self._phys_idx, fp, Nx * Ny, scale=self._grid_step_d).view(Nx, Ny)
        uch, uc = self._grid_solve(fc)
        uc = uc.contiguous()
        if self.interpolation_order == np.inf and uch is None:
            # values and gradient on all interface nodes (:80-88) from the spectrum the grid
            # solve left on the device: 2x oversampled inverse FFT + window gather (csrc/nufft.hip)
            all_bvs = self.plan.interp_gradient(self._ifx_d, self._ify_d)
        elif self.interpolation_order == np.inf:
            # the same from the full spectrum by dense Fourier sums; the three fields share one
            # set of exponential matrices
            all_bvs = periodic_interp2d_gradient(uch, self._ifx_d, self._ify_d, self._ikx_d, self._iky_d)
        else:
            # (the library's own D2Z plan: one rocFFT in the process, no torch.fft)
            stack = torch.stack([self.plan.fft2(g.contiguous()) for g in (uc, self.dx(uc), self.dy(uc))])
            all_bvs = periodic_interp2d(stack, self._ifx_d, self._ify_d, real_part=True)
        # The per-boundary vectors stay in HBM through all the stages below (helpers' device forms,
        # qfs.call_many / u2s_many and the layer applies pass device tensors through); under
        # torch.distributed the exchanges of the owned boundaries' vectors take the tensors as they
        # are (sharding.exchange_owned: a flat device buffer, nothing through the host with nccl).
        device_flow = self.DEVICE_FLOW
        if not device_flow:
            all_bvs = all_bvs.cpu().numpy()
        bvl, bxl, byl = e.v2l(all_bvs[0]), e.v2l(all_bvs[1]), e.v2l(all_bvs[2])
        # annular solves boundary by boundary, then the QFS solves of all boundaries in one
        # batched substitution (qfs.call_many); under torch.distributed each rank does this
        # for the boundaries it owns and the grid-side densities are exchanged
        mine, distributed = _owned(self)
        args = list(zip(fr_list, bvl, bxl, byl))
        sigmag_list = _run_owned(self, mine, 'start_call', 'finish_call', args, call_many, **kwargs)
        its = [float(h.iterations_last_call) if i in mine else 0.0 for i, h in enumerate(self.helpers)]
        if distributed:
            # (the kind of the result is stated, not inferred: a rank that owns no boundary — world >
            # number of boundaries — must come back with tensors like the others)
            sigmag_list, its = exchange_owned(sigmag_list, [(h.interface_qfs_g.source.N,) for h in self.helpers],
                                              device=self._dev, extra=its, as_tensors=device_flow)
        self.iteration_counts = [int(i) for i in its]
        sigmag = gridops.concat(list(sigmag_list)) if device_flow else np.concatenate(sigmag_list)
        n_pna = e.grid_pna.N
        ucf = uc.view(-1)
        owned = None
        if sharded_result and hasattr(self.Grid_Evaluator, "sharded") and not self.split_grid_evaluation:
            # the sum onto grid_pnai with its grid part LEFT sharded ("outputs stay
            # sharded for the subsequent masked add"): this rank adds its slice of the list onto its
            # copy of the grid; only the tail of the list, the values on the interface nodes every
            # boundary's correction needs, is exchanged (n_interface numbers instead of the list)
            so = self.Grid_Evaluator.sharded(sigmag, int(e.grid_pnai.N - n_pna))
            a, b = so.slice.start, min(so.slice.stop, n_pna)
            if b > a:
                gridops.add_at(self._pna_idx[a:b], so.local[:b - a].contiguous(), ucf)
            tail = so.tail
            owned = (a, max(a, b))
        else:
            out = self.evaluate_to_grid_pnai(sigmag)                 # device, len(grid_pnai)
            gridops.add_at(self._pna_idx, out[:n_pna].contiguous(), ucf)
            tail = out[n_pna:]
        bus = e.v2l(tail if device_flow else tail.cpu().numpy())
        urs = _run_owned(self, mine, 'start_correct', 'finish_correct', [(bu,) for bu in bus], u2s_many)
        if distributed:
            urs = exchange_owned(urs, [h.ebdy.radial_shape for h in self.helpers], device=self._dev,
                                 as_tensors=device_flow)
            for h, ur in zip(self.helpers, urs):
                h.ur = ur
        for ur, (idx, xi, t) in zip(urs, self._ia):
            radial_to_grid([ur], xi, t, idx=idx, outs=[ucf])
        # (only the physical points leave the device: the reference's masking of the rest, :117, has
        # nothing to act on)
        # the answer is built over pinned memory: the device->host copy writes the caller's array
        if resident:
            if owned is not None:
                raise ValueError("sharded_result answers are host containers (the `owned` mask)")
            ud = hostio.DeviceFunction(e, device=self._dev)
            for sl, ur in zip(ud.radial_slices, urs):
                ud.data[sl].copy_(torch.as_tensor(ur, device=self._dev).reshape(-1))
            gridops.gather(self._phys_idx, ucf, out=ud.data[:ud.n_grid])
            return ud
        ue, block = hostio.pinned_function(e)
        for i, (sl, ur) in enumerate(zip(ue.radial_slices, urs)):
            if isinstance(ur, torch.Tensor):     # device flow: the annular solutions' one transfer
                block[sl].copy_(ur.reshape(-1), non_blocking=True)
            else:
                ue[i] = ur                       # (numpy copy: no torch CPU op, see VectorSolver)
        block[:e.grid_phys.N].copy_(gridops.gather(self._phys_idx, ucf), non_blocking=False)   # (in stream order: the last)
        if owned is not None:
            # which entries of the answer are complete on THIS rank: the radial parts and the grid
            # points inside the annuli everywhere, a grid point outside the annuli on the rank whose
            # slice of grid_pnai holds it
            ue.owned = self._owned_mask(ue, owned)
        return ue

    def _owned_mask(self, ue, owned):
        """A PARTITION of the answer's entries over the ranks: a grid point outside the annuli belongs
        to the rank whose slice of grid_pnai holds it (only there is its value complete); the grid
        points inside the annuli and the radial values, complete on every rank, are split into
        contiguous runs so that sums over the answer (corrections, norms) count each entry once."""
        from ...sharding import target_slice
        e = self.ebdyc
        _, rank, world = _dist_state()
        if getattr(self, "_pna_in_phys", None) is None:
            in_pna = np.zeros(ue.shape[0], dtype=bool)
            in_pna[np.nonzero(e.phys_not_in_annulus[e.phys])[0]] = True
            self._pna_in_phys = np.nonzero(in_pna)[0]
            self._not_pna = np.nonzero(~in_pna)[0]
        m = np.zeros(ue.shape[0], dtype=bool)
        m[self._pna_in_phys[owned[0]:owned[1]]] = True
        m[self._not_pna[target_slice(self._not_pna.shape[0], rank, world)]] = True
        return m

    def _define_layer_apply(self):
        self.Layer_Apply = self.helpers[0].Layer_Apply
