"""ScalarHelper — per-boundary helper of the scalar solvers, mirrors
ipde/solvers/internals/scalar.py:5-116: owns the annular solver, the interface QFS
pair and the `Layer_Apply` closure (the plug point of the GPU kernels)."""
import numpy as np

from ...qfs import call_many, u2s_many
from ...annular.annular_full import ApproximateAnnularGeometry, RealAnnularGeometry


class ScalarHelper(object):
    def __init__(self, ebdy, helper=None, grid_backend='hip', private_ctx=False):
        # private_ctx: a library context of its own for the annular solver (stream, work
        # buffers, FFT plans) — the multi-boundary solver then runs the annular solves of its
        # boundaries concurrently from separate host threads
        self.ebdy = ebdy
        self._private_ctx = private_ctx
        self.grid_backend = grid_backend
        self.interior = self.ebdy.interior
        if helper is None:
            self.AAG = ApproximateAnnularGeometry(self.ebdy.bdy.N, self.ebdy.M,
                                                  self.ebdy.radial_width, self.ebdy.approximate_radius)
            self._define_annular_solver()
        else:
            self.annular_solver = helper.annular_solver
            self.AAG = self.annular_solver.AAG
        self._set_boundary_estimators()
        self._get_RAG()
        self._get_qfs()
        self._define_layer_apply()
        # fixed target sets of correct(): resident in HBM
        from ...layer_potentials import DeviceTargets
        self._interface_dev = DeviceTargets(self.ebdy.interface)
        self._radial_dev = DeviceTargets(self.ebdy.radial_targ, columns=self.ebdy.radial_shape)
        # set by the multi-boundary solver: True when every rank of a torch.distributed job
        # runs this helper's whole flow (single boundary), so that the M*N x N radial sum of
        # correct() can be split over the ranks like the grid sum
        self.shard_radial_sums = False
        self._radial_sharded = None

    def _radial_sum(self, src, density):
        """Layer_Apply onto the radial targets (reference :113-114), target-sharded under
        torch.distributed when the sum is large enough to pay for the all-gather"""
        if not self.shard_radial_sums:
            return self.Layer_Apply(src, self._radial_dev, density)
        if self._radial_sharded is None:
            from ... import sharding
            from ...layer_potentials import DeviceTargets
            from ...pybie2d_compat import PointSet
            self._radial_sharded = sharding.make_sharded_evaluator(
                lambda s, t, d: self.Layer_Apply(s, t, d), self.ebdy.radial_targ,
                lambda x, y: DeviceTargets(PointSet(x=x, y=y)), min_pairs=sharding.MIN_PAIRS_TO_SHARD)
        return self._radial_sharded(src, density)

    def _annular_ctx(self):
        if not self._private_ctx:
            return None
        from ...device import private_context
        return private_context()

    def _define_annular_solver(self):
        raise NotImplementedError

    def _get_qfs(self):
        raise NotImplementedError

    def _define_layer_apply(self):
        raise NotImplementedError

    def _get_RAG(self):
        bb = self.ebdy.bdy if self.interior else self.ebdy.interface
        self.RAG = RealAnnularGeometry(bb.speed, bb.curvature, self.annular_solver.AAG)

    def _set_boundary_estimators(self):
        CO = self.annular_solver.AAG.CO
        if self.interior:
            self._bv_estimator, self._bn_estimator = CO.obc_dirichlet[0], CO.obc_neumann[0]
            self._iv_estimator, self._in_estimator = CO.ibc_dirichlet[0], CO.ibc_neumann[0]
        else:
            self._bv_estimator, self._bn_estimator = CO.ibc_dirichlet[0], CO.ibc_neumann[0]
            self._iv_estimator, self._in_estimator = CO.obc_dirichlet[0], CO.obc_neumann[0]

    def get_boundary_values(self, ur):
        return self._bv_estimator.dot(ur)

    def get_boundary_normal_derivatives(self, ur):
        return self._bn_estimator.dot(ur)

    def get_interface_values(self, ur):
        return self._iv_estimator.dot(ur)

    def get_interface_normal_derivatives(self, ur):
        return self._in_estimator.dot(ur)

    # The reference's __call__ (:68-94) and correct (:95-116) each contain dense QFS solves.
    # They are split in two around them (start_* returns the solves it needs, finish_* takes
    # their results) so that the multi-boundary solver can run the solves of ALL boundaries
    # in one batched substitution (qfs.call_many / u2s_many); __call__ and correct keep the
    # reference's one-boundary form.
    def __call__(self, fr, bv, bx, by, **kwargs):
        """kwargs go to the annular solver (reference :68-94)."""
        return self.finish_call(*call_many(self.start_call(fr, bv, bx, by, **kwargs)))

    # Device-resident form of the same two stages (the multi-boundary solver's default in a
    # single process): interface data arrives as device tensors and every per-boundary vector
    # — annular solution, jumps, densities, corrections — stays in HBM until the solve's one
    # transfer of the answer; the arithmetic is the host form's, statement for statement.
    def _device_constants(self):
        c = getattr(self, '_dev_const', None)
        if c is None:
            import torch
            dev = self._interface_dev.x.device
            up = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=float), device=dev)
            iface = self.ebdy.interface
            c = self._dev_const = dict(nrm=up(np.stack([iface.normal_x, iface.normal_y])),
                                       in_est=up(self._in_estimator),
                                       zero=torch.zeros(iface.N, dtype=torch.float64, device=dev))
        return c

    def _start_call_device(self, fr, bv, bx, by, **kwargs):
        import torch
        from ...device import get_context, ptr
        c = self._device_constants()
        frd = fr if isinstance(fr, torch.Tensor) else \
            torch.as_tensor(np.ascontiguousarray(fr, dtype=float), device=bv.device)
        ur = self.annular_solver.solve(self.RAG, frd, c['zero'], c['zero'], **kwargs)
        self.iterations_last_call = self.annular_solver.iterations_last_call
        # interface normal derivative (:83) and the jumps (:84-90): one kernel on the annular
        # solver's context (csrc/annular.hip: ipde_scalar_interface_jumps)
        ctx = getattr(self.annular_solver, 'ctx', None) or get_context()
        M, N = ur.shape
        from ... import gridops
        bdata = gridops.rows([bv, bx, by])
        slp = torch.empty(N, dtype=torch.float64, device=bv.device)
        dlp = torch.empty(N, dtype=torch.float64, device=bv.device)
        ctx.check(ctx.lib.ipde_scalar_interface_jumps(ctx.handle, M, N, ptr(ur), ptr(c['in_est']), ptr(c['nrm']),
                                                      ptr(bdata), 1.0 if self.interior else -1.0, ptr(slp), ptr(dlp)))
        self.ur = ur
        return [(self.interface_qfs_g, [slp, dlp]), (self.interface_qfs_r, [slp, dlp])]

    def start_call(self, fr, bv, bx, by, **kwargs):
        if type(bv).__module__.startswith('torch'):
            return self._start_call_device(fr, bv, bx, by, **kwargs)
        ebdy = self.ebdy
        ucn = bx * ebdy.interface.normal_x + by * ebdy.interface.normal_y
        zer = np.zeros_like(bv)
        ur = np.asarray(self.annular_solver.solve(self.RAG, fr, zer, zer, **kwargs))
        self.iterations_last_call = self.annular_solver.iterations_last_call
        urn = self.get_interface_normal_derivatives(ur)
        slp = urn - ucn
        dlp = np.array(bv, copy=True)
        if not self.interior:
            slp *= -1.0
            dlp *= -1.0
        self.ur = ur
        return [(self.interface_qfs_g, [slp, dlp]), (self.interface_qfs_r, [slp, dlp])]

    def finish_call(self, sigma_g, sigma_r):
        self.sigma_r = sigma_r
        self.sigma_g = sigma_g
        return sigma_g

    def correct(self, ub):
        """(reference :95-116)"""
        return self.finish_correct(*u2s_many(self.start_correct(ub)))

    def start_correct(self, ub):
        src = self.interface_qfs_g.source
        w = self.Layer_Apply(src, self._interface_dev, self.sigma_g)
        if not type(ub).__module__.startswith('torch'):
            w = w.cpu().numpy()
        return [(self.interface_qfs_r, ub - w)]

    def finish_correct(self, sigma_r_adj):
        sigma_r_tot = sigma_r_adj + self.sigma_r
        src = self.interface_qfs_r.source
        rslp = self._radial_sum(src, sigma_r_tot)
        if not type(self.ur).__module__.startswith('torch'):
            rslp = rslp.cpu().numpy()
        self.ur = self.ur + rslp.reshape(self.ur.shape)
        return self.ur
