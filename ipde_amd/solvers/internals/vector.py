"""VectorHelper — per-boundary helper of the vector (Stokes) solver, mirrors
ipde/solvers/internals/vector.py:7-162: owns the annular solver, the interface QFS pair
(single + double layer of the traction / velocity jumps) and the `Layer_Apply` closure
returning (u, v, p) — the plug point of the GPU Stokes kernel."""
import numpy as np

from ...qfs import call_many, u2s_many
from ...annular.annular import ApproximateAnnularGeometry
from ...annular.annular_full import RealAnnularGeometry


def v2f(x):
    return x.reshape(2, -1)


def _is_dev(a):
    return type(a).__module__.startswith('torch')


class VectorHelper(object):
    def __init__(self, ebdy, annular_solver=None, **kwargs):
        self.ebdy = ebdy
        self.interior = self.ebdy.interior
        self._extract_extra_kwargs(**kwargs)
        if annular_solver is None:
            self.AAG = ApproximateAnnularGeometry(self.ebdy.bdy.N, self.ebdy.M,
                                                  self.ebdy.radial_width, self.ebdy.approximate_radius)
            self._define_annular_solver()
        else:
            self.annular_solver = annular_solver
            self.AAG = self.annular_solver.AAG
        self._set_boundary_estimators()
        self._get_RAG()
        self._get_qfs()
        self._define_layer_apply()
        from ...layer_potentials import DeviceTargets
        self._interface_dev = DeviceTargets(self.ebdy.interface)
        self._radial_dev = DeviceTargets(self.ebdy.radial_targ, columns=self.ebdy.radial_shape)
        self.shard_radial_sums = False     # see ScalarHelper
        self._radial_sharded = None

    def _radial_sum(self, src, density):
        """(u, v, p) of the stokeslet sum onto the radial targets, target-sharded under
        torch.distributed when every rank runs this helper (see ScalarHelper._radial_sum)"""
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

    def _extract_extra_kwargs(self, **kwargs):
        pass

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
        CO = self.AAG.CO
        if self.interior:
            self._bv_estimator, self._bn_estimator = CO.obc_dirichlet[0], CO.obc_neumann[0]
            self._iv_estimator, self._in_estimator = CO.ibc_dirichlet[0], CO.ibc_neumann[0]
        else:
            self._bv_estimator, self._bn_estimator = CO.ibc_dirichlet[0], CO.ibc_neumann[0]
            self._iv_estimator, self._in_estimator = CO.obc_dirichlet[0], CO.obc_neumann[0]

    def get_boundary_values(self, fr):
        return self._bv_estimator.dot(fr)

    def get_interface_values(self, fr):
        return self._iv_estimator.dot(fr)

    # -- tractions T(U) n on the boundary / interface (reference :65-112) -------------
    def get_boundary_traction_uvp(self, u, v, p):
        return self._get_traction_uvp(u, v, p, self._bv_estimator)

    def get_boundary_traction_rtp(self, Ur, Ut, p):
        return self._get_traction_rtp(Ur, Ut, p, self._bv_estimator)

    def get_interface_traction_uvp(self, u, v, p):
        return self._get_traction_uvp(u, v, p, self._iv_estimator)

    def get_interface_traction_rtp(self, Ur, Ut, p):
        return self._get_traction_rtp(Ur, Ut, p, self._iv_estimator)

    def _get_traction_uvp(self, u, v, p, estimator):
        Ur, Ut = self.ebdy.convert_uv_to_rt(u, v)
        Tr, Tt = self._get_traction_rtp(Ur, Ut, p, estimator)
        return self.ebdy.convert_rt_to_uv(Tr, Tt)

    def _get_traction_rtp(self, Ur, Ut, p, estimator):
        ebdy = self.ebdy
        Urr = ebdy._radial_grid_r_derivative(Ur)
        Urt = ebdy._radial_grid_tau_derivative(Ur)
        Utr = ebdy.radial_speed * ebdy._radial_grid_r_derivative(Ut * ebdy.inverse_radial_speed)
        Tr = 2 * estimator.dot(Urr) - estimator.dot(p)
        Tt = estimator.dot(Utr) + estimator.dot(Urt)
        return Tr, Tt

    # (start_* / finish_* around the dense QFS solves: see ScalarHelper)
    def __call__(self, fur, fvr, bu, bv, btxx, btxy, btyy, **kwargs):
        """Annular solve with homogeneous data, traction / velocity jumps against the grid
        solution, QFS densities for both sides (reference :113-144)."""
        return self.finish_call(*call_many(self.start_call(fur, fvr, bu, bv, btxx, btxy, btyy, **kwargs)))

    # Device-resident form of the two stages (the multi-boundary solver's default in one process):
    # the interface data arrive as device tensors and the annular solution, tractions, jumps,
    # densities and corrections stay in HBM; the arithmetic is the host form's, statement for
    # statement (radial derivative = D00 product, tangential derivative = the library's batched
    # 1-D FFT on this helper's own context, estimators = matrix-vector products).
    def _device_constants(self):
        c = getattr(self, '_dev_const', None)
        if c is None:
            import torch
            dev = self._interface_dev.x.device
            up = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=float), device=dev)
            e, b, iface = self.ebdy, self.ebdy.bdy, self.ebdy.interface
            c = self._dev_const = dict(
                # boundary normal / tangent, interface normal: the `geom` block of the library calls
                geom=up(np.stack([b.normal_x, b.normal_y, b.tangent_x, b.tangent_y, iface.normal_x, iface.normal_y])),
                iw=up(iface.weights), D00=up(e.D00), rk=up(e.radial_k), rs=up(e.radial_speed),
                irs=up(e.inverse_radial_speed), iv_est=up(self._iv_estimator),
                zero=torch.zeros(b.N, dtype=torch.float64, device=dev))
        return c

    def _start_call_device(self, fur, fvr, bu, bv, btxx, btxy, btyy, **kwargs):
        """start_call on device tensors: two library calls around the annular solve
        (ipde_stokes_rotate, ipde_stokes_interface_jumps in csrc/annular.hip) on this helper's own
        context — the forcing's (r, t) components; the solution's (u, v) components, its
        interface traction and the jumps"""
        import torch
        from ... import _lib
        from ...device import get_context, ptr
        c = self._device_constants()
        ctx = getattr(self.annular_solver, 'ctx', None) or get_context()
        M, N = self.ebdy.radial_shape
        dev = bu.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        fr, ft = new(M, N), new(M, N)
        if isinstance(fur, torch.Tensor):       # forcing resident in HBM (hostio.DeviceFunction)
            loc, fu, fv = _lib.IPDE_DEVICE, fur.contiguous(), fvr.contiguous()
        else:
            loc = _lib.IPDE_HOST
            fu, fv = (np.ascontiguousarray(a, dtype=float) for a in (fur, fvr))
        ctx.check(ctx.lib.ipde_stokes_rotate(ctx.handle, loc, M, N, ptr(fu), ptr(fv), ptr(c['geom']), 1,
                                             ptr(fr), ptr(ft)))
        z = c['zero']
        rr, tr, pr = self.annular_solver.solve(self.RAG, fr, ft, z, z, z, z, **kwargs)
        self.iterations_last_call = self.annular_solver.iterations_last_call
        from ... import gridops
        bdata = gridops.rows([bu, bv, btxx, btxy, btyy])
        ur, vr, taus, taud = new(M, N), new(M, N), new(2 * N), new(2 * N)
        ctx.check(ctx.lib.ipde_stokes_interface_jumps(
            ctx.handle, M, N, ptr(rr), ptr(tr), ptr(pr), ptr(c['geom']), ptr(c['rs']), ptr(c['irs']), ptr(c['D00']),
            ptr(c['iv_est']), ptr(c['rk']), ptr(bdata), 1.0 if self.interior else -1.0, ptr(ur), ptr(vr),
            ptr(taus), ptr(taud)))
        self.ur, self.vr, self.pr = ur, vr, pr
        return [(self.interface_qfs_g, [taus, taud]), (self.interface_qfs_r, [taus, taud])]

    def start_call(self, fur, fvr, bu, bv, btxx, btxy, btyy, **kwargs):
        if _is_dev(bu):
            return self._start_call_device(fur, fvr, bu, bv, btxx, btxy, btyy, **kwargs)
        ebdy = self.ebdy
        btx = btxx * ebdy.interface.normal_x + btxy * ebdy.interface.normal_y
        bty = btxy * ebdy.interface.normal_x + btyy * ebdy.interface.normal_y
        fr, ft = ebdy.convert_uv_to_rt(fur, fvr)
        zer = np.zeros(ebdy.bdy.N)
        rr, tr, pr = self.annular_solver.solve(self.RAG, fr, ft, zer, zer, zer, zer, **kwargs)
        self.iterations_last_call = self.annular_solver.iterations_last_call
        rr, tr, pr = np.asarray(rr), np.asarray(tr), np.asarray(pr)
        ur, vr = ebdy.convert_rt_to_uv(rr, tr)
        rtx, rty = self.get_interface_traction_uvp(ur, vr, pr)
        taus = np.concatenate([rtx - btx, rty - bty])
        taud = np.concatenate([bu, bv])
        if not self.interior:
            taus *= -1.0
            taud *= -1.0
        self.ur, self.vr, self.pr = ur, vr, pr
        return [(self.interface_qfs_g, [taus, taud]), (self.interface_qfs_r, [taus, taud])]

    def finish_call(self, mu_g, mu_r):
        self.sigma_r = v2f(mu_r)
        self.sigma_g = v2f(mu_g)
        return self.sigma_g

    def correct(self, ub, vb, pb, single_ebdy):
        """Effect of every OTHER boundary's grid sources on this annulus (reference
        :145-162).  The reference leaves the pressure of that part undetermined (its
        comment at :150); here the constant is fixed by matching the mean pressure on the
        interface, which is all a velocity-matching density leaves open."""
        return self.finish_correct(*u2s_many(self.start_correct(ub, vb, pb, single_ebdy)))

    def start_correct(self, ub, vb, pb, single_ebdy):
        self._single = single_ebdy
        if single_ebdy:
            return []
        import torch
        src = self.interface_qfs_g.source
        w = self.Layer_Apply(src, self._interface_dev, self.sigma_g)
        if _is_dev(ub):
            self._pb_rest = pb - w[2]
            return [(self.interface_qfs_r, torch.cat([ub - w[0], vb - w[1]]))]
        # (one device -> host transfer for the three fields, not three synchronisations)
        w = torch.stack(list(w)).cpu().numpy()
        self._pb_rest = pb - w[2]
        return [(self.interface_qfs_r, np.concatenate([ub - w[0], vb - w[1]]))]

    def _finish_correct_device(self, mu_adj=None):
        import torch
        if self._single:
            sigma_r_tot = self.sigma_r
            p_shift = 0.0
        else:
            sigma_r_adj = v2f(mu_adj)
            p_adj = self.Layer_Apply(self.interface_qfs_r.source, self._interface_dev, sigma_r_adj)[2]
            wi = self._device_constants()['iw']
            p_shift = torch.sum((self._pb_rest - p_adj) * wi) / torch.sum(wi)
            sigma_r_tot = sigma_r_adj + self.sigma_r
        rslp = self._radial_sum(self.interface_qfs_r.source, sigma_r_tot)
        self.ur = self.ur + rslp[0].reshape(self.ur.shape)
        self.vr = self.vr + rslp[1].reshape(self.ur.shape)
        self.pr = self.pr + rslp[2].reshape(self.pr.shape) + p_shift
        return self.ur, self.vr, self.pr

    def finish_correct(self, mu_adj=None):
        if _is_dev(self.ur):
            return self._finish_correct_device(mu_adj)
        if self._single:
            sigma_r_tot = self.sigma_r
            p_shift = 0.0
        else:
            sigma_r_adj = v2f(mu_adj)
            p_adj = self.Layer_Apply(self.interface_qfs_r.source, self._interface_dev,
                                     sigma_r_adj)[2].cpu().numpy()
            wi = self.ebdy.interface.weights
            p_shift = np.sum((self._pb_rest - p_adj) * wi) / np.sum(wi)
            sigma_r_tot = sigma_r_adj + self.sigma_r
        src = self.interface_qfs_r.source
        import torch
        rslp = torch.stack(list(self._radial_sum(src, sigma_r_tot))).cpu().numpy()
        self.ur = self.ur + rslp[0].reshape(self.ur.shape)
        self.vr = self.vr + rslp[1].reshape(self.ur.shape)
        self.pr = self.pr + rslp[2].reshape(self.pr.shape) + p_shift
        return self.ur, self.vr, self.pr
