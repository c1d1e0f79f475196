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
        self._radial_dev = DeviceTargets(self.ebdy.radial_targ)
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
                inx=up(iface.normal_x), iny=up(iface.normal_y), iw=up(iface.weights),
                bnx=up(b.normal_x), bny=up(b.normal_y), btx=up(b.tangent_x), bty=up(b.tangent_y),
                D00=up(e.D00), ik=torch.as_tensor(1j * np.asarray(e.radial_k, dtype=float), device=dev),
                rs=up(e.radial_speed), irs=up(e.inverse_radial_speed), iv_est=up(self._iv_estimator),
                zero=torch.zeros(b.N, dtype=torch.float64, device=dev))
        return c

    def _interface_traction_uvp_device(self, u, v, p):
        """get_interface_traction_uvp (:65-112) on device tensors"""
        import torch
        from ...spectral import fft1
        c = self._device_constants()
        ctx = getattr(self.annular_solver, 'ctx', None)
        tder = lambda f: fft1(fft1(f, -1, ctx) * c['ik'], +1, ctx).real      # (fft1 scales the inverse)
        est = lambda X: torch.mv(X.t(), c['iv_est'])
        Ur, Ut = u * c['bnx'] + v * c['bny'], u * c['btx'] + v * c['bty']
        Urr = c['D00'] @ Ur
        Urt = tder(Ur) * c['irs']
        Utr = c['rs'] * (c['D00'] @ (Ut * c['irs']))
        Tr = 2 * est(Urr) - est(p)
        Tt = est(Utr) + est(Urt)
        return Tr * c['bnx'] + Tt * c['btx'], Tr * c['bny'] + Tt * c['bty']

    def _start_call_device(self, fur, fvr, bu, bv, btxx, btxy, btyy, **kwargs):
        import torch
        c = self._device_constants()
        btx = btxx * c['inx'] + btxy * c['iny']
        bty = btxy * c['inx'] + btyy * c['iny']
        f2 = torch.as_tensor(np.ascontiguousarray(np.stack([fur, fvr]), dtype=float), device=bu.device)
        fr, ft = f2[0] * c['bnx'] + f2[1] * c['bny'], f2[0] * c['btx'] + f2[1] * c['bty']
        z = c['zero']
        rr, tr, pr = self.annular_solver.solve(self.RAG, fr, ft, z, z, z, z, **kwargs)
        self.iterations_last_call = self.annular_solver.iterations_last_call
        ur, vr = rr * c['bnx'] + tr * c['btx'], rr * c['bny'] + tr * c['bty']
        rtx, rty = self._interface_traction_uvp_device(ur, vr, pr)
        taus = torch.cat([rtx - btx, rty - bty])
        taud = torch.cat([bu, bv])
        if not self.interior:
            taus = -taus
            taud = -taud
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
