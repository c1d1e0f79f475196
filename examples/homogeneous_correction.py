"""The boundary-condition (homogeneous) correction shared by the single-boundary scalar examples —
reference examples/interior_poisson.py:84-92, examples/interior_modified_helmholtz.py:66-84."""
import os

import numpy as np

from ipde_amd.layer_potentials import ShardedTargets
from ipde_amd.qfs import QFS_Evaluator, DenseSolver


class HomogeneousCorrection(object):
    """The boundary-condition correction of the reference's example (examples/interior_poisson.py:84-92;
    timed as "homogeneous form" / "homogeneous apply" in examples/poisson_for_paper.py:78-92): form = the
    on-surface double-layer matrix, its LU, the QFS evaluator and the resident target set; apply =
    boundary values of the inhomogeneous solution, density tau, QFS source density sigma, ONE single-layer
    sum onto grid_and_radial_pts, added onto the solution.

    The sum goes onto `ebdyc.resident_grid_and_radial_pts()`: the physical grid points in 4 x 4 patches with
    every block's far sources in a local expansion, the radial grid in blocks of 64 radial lines
    (ipde_{laplace,modhelm}_apply_patches_far / _columns_far; far=False: every pair directly).  Given a
    hostio.DeviceFunction (a resident solve's answer) every step stays in HBM and the answer is a
    DeviceFunction; given the reference's host container the result comes back through PCIe once."""

    def __init__(self, solver, bdy, ebdy, bc, singular_dlp, naive_slp, layer_apply, owned=None, far=True):
        """singular_dlp(src, trg): the second-kind on-surface matrix (D -/+ I/2); naive_slp(src, trg): the
        source-to-boundary single-layer matrix of the QFS evaluator; layer_apply(source, targets, charge): the
        kernel family's single-layer sum"""
        import torch
        self.solver, self.ebdyc, self.owned, self.layer_apply = solver, solver.ebdyc, owned, layer_apply
        A = singular_dlp(bdy, bdy)
        self.qfs = QFS_Evaluator(ebdy.bdy_qfs, True, [lambda src, trg: A, ], naive_slp, on_surface=True,
                                 form_b2c=False)
        # (second-kind system, condition O(10): the plain substitution already has LAPACK's residual — measured at
        # 2048^2 / 4096 nodes, tools/ab_correction.py: error 1.2553e-14 without the refinement step, 1.2733e-14 with
        # it, 0.27 ms of the stage saved; IPDE_CORRECTION_REFINE=1 puts it back)
        self.Alu = DenseSolver(A, refine=int(os.environ.get('IPDE_CORRECTION_REFINE', '0')))
        from ipde_amd.sharding import is_distributed
        if owned is not None or is_distributed():
            self.targets = ShardedTargets(self.ebdyc.grid_and_radial_pts, owned=owned)
        else:
            self.targets = self.ebdyc.resident_grid_and_radial_pts(far=far)
        self.bc_host = np.concatenate(bc.bdy_value_list)
        self.bc_dev = torch.as_tensor(self.bc_host, device=torch.device('cuda', torch.cuda.current_device()))
        self._est = None

    def boundary_values(self, ue):
        """boundary trace of the radial part (solver.get_boundary_values, reference :85) — on the device
        for a resident answer: the Chebyshev boundary row times the (M, N) radial block"""
        import torch
        if self._est is None:
            self._est = [torch.as_tensor(np.ascontiguousarray(h._bv_estimator, dtype=float), device=self.bc_dev.device)
                         for h in self.solver.helpers]
        return torch.cat([e @ ur for e, ur in zip(self._est, ue.get_radial_value_list())])

    def __call__(self, ue):
        import torch
        from ipde_amd.hostio import DeviceFunction
        if isinstance(ue, DeviceFunction):
            tau = self.Alu.solve(self.bc_dev - self.boundary_values(ue))
            sigma = self.qfs([tau, ])
            ue.data += self.layer_apply(self.ebdyc.bdy_inward_sources, self.targets, sigma)
            return ue
        bv = self.solver.get_boundary_values(ue.get_radial_value_list())
        tau = self.Alu.solve(self.bc_host - np.concatenate(bv.bdy_value_list))
        sigma = self.qfs([tau, ])
        out = self.layer_apply(self.ebdyc.bdy_inward_sources, self.targets, sigma)
        if self.owned is None and isinstance(out, torch.Tensor) and out.is_cuda and ue.flags.c_contiguous:
            # grid_and_radial_pts is the container's own storage order (reference ipde/embedded_function.py:24-31):
            # the sum is added onto it where it lies, copy and additions overlapped (hostio.add_from_device)
            from ipde_amd import hostio
            self._pin = hostio.add_from_device(np.asarray(ue).view(np.ndarray), out, getattr(self, '_pin', None))
            return ue
        gslp, rslpl = self.ebdyc.divide_grid_and_radial(out.cpu().numpy())
        for i, r in enumerate(rslpl):
            ue[i] += r.reshape(self.ebdyc[i].radial_shape)
        ue['grid'] += gslp
        return ue
