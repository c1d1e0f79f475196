"""ScalarSolver — the coupled grid + annuli driver, mirrors
ipde/solvers/multi_boundary/scalar.py:7-119 step for step:
  FFT grid solve -> values/gradient on the interfaces -> per-boundary annular solves
  and QFS densities -> Grid_Evaluator (dense GPU sum) -> correct() -> radial->grid.
"""
import numpy as np

from ...derivatives import fd_x_4, fd_y_4
from ...embedded_function import EmbeddedFunction, BoundaryFunction
from ...interp import periodic_interp2d
from ...layer_potentials import DeviceTargets
from ...pybie2d_compat import BoundaryCollection
from ...spectral import get_plan


class ScalarSolver(object):
    def __init__(self, ebdyc, solver_type='spectral', helpers=None, grid_backend='hip'):
        self.ebdyc = ebdyc
        self.solver_type = solver_type
        self.grid_backend = grid_backend
        if helpers is None:
            helpers = [None, ] * self.ebdyc.N
        self.helpers = [self._get_helper(ebdy, helper) for ebdy, helper in zip(self.ebdyc, helpers)]
        self.grid = self.ebdyc.grid
        self.kx, self.ky = self.ebdyc.kx, self.ebdyc.ky
        self.ikx, self.iky = self.ebdyc.ikx, self.ebdyc.iky
        self.lap = -self.kx * self.kx - self.ky * self.ky
        self.plan = get_plan(self.grid.Nx, self.grid.Ny, self.grid.xh, self.grid.yh)
        self._get_specific_operators()
        self._set_derivative_method()
        self.interpolation_order = 3 if self.solver_type == 'fourth' else np.inf
        self.grid_step = self.ebdyc.grid_step
        self._define_layer_apply()
        self._collect_grid_sources()
        self._define_grid_evaluator()

    def _collect_grid_sources(self):
        self.grid_sources = BoundaryCollection()
        for helper in self.helpers:
            self.grid_sources.add(helper.interface_qfs_g.source, 'i' if helper.interior else 'e')
        self.grid_sources.amass_information()
        # the solver evaluates onto the same target set in every solve: keep it in HBM
        self._grid_pnai_dev = DeviceTargets(self.ebdyc.grid_pnai)

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
        """(reference :63-71)"""
        if self.split_grid_evaluation:
            grid_out = self.Grid_Evaluator(sigmag)
            grid_pna = grid_out[self.ebdyc.phys_not_in_annulus]
            interface_out = self.Layer_Apply(self.grid_sources, self.ebdyc.all_iv, sigmag)
            return np.concatenate([grid_pna, interface_out])
        return self.Grid_Evaluator(sigmag)

    def __call__(self, f, **kwargs):
        """f: EmbeddedFunction -> EmbeddedFunction (reference :72-117)."""
        _, fc, fr_list = f.get_components()
        uch, uc = self._grid_solve(fc)
        if self.interpolation_order == np.inf:
            # values and gradient on all interface nodes from the spectrum (:80-88);
            # the three fields share one set of exponential matrices
            import torch
            uch_d = uch if isinstance(uch, torch.Tensor) else torch.as_tensor(uch, device="cuda")
            ikx = torch.as_tensor(self.ikx, device=uch_d.device)
            iky = torch.as_tensor(self.iky, device=uch_d.device)
            stack = torch.stack([uch_d, ikx * uch_d, iky * uch_d])
            all_bvs = periodic_interp2d(stack, self.ebdyc.interfaces_x_transf,
                                        self.ebdyc.interfaces_y_transf).real.cpu().numpy()
            bvs, bxs, bys = all_bvs[0], all_bvs[1], all_bvs[2]
        else:
            bvs = self.ebdyc.interpolate_grid_to_interface(uc)
            bxs = self.ebdyc.interpolate_grid_to_interface(np.asarray(self.dx(uc)))
            bys = self.ebdyc.interpolate_grid_to_interface(np.asarray(self.dy(uc)))
        uc = np.array(uc.cpu().numpy() if hasattr(uc, "cpu") else uc, copy=True)
        bvl, bxl, byl = self.ebdyc.v2l(bvs), self.ebdyc.v2l(bxs), self.ebdyc.v2l(bys)
        sigmag_list = []
        for helper, fr, bv, bx, by in zip(self.helpers, fr_list, bvl, bxl, byl):
            sigmag_list.append(helper(fr, bv, bx, by, **kwargs))
        self.iteration_counts = [helper.iterations_last_call for helper in self.helpers]
        sigmag = np.concatenate(sigmag_list)
        out = np.asarray(self.evaluate_to_grid_pnai(sigmag))
        gu, bus = self.ebdyc.divide_pnai(out)
        uc[self.ebdyc.phys_not_in_annulus] += gu
        urs = [helper.correct(bu) for helper, bu in zip(self.helpers, bus)]
        self.ebdyc.interpolate_radial_to_grid1(urs, uc)
        uc *= self.ebdyc.phys
        ue = EmbeddedFunction(self.ebdyc)
        ue.load_data(uc, urs)
        return ue

    def _define_layer_apply(self):
        self.Layer_Apply = self.helpers[0].Layer_Apply
