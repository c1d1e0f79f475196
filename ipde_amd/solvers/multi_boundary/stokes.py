"""StokesSolver — mirrors ipde/solvers/multi_boundary/stokes.py:6-50."""
import numpy as np
import torch

from .vector import VectorSolver
from ..internals.stokes import StokesHelper


class StokesSolver(VectorSolver):
    def __init__(self, ebdyc, solver_type='spectral', helpers=None, grid_backend=None):
        if not ebdyc.bumpy_readied:
            raise Exception('Stokes solver requires embedded boundary collection with a bump function.')
        super().__init__(ebdyc, solver_type, helpers, grid_backend=grid_backend)

    def _get_helper_compatability(self, ebdy, helper):
        """0: helper is of no use; 1: its annular solver can be reused; 2: reuse as is
        (reference :13-26)."""
        if helper is None:
            return 0
        if ebdy.bdy.N != helper.ebdy.bdy.N:
            return 0
        if helper.ebdy is not ebdy:
            return 1
        return 2

    def _get_helper(self, ebdy, helper):
        c = self._get_helper_compatability(ebdy, helper)
        if c == 0:
            return StokesHelper(ebdy, private_ctx=self.ebdyc.N > 1)
        elif c == 1:
            return StokesHelper(ebdy, helper.annular_solver)
        return helper

    def _grid_solve(self, fuc, fvc):
        """Demean both forcings with the bump (ebdy_collection.py:808-810), then the periodic
        Stokes solve  p^ = ilap (ikx fu^ + iky fv^),  u^ = ilap (ikx p^ - fu^),
        v^ = ilap (iky p^ - fv^)  (reference :34-45) in one library call."""
        bumpy = self.ebdyc.bumpy
        if getattr(self, "_bumpy_src", None) is not bumpy:
            self._bumpy_d = torch.as_tensor(np.ascontiguousarray(bumpy), device=fuc.device)
            self._bumpy_src = bumpy
        dA = self.grid.xh * self.grid.yh
        fuc = fuc - (fuc.sum() * dA) * self._bumpy_d
        fvc = fvc - (fvc.sum() * dA) * self._bumpy_d
        return self.plan.stokes_solve(fuc.contiguous(), fvc.contiguous())

    def _get_specific_operators(self):
        self.lap = -self.kx * self.kx - self.ky * self.ky
        with np.errstate(divide='ignore'):
            self.ilap = 1.0 / self.lap
        self.ilap[0, 0] = 0.0
