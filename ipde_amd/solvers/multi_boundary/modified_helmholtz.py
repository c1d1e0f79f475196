"""ModifiedHelmholtzSolver — mirrors ipde/solvers/multi_boundary/modified_helmholtz.py:8-67."""

from .scalar import ScalarSolver
from ..internals.modified_helmholtz import ModifiedHelmholtzHelper
from ...grid_evaluators.modified_helmholtz_grid_evaluator import (
    ModifiedHelmholtzFreespaceGridEvaluator, ModifiedHelmholtzGridBackend)


class ModifiedHelmholtzSolver(ScalarSolver):
    PATCH_TARGETS = True      # the grid_pnai list planned into patches ...
    FAR_EXPANSION = True      # ... and summed with far sources in local expansions (ipde_modhelm_apply_patches_far)

    def __init__(self, ebdyc, k, solver_type='spectral', helpers=None, grid_backend=None,
                 source_upsample_factor=1.0):
        self.k = k
        self.source_upsample_factor = source_upsample_factor
        # the kernel's K0 / K1 tables (one set per device: ~0.2 s of long-double Bessel evaluations on the
        # host at the first modified Helmholtz sum of a process) under the rest of the set-up instead of in
        # front of the first solve: a one-pair sum on a context of its own, in a thread joined below
        import threading
        warm = threading.Thread(target=self._build_kernel_tables, name="ipde-ktab", daemon=True)
        warm.start()
        try:
            super().__init__(ebdyc, solver_type, helpers, grid_backend)
        finally:
            warm.join()

    @staticmethod
    def _build_kernel_tables():
        try:
            import numpy as np
            import torch
            from ...device import private_context
            from ...layer_potentials import modified_helmholtz_apply
            if not torch.cuda.is_available():
                return
            ctx = private_context()
            one = np.ones(1)
            modified_helmholtz_apply(one * 0.0, one * 0.0, one, one, 1.0, w_sigma=one, ctx=ctx)
            ctx.sync()
        except Exception:       # (the first sum builds them then, as before)
            pass

    def _get_helper_combatibility(self, ebdy, helper):
        """0: start over; 1: the annular solver can be reused; 2: reuse the helper"""
        if helper is None:
            return 0
        if self.k != helper.k:
            return 0
        if ebdy.bdy.N != helper.ebdy.bdy.N:
            return 0
        if self.source_upsample_factor != helper.source_upsample_factor:
            return 0
        if helper.ebdy is not ebdy:
            return 1
        return 2

    def _get_helper(self, ebdy, helper):
        c = self._get_helper_combatibility(ebdy, helper)
        if c == 0:
            return ModifiedHelmholtzHelper(ebdy, k=self.k,
                                           source_upsample_factor=self.source_upsample_factor,
                                           grid_backend=self.grid_backend, private_ctx=self.ebdyc.N > 1)
        elif c == 1:
            return ModifiedHelmholtzHelper(ebdy, helper, k=self.k,
                                           source_upsample_factor=self.source_upsample_factor,
                                           grid_backend=self.grid_backend)
        return helper

    def _grid_solve(self, fc):
        """fc: device (Nx, Ny) -> (uch, uc) on the device"""
        if self._fast_interp:     # the spectrum stays inside the plan (plan.interp_gradient)
            return None, self.plan.modhelm_solve(fc.contiguous(), self.k)
        uch, uc = self.plan.modhelm_solve(fc.contiguous(), self.k, want_uhat=True)
        return uch, uc

    def _get_specific_operators(self):
        self.lap = -self.kx * self.kx - self.ky * self.ky
        self.helm = self.k ** 2 - self.lap
        self.ihelm = 1.0 / self.helm

    def _define_grid_evaluator(self):
        if self.grid_backend == 'ewald':
            self.grid_backend = ModifiedHelmholtzGridBackend(self.grid.xh, 24, self.k, method='ewald')
        if type(self.grid_backend) in [ModifiedHelmholtzGridBackend,
                                       ModifiedHelmholtzFreespaceGridEvaluator]:
            if type(self.grid_backend) == ModifiedHelmholtzGridBackend:
                self.ewald_evaluator = ModifiedHelmholtzFreespaceGridEvaluator(
                    self.grid_backend, self.grid.xv, self.grid.yv,
                    allow_rectangular=self.grid_backend.method == 'ewald')
            else:
                self.ewald_evaluator = self.grid_backend

            def evaluator(ch):
                if type(ch).__module__.startswith('torch'):      # device flow: sources resident too
                    from ...layer_potentials import _device_source
                    from ...device import get_context
                    import torch
                    d = _device_source(self.grid_sources, get_context())
                    return self.ewald_evaluator(torch.stack([d.x, d.y]), ch * d.weights, device_result=True)
                return self.ewald_evaluator(self.grid_sources.get_stacked_boundary(),
                                            ch * self.grid_sources.weights, device_result=True)
            self.Grid_Evaluator = evaluator
            self.split_grid_evaluation = True
        else:
            self.Grid_Evaluator = self._pnai_evaluator()
            self.split_grid_evaluation = False
