"""The reference's own recorded results for examples/poisson_for_paper.py (:100-130: degrees of
freedom, maximum errors and GMRES iterations for nb = 200 .. 4000) as known answers for the whole
stack: geometry (arc-length reparametrisation, grid, inside / outside classification) on the CPU,
the full solve (annular solver, grid solve, layer potentials, QFS) on the GPU.  See
tests/paper_table_data.py for how the tables were matched to their parameters."""
import numpy as np
import pytest

from paper_table_data import REF, REF_REPARM_M2, run_case

# grids on which a grid point falls on the curve to rounding (the star's inward tips at |x| = 0.8 are
# grid lines when ng is a multiple of 6): its classification is a coin toss, the count may differ by one
ON_CURVE = {5, 7, 9, 16, 18}


@pytest.mark.parametrize("adj", [1, 2, 3, 4, 5, 6, 7, 8])
def test_degrees_of_freedom_are_the_reference_table(adj):
    r = run_case(adj, reparametrize=True, m_factor=2, geometry_only=True)
    if adj in ON_CURVE:
        assert abs(r["dof"] - REF["dof"][adj - 1]) <= 1
    else:
        assert r["dof"] == REF["dof"][adj - 1]


@pytest.mark.gpu
@pytest.mark.parametrize("adj", [2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20])
def test_solution_error_follows_the_reference_convergence_table(adj):
    """error <= the reference's recorded error (x 1.5 for the last digit of a rounded table entry
    and the on-curve grid point) down to its plateau; measured: 1.08e-12 vs 1.09e-12 at adj = 7,
    1.2e-13 vs 1.1e-13 at 8, 1e-14 on the plateau (profiles/r03_paper_table.jsonl)"""
    r = run_case(adj, reparametrize=True, m_factor=2)
    ref_err, ref_its = REF_REPARM_M2["errs"][adj - 1], REF_REPARM_M2["gmres"][adj - 1]
    assert abs(r["dof"] - REF["dof"][adj - 1]) <= (1 if adj in ON_CURVE else 0)
    assert r["err"] <= max(1.5 * ref_err, 1e-13), (r["err"], ref_err)
    # the annular GMRES needs no more iterations than the reference recorded (it counts ~4 more
    # throughout: scipy's callback convention)
    assert r["gmres"] <= ref_its, (r["gmres"], ref_its)
    # and the resolved cases sit ON the reference's curve, not just under it
    if 5 <= adj <= 8:
        assert r["err"] >= ref_err / 5.0, (r["err"], ref_err)
