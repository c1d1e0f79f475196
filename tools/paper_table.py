"""The flow of the reference's examples/poisson_for_paper.py (:26-92) at the sizes of its recorded
table (:118-130, not reparametrised, slepian_r = 1.5 M): prints nb, dof, error / uscale and GMRES
iterations next to the reference's own numbers.  Written against the reference's import names
through ipde_amd.compat."""
import json
import sys
import os
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from paper_table_data import REF, REF_REPARM, run_case   # noqa: E402

if __name__ == "__main__":
    from paper_table_data import REF_REPARM_M2
    adjs = [int(a) for a in sys.argv[1:]] or list(range(1, 21))
    # (factor, reparametrised): the run the reference's dof / M2 tables belong to first
    for m_factor, reparm, T in ((2, True, REF_REPARM_M2),):
        for adj in adjs:
            t0 = time.time()
            r = run_case(adj, reparametrize=reparm, m_factor=m_factor)
            r["wall_s"] = time.time() - t0
            r["ref_err"] = T["errs"][adj - 1]
            r["ref_gmres"] = T["gmres"][adj - 1]
            r["ref_dof"] = REF["dof"][adj - 1]
            print(json.dumps(r), flush=True)
