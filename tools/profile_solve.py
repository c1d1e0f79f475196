import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
import numpy as np
import interior_poisson
err, scale, solver, ue, T = interior_poisson.run(nb=4096, M=20, Ns=[2048,2048], solver_tol=1e-12)
from ipde_amd.embedded_function import EmbeddedFunction
f = EmbeddedFunction(solver.ebdyc)
f.define_via_function(lambda x, y: np.sin(x)*np.cos(y))
solver(f, tol=1e-12, maxiter=100, restart=20)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): solver(f, tol=1e-12, maxiter=100, restart=20)
pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(35)
