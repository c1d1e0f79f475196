/*
 * TEST INFRASTRUCTURE ONLY — plain C (OpenMP) restatement of the Laplace and
 * Stokes layer-potential sums, same formulas and conventions as
 * oracle/layer_potentials.py (see its header for the reference citations:
 * ipde/grid_evaluators/laplace_grid_evaluator.py:8-12,
 * ipde/solvers/internals/poisson.py:27-36,
 * ipde/solvers/internals/stokes_save.py:29-81).  Parity status as in that header:
 * Laplace SLP and the Stokes pressures are PINNED by reference-computed fixtures
 * (tests/golden/layer_kernels.npz), Laplace DLP and the Stokes velocities are UNPINNED
 * upstream (pybie2d / pyfmmlib2d are not in the reference tree) and held by analytic
 * identities and derivative relations to the pinned kernels.
 *
 * This is the "port" CPU baseline timed by bench.py and the checker for the
 * larger parity tests.  The product never links or calls it.
 *
 * Densities arrive already multiplied by the quadrature weights.
 * Build: see oracle/Makefile (gcc -O3 -march=native -fopenmp, no fast-math).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* out_i = sum_j [ -(1/4pi) log(d2) q_j + (1/2pi) (a_j . d)/d2 ],  a = n * w_tau */
void oracle_laplace_apply(int64_t ns, const double* sx, const double* sy, const double* q,
                          const double* nx, const double* ny, const double* tau, int64_t nt,
                          const double* tx, const double* ty, double* out, int skip_coincident) {
    const double cs = -0.25 / M_PI, cd = 0.5 / M_PI;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nt; ++i) {
        const double x = tx[i], y = ty[i];
        double as = 0.0, ad = 0.0;
        for (int64_t j = 0; j < ns; ++j) {
            double dx = x - sx[j], dy = y - sy[j];
            double d2 = dx * dx + dy * dy;
            if (skip_coincident && d2 == 0.0) continue;
            if (q) as += q[j] * log(d2);
            if (tau) ad += tau[j] * (nx[j] * dx + ny[j] * dy) / d2;
        }
        out[i] = cs * as + cd * ad;
    }
}

/* Stokeslet f (fx,fy) + stresslet g (gx,gy) with normals n; mu = 1 */
void oracle_stokes_apply(int64_t ns, const double* sx, const double* sy, const double* fx,
                         const double* fy, const double* nx, const double* ny, const double* gx,
                         const double* gy, int64_t nt, const double* tx, const double* ty,
                         double* ou, double* ov, double* op, int skip_coincident) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nt; ++i) {
        const double x = tx[i], y = ty[i];
        double us = 0, vs = 0, ps = 0, ud = 0, vd = 0, pd = 0;
        for (int64_t j = 0; j < ns; ++j) {
            double dx = x - sx[j], dy = y - sy[j];
            double d2 = dx * dx + dy * dy;
            if (skip_coincident && d2 == 0.0) continue;
            double ir2 = 1.0 / d2;
            if (fx) {
                double mlogr = -0.5 * log(d2);
                double df = (dx * fx[j] + dy * fy[j]) * ir2;
                us += mlogr * fx[j] + df * dx;
                vs += mlogr * fy[j] + df * dy;
                ps += df;
            }
            if (gx) {
                double dn = dx * nx[j] + dy * ny[j];
                double dg = dx * gx[j] + dy * gy[j];
                double ng = nx[j] * gx[j] + ny[j] * gy[j];
                double w = dn * dg * ir2 * ir2;
                ud += w * dx;
                vd += w * dy;
                pd += -ng * ir2 + 2.0 * w;
            }
        }
        ou[i] = us * (0.25 / M_PI) + ud / M_PI;
        ov[i] = vs * (0.25 / M_PI) + vd / M_PI;
        if (op) op[i] = ps * (0.5 / M_PI) + pd / M_PI;
    }
}
