/* CPU sanitizer build: oracle/layer_oracle.c compiled with
 * -fsanitize=address,undefined and driven over the edge cases the parity tests use
 * (empty source / target sets, absent optional arrays, coincident pairs, exact-size heap
 * buffers so that any out-of-bounds index trips ASan).  Prints "ok" and exits 0. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void oracle_laplace_apply(int64_t ns, const double* sx, const double* sy, const double* q,
                          const double* nx, const double* ny, const double* tau, int64_t nt,
                          const double* tx, const double* ty, double* out, int skip_coincident);
void oracle_stokes_apply(int64_t ns, const double* sx, const double* sy, const double* fx,
                         const double* fy, const double* nx, const double* ny, const double* gx,
                         const double* gy, int64_t nt, const double* tx, const double* ty,
                         double* ou, double* ov, double* op, int skip_coincident);

static double* vec(int64_t n, double a, double b) {
    double* v = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));   /* exact size */
    for (int64_t i = 0; i < n; ++i) v[i] = a + b * (double)i / (double)(n > 1 ? n - 1 : 1);
    return v;
}

int main(void) {
    const int64_t sizes[][2] = {{0, 5}, {5, 0}, {1, 1}, {7, 3}, {129, 1025}, {64, 64}};
    for (size_t c = 0; c < sizeof(sizes) / sizeof(sizes[0]); ++c) {
        const int64_t ns = sizes[c][0], nt = sizes[c][1];
        double *sx = vec(ns, -1, 2), *sy = vec(ns, 0.5, -1), *q = vec(ns, 1, 1), *nx = vec(ns, 0, 1),
               *ny = vec(ns, 1, -1), *tau = vec(ns, -2, 3);
        /* the 64 x 64 case puts the targets ON the sources (skip_coincident path) */
        double *tx = (ns == 64 && nt == 64) ? vec(nt, -1, 2) : vec(nt, 2, 1),
               *ty = (ns == 64 && nt == 64) ? vec(nt, 0.5, -1) : vec(nt, 3, 1);
        double *o = vec(nt, 0, 0), *u = vec(nt, 0, 0), *v = vec(nt, 0, 0), *p = vec(nt, 0, 0);
        const int skip = (ns == 64 && nt == 64);
        oracle_laplace_apply(ns, sx, sy, q, NULL, NULL, NULL, nt, tx, ty, o, skip);
        oracle_laplace_apply(ns, sx, sy, NULL, nx, ny, tau, nt, tx, ty, o, skip);
        oracle_laplace_apply(ns, sx, sy, q, nx, ny, tau, nt, tx, ty, o, skip);
        oracle_stokes_apply(ns, sx, sy, q, tau, NULL, NULL, NULL, NULL, nt, tx, ty, u, v, p, skip);
        oracle_stokes_apply(ns, sx, sy, NULL, NULL, nx, ny, q, tau, nt, tx, ty, u, v, p, skip);
        oracle_stokes_apply(ns, sx, sy, q, tau, nx, ny, tau, q, nt, tx, ty, u, v, p, skip);
        for (int64_t i = 0; i < nt; ++i)
            if (!isfinite(o[i]) || !isfinite(u[i]) || !isfinite(v[i]) || !isfinite(p[i])) {
                fprintf(stderr, "non-finite output, case %zu\n", c);
                return 2;
            }
        free(sx); free(sy); free(q); free(nx); free(ny); free(tau);
        free(tx); free(ty); free(o); free(u); free(v); free(p);
    }
    puts("ok");
    return 0;
}
