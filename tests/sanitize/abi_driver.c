/* Host side of libipde_hip.so under ASan + UBSan (no GPU needed): the argument-checking and
 * failure paths every entry point reaches before it touches the device.
 *   - ipde_ctx_create on a machine without a usable gfx950 device (or with one): the
 *     failure path goes through ipde_ctx_destroy; on success the context is destroyed again
 *   - every entry point with a NULL context / plan / handle returns IPDE_ERR_INVALID
 * Prints "ok" and exits 0; a sanitizer report aborts. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ipde_hip.h"

#define EXPECT_INVALID(call)                                             \
    do {                                                                 \
        int _s = (call);                                                 \
        if (_s != IPDE_ERR_INVALID) {                                    \
            fprintf(stderr, "%s -> %d, expected IPDE_ERR_INVALID\n", #call, _s); \
            return 3;                                                    \
        }                                                                \
    } while (0)

int main(void) {
    ipde_ctx* ctx = NULL;
    int s = ipde_ctx_create(0, &ctx);
    if (s == IPDE_OK) {
        int v = -1;
        if (ipde_ctx_get_option(ctx, "laplace_variant", &v) != IPDE_OK || v != 9) return 4;
        if (ipde_ctx_set_option(ctx, "no_such_option", 1) != IPDE_ERR_INVALID) return 5;
        if (!strstr(ipde_last_error(ctx), "no_such_option")) return 6;
        if (ipde_ctx_destroy(ctx) != IPDE_OK) return 7;
    } else if (ctx != NULL) {
        return 8;                       /* a failed create must not hand out a context */
    }
    if (ipde_ctx_create(0, NULL) != IPDE_ERR_INVALID) return 9;
    double x = 0.0;
    int iters = 0;
    EXPECT_INVALID(ipde_ctx_destroy(NULL));
    EXPECT_INVALID(ipde_ctx_sync(NULL));
    EXPECT_INVALID(ipde_ctx_set_option(NULL, "laplace_variant", 1));
    EXPECT_INVALID(ipde_ctx_get_option(NULL, "laplace_variant", &iters));
    EXPECT_INVALID(ipde_ctx_enable_timing(NULL, 1));
    EXPECT_INVALID(ipde_ctx_last_kernel_ms(NULL, &x));
    EXPECT_INVALID(ipde_laplace_apply(NULL, IPDE_HOST, 1, &x, &x, &x, NULL, NULL, NULL, 1, &x, &x, &x, 0));
    { int32_t pi = 0;
      EXPECT_INVALID(ipde_laplace_apply_patches(NULL, 1, &x, &x, &x, NULL, NULL, NULL, 1, &x, &pi, &x)); }
    EXPECT_INVALID(ipde_modhelm_apply(NULL, IPDE_HOST, 1.0, 1, &x, &x, &x, NULL, NULL, NULL, 1, &x, &x, &x, 0));
    EXPECT_INVALID(ipde_stokes_apply(NULL, IPDE_HOST, 1, &x, &x, &x, &x, NULL, NULL, NULL, NULL, 1, &x, &x,
                                     &x, &x, &x, 0));
    EXPECT_INVALID(ipde_fft_plan2d_create(NULL, 8, 8, 1.0, 1.0, NULL));
    EXPECT_INVALID(ipde_fft_plan2d_destroy(NULL));
    EXPECT_INVALID(ipde_poisson_grid_solve(NULL, IPDE_HOST, &x, &x, NULL));
    EXPECT_INVALID(ipde_modhelm_grid_solve(NULL, IPDE_HOST, 1.0, &x, &x, NULL));
    EXPECT_INVALID(ipde_stokes_grid_solve(NULL, IPDE_HOST, &x, &x, &x, &x, &x));
    EXPECT_INVALID(ipde_fourier_deriv(NULL, IPDE_HOST, &x, 0, &x));
    EXPECT_INVALID(ipde_fourier_multiply(NULL, IPDE_HOST, &x, &x, &x));
    EXPECT_INVALID(ipde_fd4(NULL, IPDE_HOST, 8, 8, 1.0, 0, 0, &x, &x));
    EXPECT_INVALID(ipde_fft1_prepare(NULL, 1, 8));
    EXPECT_INVALID(ipde_fft1_c2c(NULL, IPDE_HOST, 1, 8, -1, &x, &x));
    EXPECT_INVALID(ipde_annular_scalar_destroy(NULL));
    EXPECT_INVALID(ipde_annular_stokes_destroy(NULL));
    EXPECT_INVALID(ipde_annular_scalar_solve(NULL, IPDE_HOST, &x, &x, &x, 0, 1e-12, 10, 5, &x, &iters, &x));
    EXPECT_INVALID(ipde_ewald_destroy(NULL));
    EXPECT_INVALID(ipde_dense_lu_solve(NULL, 1, &x, NULL, &x, &x));
    {
        int perm = 0;
        int64_t i64 = 0;
        double* outs[1] = {&x};
        EXPECT_INVALID(ipde_dense_lu_factor(NULL, 128, &x, &perm));
        EXPECT_INVALID(ipde_dense_gemv(NULL, 1, 1, &x, &x, &x, 0));
        EXPECT_INVALID(ipde_radial_to_grid(NULL, IPDE_HOST, 1, 4, 32, &x, &x, 1, &x, &x, NULL, outs));
        EXPECT_INVALID(ipde_grid_scatter(NULL, 1, 1, &i64, &x, NULL, &x));
        EXPECT_INVALID(ipde_grid_add_at(NULL, 1, &i64, &x, &x));
        EXPECT_INVALID(ipde_grid_gather(NULL, 1, &i64, &x, &x));
        EXPECT_INVALID(ipde_scalar_interface_jumps(NULL, 4, 8, &x, &x, &x, &x, 1.0, &x, &x));
        EXPECT_INVALID(ipde_stokes_rotate(NULL, IPDE_HOST, 4, 8, &x, &x, &x, 1, &x, &x));
        EXPECT_INVALID(ipde_stokes_interface_jumps(NULL, 4, 8, &x, &x, &x, &x, &x, &x, &x, &x, &x, &x, 1.0, &x, &x, &x,
                                                   &x));
    }
    puts("ok");
    return 0;
}
