// Ewald-type split of the grid evaluation (reference
// ipde/grid_evaluators/scalar_grid_evaluator.py:50-307):
//
//   sum_j q_j G(|x - y_j|)  =  sum_j q_j chi(r) G(r)            (compact: r <= R = sw h)
//                            + G * [ sum_j q_j rho(|. - y_j|) ]  (smooth, by FFT)
//   rho = L[(1 - chi) G] = 2 chi' G' + (chi'' + chi'/r) G,   L = -Lap  or  k^2 - Lap,
//
// rho is smooth and supported in r <= R, so the second term is a grid convolution
// (truncated-kernel FFT in free space, symbol division for the periodic sum) done by
// the caller with ipde_fourier_multiply.  This file is the first half: for every
// source the two radial functions are evaluated on the (2 sw + 3)^2 grid points
// around it and accumulated into the two grids (`ewald_local_freespace`, :189-229,
// and `ewald_local_periodic`, :131-178, of the reference).
//
// chi is a Kaiser-Bessel step (own choice: with beta = 1.6 sw the split reaches
// 1e-12 at sw = 20 and 7e-15 at sw = 24, where the reference's spline/function-
// generator mollifier stops at ~1e-10, :60-67).  chi, d chi/dr, d2 chi/dr2 come from
// piecewise polynomials (NI intervals in x = 1 - 2 r / R, degree DEG, built by the
// host from the closed forms) held in LDS; log / K0 / K1 are evaluated in full
// precision per point.
//
// One workgroup per source, threads over the stencil; fp64 hardware atomics
// (global_atomic_add_f64) accumulate — summation order, hence the last bit, is not
// reproducible.  Cost: ns (2 sw + 3)^2 evaluations (1e7 at ns = 4096, sw = 24): the
// FFTs of the second half dominate.
#include "ipde_common.h"
#include "bessel_device.h"

struct ipde_ewald {
    ipde_ctx* ctx;
    int kind;      // 0 Laplace, 1 modified Helmholtz
    double k, h, R;
    int sw, ni, deg;
    double* d_tab;  // [3][ni][deg+1] monomial coefficients in t in [-1,1]
    int* d_flag;
};

namespace {

constexpr int EW_NT = 256;

template <int KIND>
__global__ __launch_bounds__(EW_NT) void ewald_spread_kernel(
    const double* __restrict__ sx, const double* __restrict__ sy, const double* __restrict__ q,
    int64_t ns, double x0, double y0, double h, double k, int sw, double R,
    const double* __restrict__ gtab, int ni, int deg, int64_t nbx, int64_t nby, int64_t offx,
    int64_t offy, int periodic, double* __restrict__ u_loc, double* __restrict__ op,
    int* __restrict__ flag) {
    extern __shared__ double tab[];
    const int ntab = 3 * ni * (deg + 1);
    for (int i = threadIdx.x; i < ntab; i += EW_NT) tab[i] = gtab[i];
    __syncthreads();
    const int64_t j = blockIdx.x;
    if (j >= ns) return;
    const double xs = sx[j], ys = sy[j], qs = q[j];
    const int64_t cx = (int64_t)floor((xs - x0) / h), cy = (int64_t)floor((ys - y0) / h);
    const int W = 2 * sw + 3;
    const double R2 = R * R, iR2 = 2.0 / R, inv2pi = 0.15915494309189535;
    const int stride = deg + 1;
    for (int p = threadIdx.x; p < W * W; p += EW_NT) {
        const int a = p / W, b = p - a * W;
        const int64_t ix = cx - sw - 1 + a, iy = cy - sw - 1 + b;
        const double dx = fma((double)ix, h, x0) - xs, dy = fma((double)iy, h, y0) - ys;
        const double d2 = fma(dx, dx, dy * dy);
        if (d2 > R2 || d2 == 0.0) continue;
        int64_t gx = ix + offx, gy = iy + offy;
        if (periodic) {
            gx %= nbx;
            gy %= nby;
            if (gx < 0) gx += nbx;
            if (gy < 0) gy += nby;
        } else if (gx < 0 || gy < 0 || gx >= nbx || gy >= nby) {
            atomicOr(flag, 1);
            continue;
        }
        const double r = sqrt(d2);
        // x = 1 - 2 r / R in [-1, 1]  ->  interval i, local t in [-1, 1]
        double fi = (1.0 - 0.5 * iR2 * r) * (double)ni;   // = (x + 1)/2 * ni
        int i = (int)fi;
        i = i < 0 ? 0 : (i >= ni ? ni - 1 : i);
        const double t = 2.0 * (fi - (double)i) - 1.0;
        const double* c0 = tab + (size_t)i * stride;
        const double* c1 = c0 + (size_t)ni * stride;
        const double* c2 = c1 + (size_t)ni * stride;
        double chi = c0[deg], chi_r = c1[deg], chi_rr = c2[deg];
        for (int m = deg - 1; m >= 0; --m) {
            chi = fma(chi, t, c0[m]);
            chi_r = fma(chi_r, t, c1[m]);
            chi_rr = fma(chi_rr, t, c2[m]);
        }
        double G, Gp;
        if (KIND == 0) {
            G = -0.5 * inv2pi * log(d2);
            Gp = -inv2pi / r;
        } else {
            double k0v, k1x;
            bessel_k01<3>(k * k * d2, k0v, k1x);
            G = inv2pi * k0v;
            Gp = -inv2pi * k * k * r * k1x;
        }
        const double loc = chi * G;
        // rho -> 0 as r -> 0, but chi' is only ~1e-16 there (the Kaiser-Bessel bump ends on
        // 1/I0(beta), not on 0): closer than 1e-6 R the 1/r factors would turn that into O(1)
        const double rho = (r > 1e-6 * R) ? fma(2.0 * chi_r, Gp, (chi_rr + chi_r / r) * G) : 0.0;
        const int64_t idx = gx * nby + gy;
        unsafeAtomicAdd(&u_loc[idx], qs * loc);
        unsafeAtomicAdd(&op[idx], qs * rho);
    }
}

// Stokeslet (with pressure) through the Laplace split — see oracle/ewald.py and
// ipde_amd/grid_evaluators/ewald.py:StokesFreespaceEwald for the identity.  Per pair the
// near part is complete here,
//   u_i += (chi G f_i - (chi G)' r_i (r.f)/r) / 2,    p -= (chi G)' (r.f)/r,
// and six densities rho q, q in {f_x, f_y, y_x f_x, y_x f_y, y_y f_x, y_y f_y} (y relative
// to the grid centre), are spread for the far field.
__global__ __launch_bounds__(EW_NT) void ewald_spread_stokes_kernel(
    const double* __restrict__ sx, const double* __restrict__ sy, const double* __restrict__ fx,
    const double* __restrict__ fy, int64_t ns, double x0, double y0, double h, int sw, double R,
    const double* __restrict__ gtab, int ni, int deg, int64_t nbx, int64_t nby, int64_t offx,
    int64_t offy, double cx, double cy, double* __restrict__ loc3, double* __restrict__ op6,
    int* __restrict__ flag) {
    extern __shared__ double tab[];
    const int ntab = 3 * ni * (deg + 1);
    for (int i = threadIdx.x; i < ntab; i += EW_NT) tab[i] = gtab[i];
    __syncthreads();
    const int64_t j = blockIdx.x;
    if (j >= ns) return;
    const double xs = sx[j], ys = sy[j], fxs = fx[j], fys = fy[j];
    const double yx = xs - cx, yy = ys - cy;
    const int64_t ccx = (int64_t)floor((xs - x0) / h), ccy = (int64_t)floor((ys - y0) / h);
    const int W = 2 * sw + 3;
    const double R2 = R * R, iR2 = 2.0 / R, inv2pi = 0.15915494309189535;
    const int stride = deg + 1;
    const size_t plane = (size_t)nbx * nby;
    for (int p = threadIdx.x; p < W * W; p += EW_NT) {
        const int a = p / W, b = p - a * W;
        const int64_t ix = ccx - sw - 1 + a, iy = ccy - sw - 1 + b;
        const double rx = fma((double)ix, h, x0) - xs, ry = fma((double)iy, h, y0) - ys;
        const double d2 = fma(rx, rx, ry * ry);
        if (d2 > R2 || d2 == 0.0) continue;
        const int64_t gx = ix + offx, gy = iy + offy;
        if (gx < 0 || gy < 0 || gx >= nbx || gy >= nby) {
            atomicOr(flag, 1);
            continue;
        }
        const double r = sqrt(d2);
        double fi = (1.0 - 0.5 * iR2 * r) * (double)ni;
        int i = (int)fi;
        i = i < 0 ? 0 : (i >= ni ? ni - 1 : i);
        const double t = 2.0 * (fi - (double)i) - 1.0;
        const double* c0 = tab + (size_t)i * stride;
        const double* c1 = c0 + (size_t)ni * stride;
        const double* c2 = c1 + (size_t)ni * stride;
        double chi = c0[deg], chi_r = c1[deg], chi_rr = c2[deg];
        for (int m = deg - 1; m >= 0; --m) {
            chi = fma(chi, t, c0[m]);
            chi_r = fma(chi_r, t, c1[m]);
            chi_rr = fma(chi_rr, t, c2[m]);
        }
        const double G = -0.5 * inv2pi * log(d2), Gp = -inv2pi / r;
        const double lg = chi * G, dlg = fma(chi_r, G, chi * Gp);
        const double rho = (r > 1e-6 * R) ? fma(2.0 * chi_r, Gp, (chi_rr + chi_r / r) * G) : 0.0;
        const double rf = (rx * fxs + ry * fys) / r;
        const size_t idx = (size_t)gx * nby + gy;
        unsafeAtomicAdd(&loc3[idx], 0.5 * (lg * fxs - dlg * rx * rf));
        unsafeAtomicAdd(&loc3[plane + idx], 0.5 * (lg * fys - dlg * ry * rf));
        unsafeAtomicAdd(&loc3[2 * plane + idx], -dlg * rf);
        unsafeAtomicAdd(&op6[idx], rho * fxs);
        unsafeAtomicAdd(&op6[plane + idx], rho * fys);
        unsafeAtomicAdd(&op6[2 * plane + idx], rho * yx * fxs);
        unsafeAtomicAdd(&op6[3 * plane + idx], rho * yx * fys);
        unsafeAtomicAdd(&op6[4 * plane + idx], rho * yy * fxs);
        unsafeAtomicAdd(&op6[5 * plane + idx], rho * yy * fys);
    }
}

}  // namespace

extern "C" int ipde_ewald_create(ipde_ctx* ctx, int kind, double k, double h, int sw,
                                 const double* mol_tab, int ni, int deg, ipde_ewald** out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, out && mol_tab);
    IPDE_CHECK_ARG(ctx, kind == 0 || kind == 1);
    IPDE_CHECK_ARG(ctx, h > 0 && sw >= 1 && sw <= 256 && ni >= 1 && deg >= 1);
    IPDE_CHECK_ARG(ctx, kind == 0 || k > 0);
    IPDE_CHECK_ARG(ctx, (size_t)3 * ni * (deg + 1) * sizeof(double) <= 60000);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ipde_ewald* e = new ipde_ewald{};
    e->ctx = ctx;
    e->kind = kind;
    e->k = k;
    e->h = h;
    e->sw = sw;
    e->R = sw * h;
    e->ni = ni;
    e->deg = deg;
    size_t bytes = (size_t)3 * ni * (deg + 1) * sizeof(double);
    if (hipMalloc(&e->d_tab, bytes) != hipSuccess || hipMalloc(&e->d_flag, sizeof(int)) != hipSuccess) {
        delete e;
        IPDE_SET_ERR(ctx, "ipde_ewald_create: out of device memory");
        return IPDE_ERR_ALLOC;
    }
    IPDE_HIP_CHECK(ctx, hipMemcpy(e->d_tab, mol_tab, bytes, hipMemcpyHostToDevice));
    IPDE_HIP_CHECK(ctx, hipMemset(e->d_flag, 0, sizeof(int)));
    IPDE_HIP_CHECK(ctx, ipde_bessel_upload());
    *out = e;
    return IPDE_OK;
}

extern "C" int ipde_ewald_destroy(ipde_ewald* e) {
    if (!e) return IPDE_ERR_INVALID;   // like every other destroy entry point
    hipFree(e->d_tab);
    hipFree(e->d_flag);
    delete e;
    return IPDE_OK;
}

extern "C" int ipde_ewald_spread(ipde_ewald* e, int loc, int64_t ns, const double* sx,
                                 const double* sy, const double* q, double x0, double y0,
                                 int64_t nbx, int64_t nby, int64_t offx, int64_t offy, int periodic,
                                 double* u_loc, double* op) {
    if (!e) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = e->ctx;
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, ns >= 0 && nbx > 0 && nby > 0 && u_loc && op);
    IPDE_CHECK_ARG(ctx, !periodic || (nbx >= 2 * e->sw + 3 && nby >= 2 * e->sw + 3));
    if (ns == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, sx && sy && q);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double *d_sx, *d_sy, *d_q;
    IPDE_TRY(ipde_stage_in(ctx, loc, 0, sx, ns, &d_sx));
    IPDE_TRY(ipde_stage_in(ctx, loc, 1, sy, ns, &d_sy));
    IPDE_TRY(ipde_stage_in(ctx, loc, 2, q, ns, &d_q));
    const size_t lds = (size_t)3 * e->ni * (e->deg + 1) * sizeof(double);
    dim3 grid((unsigned)ns);
    if (e->kind == 0)
        hipLaunchKernelGGL(ewald_spread_kernel<0>, grid, dim3(EW_NT), lds, ctx->stream, d_sx, d_sy, d_q,
                           ns, x0, y0, e->h, e->k, e->sw, e->R, e->d_tab, e->ni, e->deg, nbx, nby,
                           offx, offy, periodic, u_loc, op, e->d_flag);
    else
        hipLaunchKernelGGL(ewald_spread_kernel<1>, grid, dim3(EW_NT), lds, ctx->stream, d_sx, d_sy, d_q,
                           ns, x0, y0, e->h, e->k, e->sw, e->R, e->d_tab, e->ni, e->deg, nbx, nby,
                           offx, offy, periodic, u_loc, op, e->d_flag);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    if (!periodic) {
        int flag = 0;
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(&flag, e->d_flag, sizeof(int), hipMemcpyDeviceToHost,
                                           ctx->stream));
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (flag) {
            IPDE_HIP_CHECK(ctx, hipMemsetAsync(e->d_flag, 0, sizeof(int), ctx->stream));
            IPDE_SET_ERR(ctx, "ipde_ewald_spread: a source's stencil leaves the padded grid");
            return IPDE_ERR_INVALID;
        }
    }
    return IPDE_OK;
}

extern "C" int ipde_ewald_spread_stokes(ipde_ewald* e, int loc, int64_t ns, const double* sx,
                                        const double* sy, const double* fx, const double* fy,
                                        double x0, double y0, double cx, double cy, int64_t nbx,
                                        int64_t nby, int64_t offx, int64_t offy, double* loc3,
                                        double* op6) {
    if (!e) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = e->ctx;
    IPDE_CHECK_ARG(ctx, e->kind == 0);   // built on the Laplace cut-off tables
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, ns >= 0 && nbx > 0 && nby > 0 && loc3 && op6);
    if (ns == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, sx && sy && fx && fy);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double *d_sx, *d_sy, *d_fx, *d_fy;
    IPDE_TRY(ipde_stage_in(ctx, loc, 0, sx, ns, &d_sx));
    IPDE_TRY(ipde_stage_in(ctx, loc, 1, sy, ns, &d_sy));
    IPDE_TRY(ipde_stage_in(ctx, loc, 2, fx, ns, &d_fx));
    IPDE_TRY(ipde_stage_in(ctx, loc, 3, fy, ns, &d_fy));
    const size_t lds = (size_t)3 * e->ni * (e->deg + 1) * sizeof(double);
    hipLaunchKernelGGL(ewald_spread_stokes_kernel, dim3((unsigned)ns), dim3(EW_NT), lds, ctx->stream,
                       d_sx, d_sy, d_fx, d_fy, ns, x0, y0, e->h, e->sw, e->R, e->d_tab, e->ni, e->deg,
                       nbx, nby, offx, offy, cx, cy, loc3, op6, e->d_flag);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    int flag = 0;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(&flag, e->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) {
        IPDE_HIP_CHECK(ctx, hipMemsetAsync(e->d_flag, 0, sizeof(int), ctx->stream));
        IPDE_SET_ERR(ctx, "ipde_ewald_spread_stokes: a source's stencil leaves the padded grid");
        return IPDE_ERR_INVALID;
    }
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// Truncated spectral Green's function on a quadrant of wavenumbers (set-up of the split evaluator;
// reference ipde/grid_evaluators/laplace_grid_evaluator.py:21-33 and
// modified_helmholtz_grid_evaluator.py:14-17): for k = |(kx_i, ky_j)|
//   Laplace              (1 - J0(L k)) / k^2 - L ln L J1(L k) / k      (k = 0: its limit)
//   modified Helmholtz   (1 + L k J1(L k) K0(L kap) - L kap J0(L k) K1(L kap)) / (k^2 + kap^2)
// J0 / J1 from the host's piecewise Chebyshev table (degree `deg`, pieces of width `w`, Clenshaw).
// One kernel instead of a dozen torch operations: in a fresh process each of those loaded a code
// object of its own (0.43 s of a 0.47 s set-up step at 2048^2).
namespace {

__global__ __launch_bounds__(256) void trunc_sgf_kernel(long long nx, long long ny, const double* __restrict__ kx,
                                                        const double* __restrict__ ky, double L, double lnL,
                                                        int helmholtz, double kap, double LK0, double LkapK1,
                                                        const double* __restrict__ tab, long long ni, int deg,
                                                        double inv_w, double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny) return;
    const double k = hypot(kx[i / ny], ky[i % ny]);
    const double s = L * k * inv_w;
    long long id = (long long)floor(s);
    id = id < 0 ? 0 : (id > ni - 1 ? ni - 1 : id);
    const double t = 2.0 * (s - (double)id) - 1.0;
    double J[2];
    for (int f = 0; f < 2; ++f) {
        const double* c = tab + ((long long)f * ni + id) * (deg + 1);
        double b1 = c[deg], b2 = 0.0;
        for (int d = deg - 1; d > 0; --d) {
            const double b = c[d] + 2.0 * t * b1 - b2;
            b2 = b1;
            b1 = b;
        }
        J[f] = c[0] + t * b1 - b2;
    }
    double v;
    if (helmholtz)
        v = (1.0 + k * J[1] * LK0 - LkapK1 * J[0]) / (k * k + kap * kap);
    else if (k == 0.0)
        v = -L * L * lnL + L * L * (1.0 + 2.0 * lnL) / 4.0;
    else
        v = (1.0 - J[0]) / (k * k) - (L * lnL) * J[1] / k;
    out[i] = v;
}

}  // namespace

extern "C" int ipde_trunc_sgf_quadrant(ipde_ctx* ctx, int64_t nx, int64_t ny, const double* kx, const double* ky,
                                       double L, int helmholtz, double kap, double K0, double K1,
                                       const double* j01_tab, int64_t ni, int deg, double piece_width, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, nx >= 0 && ny >= 0 && L > 0.0 && ni > 0 && deg >= 1 && deg <= 32 && piece_width > 0.0);
    IPDE_CHECK_ARG(ctx, !helmholtz || kap > 0.0);
    if (nx == 0 || ny == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, kx && ky && j01_tab && out && nx * ny < (1ll << 40));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const long long n = nx * ny;
    hipLaunchKernelGGL(trunc_sgf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (long long)nx, (long long)ny, kx, ky, L, log(L), helmholtz, kap, L * K0, L * kap * K1,
                       j01_tab, (long long)ni, deg, 1.0 / piece_width, out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}
