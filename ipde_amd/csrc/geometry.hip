// Closest-point ("local") coordinates of grid points near a closed curve (SURVEY §8f
// rank 3: the grid-point classification either side of the hot path; reference
// ipde/embedded_boundary.py:185-214 calls the third-party near_finder for it).
//
// For p within `width` of X(t):  p = X(t) + r n(t).  Newton's method on
// g(t) = (p - X(t)) . X'(t), one thread per point.  X, X', X'' come from the 8x
// trigonometrically upsampled samples tab[3][nf] (complex: x + i y) by 12-point barycentric
// Lagrange interpolation — the iteration of ipde_amd/near.py, statement for statement
// (stencil position, exact-hit rule, Gauss-Newton fallback, step limiter), so host and
// device agree to rounding.  The table (3 x nf x 16 B, 4.7 MB at nf = 99 200) is L2
// resident; a point costs ~7 iterations x 36 complex loads.
#include "ipde_common.h"

namespace {

constexpr int NL = 12;   // stencil width (near.py _NL)

struct CurveArgs {
    const double2* tab;   // [3][nf]
    int nf;
    double hf;            // 2 pi / nf
    double w[NL];         // barycentric weights of NL equispaced nodes
};

__device__ __forceinline__ void curve_eval(const CurveArgs& c, double t, double2& X, double2& Xp,
                                           double2& Xpp) {
    const double s = t / c.hf;
    const double fl = floor(s);
    const long long i0 = (long long)fl - (NL / 2 - 1);
    const double u = s - (double)i0;
    long long base = i0 % c.nf;
    if (base < 0) base += c.nf;
    double wt[NL];
    double wsum = 0.0;
    int hit = -1;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const double d = u - (double)j;
        if (fabs(d) < 1e-14) hit = j;
        wt[j] = c.w[j] / d;
        wsum += wt[j];
    }
    X = Xp = Xpp = make_double2(0.0, 0.0);
    if (hit >= 0) {
        int idx = (int)((base + hit) % c.nf);
        X = c.tab[idx];
        Xp = c.tab[c.nf + idx];
        Xpp = c.tab[2 * c.nf + idx];
        return;
    }
    const double inv = 1.0 / wsum;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        int idx = (int)(base + j);
        if (idx >= c.nf) idx -= c.nf;
        const double a = wt[j] * inv;
        const double2 v0 = c.tab[idx], v1 = c.tab[c.nf + idx], v2 = c.tab[2 * c.nf + idx];
        X.x = fma(a, v0.x, X.x);     X.y = fma(a, v0.y, X.y);
        Xp.x = fma(a, v1.x, Xp.x);   Xp.y = fma(a, v1.y, Xp.y);
        Xpp.x = fma(a, v2.x, Xpp.x); Xpp.y = fma(a, v2.y, Xpp.y);
    }
}

__global__ __launch_bounds__(256) void local_coordinates_kernel(CurveArgs c, long long n,
                                                                const double* __restrict__ px,
                                                                const double* __restrict__ py,
                                                                const double* __restrict__ t0,
                                                                double width, double tol, int maxiter,
                                                                double* __restrict__ r_out,
                                                                double* __restrict__ t_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = px[i], y = py[i];
    double t = t0[i];
    double2 X, Xp, Xpp;
    for (int it = 0; it < maxiter; ++it) {
        curve_eval(c, t, X, Xp, Xpp);
        const double dx = x - X.x, dy = y - X.y;
        const double sp2 = Xp.x * Xp.x + Xp.y * Xp.y;
        const double g = dx * Xp.x + dy * Xp.y;
        double gp = -sp2 + dx * Xpp.x + dy * Xpp.y;
        // Newton derivative not negative definite: Gauss-Newton step (always a descent
        // direction for |p - X|^2)
        if (!(gp < -1e-300)) gp = -sp2;
        double dt = -g / gp;
        const double lim = 0.25 * width / sqrt(sp2) + c.hf;
        dt = fmax(fmin(dt, lim), -lim);
        t += dt;
        if (fabs(dt) < tol) break;
    }
    curve_eval(c, t, X, Xp, Xpp);
    const double sp = sqrt(Xp.x * Xp.x + Xp.y * Xp.y);
    const double nx = Xp.y / sp, ny = -Xp.x / sp;
    r_out[i] = (x - X.x) * nx + (y - X.y) * ny;
    const double twopi = 6.283185307179586476925286766559;
    double tm = fmod(t, twopi);
    if (tm < 0.0) tm += twopi;
    t_out[i] = tm;
}

}  // namespace

extern "C" int ipde_curve_local_coordinates(ipde_ctx* ctx, int64_t nf, const double* curve_tab,
                                            const double* bary_w, int64_t npts, const double* px,
                                            const double* py, const double* t0, double width, double tol,
                                            int maxiter, double* r_out, double* t_out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, nf >= NL && nf < (1ll << 30) && curve_tab && bary_w && npts >= 0 && width > 0.0 &&
                            tol > 0.0 && maxiter > 0);
    if (npts == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, px && py && t0 && r_out && t_out);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    CurveArgs c;
    c.tab = (const double2*)curve_tab;
    c.nf = (int)nf;
    c.hf = 6.283185307179586476925286766559 / (double)nf;
    for (int j = 0; j < NL; ++j) c.w[j] = bary_w[j];     // host array of NL weights
    const unsigned blocks = (unsigned)((npts + 255) / 256);
    hipLaunchKernelGGL(local_coordinates_kernel, dim3(blocks), dim3(256), 0, ctx->stream, c, (long long)npts,
                       px, py, t0, width, tol, maxiter, r_out, t_out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// Radial -> grid interpolation, second half (reference
// ipde/embedded_boundary.py:419-443 uses a type-2 NUFFT per Chebyshev mode): the M
// Chebyshev-coefficient rows arrive oversampled along t (cf[m][j] = c_m(2 pi j / nf), nf = 16 N,
// produced by phase-shifted FFTs in ipde_amd/interp.py); a thread evaluates one point
// (xi, t): 16-point barycentric Lagrange along t for every row, Chebyshev recurrence in xi.
// The table (M x nf doubles, 10 MB at M = 20, N = 4096) is L2 resident.
namespace {

constexpr int NLT = 16;   // stencil width along t (interp.py _NL_T)

struct BaryW {
    double w[NLT];
};

__global__ __launch_bounds__(256) void chebfourier_gather_kernel(const double* __restrict__ cf, int M, int nf,
                                                                 BaryW bw, long long n,
                                                                 const double* __restrict__ xi,
                                                                 const double* __restrict__ t,
                                                                 double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double twopi = 6.283185307179586476925286766559;
    const double hf = twopi / (double)nf;
    double tm = fmod(t[i], twopi);
    if (tm < 0.0) tm += twopi;
    const double s = tm / hf;
    const long long i0 = (long long)floor(s) - (NLT / 2 - 1);
    const double u = s - (double)i0;
    double wt[NLT];
    double wsum = 0.0;
    int hit = -1;
#pragma unroll
    for (int j = 0; j < NLT; ++j) {
        const double d = u - (double)j;
        if (d == 0.0) hit = j;
        wt[j] = bw.w[j] / (d == 0.0 ? 1.0 : d);
        wsum += wt[j];
    }
    if (hit >= 0) {
#pragma unroll
        for (int j = 0; j < NLT; ++j) wt[j] = (j == hit) ? 1.0 : 0.0;
        wsum = 1.0;
    }
    const double inv = 1.0 / wsum;
    long long base = i0 % nf;
    if (base < 0) base += nf;
    int idx[NLT];
#pragma unroll
    for (int j = 0; j < NLT; ++j) {
        int k = (int)(base + j);
        idx[j] = k >= nf ? k - nf : k;
        wt[j] *= inv;
    }
    const double x = xi[i];
    double T0 = 1.0, T1 = x, acc = 0.0;
    for (int m = 0; m < M; ++m) {
        const double* row = cf + (size_t)m * nf;
        double b = 0.0;
#pragma unroll
        for (int j = 0; j < NLT; ++j) b = fma(row[idx[j]], wt[j], b);
        double Tm;
        if (m == 0) {
            Tm = 1.0;
        } else if (m == 1) {
            Tm = x;
        } else {
            Tm = 2.0 * x * T1 - T0;
            T0 = T1;
            T1 = Tm;
        }
        acc = fma(b, Tm, acc);
    }
    out[i] = acc;
}

}  // namespace

extern "C" int ipde_chebfourier_gather(ipde_ctx* ctx, int64_t M, int64_t nf, const double* cf,
                                       const double* bary_w, int64_t npts, const double* xi, const double* t,
                                       double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, M > 0 && M < 4096 && nf >= NLT && nf < (1ll << 30) && cf && bary_w && npts >= 0);
    if (npts == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, xi && t && out);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    BaryW bw;
    for (int j = 0; j < NLT; ++j) bw.w[j] = bary_w[j];
    hipLaunchKernelGGL(chebfourier_gather_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, ctx->stream,
                       cf, (int)M, (int)nf, bw, (long long)npts, xi, t, out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}
