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

// ---------------------------------------------------------------------------
// Radial -> grid interpolation, whole (reference ipde/embedded_boundary.py:419-443; the
// solvers call it once per boundary and field at the end of every solve,
// ipde/solvers/multi_boundary/scalar.py:112-116).  One library call, six launches on the
// context's stream, no host round trip:
//   1. Chebyshev analysis along r.  For Chebyshev-Gauss nodes the inverse of the Vandermonde
//      matrix is the discrete cosine sum  c_k = (2 - [k = 0])/M sum_j f_j cos(k theta_j):
//      an M x M table per M, built once in long double and kept in HBM;
//   2. forward FFT of the nf*M coefficient rows along t (rocFFT, batched);
//   3. 16 phase-shifted copies of every spectrum, scaled by 1/N (row (s, r) will hold
//      c_r(t_j + s h/16): transforms of the SAME length N, so no new FFT length);
//   4. inverse FFT of the 16 nf M rows;
//   5. real parts interleaved into the oversampled table cf[r][16 j + s];
//   6. the gather: 16-point barycentric Lagrange along t, Chebyshev recurrence in xi, all
//      fields of a point from one set of weights, written straight to out[f][idx[p]].
namespace {

constexpr int R2G_UP = 16;       // oversampling along t
constexpr int R2G_MAXF = 8;      // fields per call

struct R2gOut {
    double* p[R2G_MAXF];
};

__global__ __launch_bounds__(256) void r2g_analysis_kernel(const double* __restrict__ fr,
                                                           const double* __restrict__ tab, int M, int N,
                                                           double2* __restrict__ c) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const int r = blockIdx.y;          // f*M + k
    const int f = r / M, k = r - f * M;
    const double* col = fr + (size_t)f * M * N + j;
    const double* w = tab + (size_t)k * M;
    double s = 0.0;
    for (int m = 0; m < M; ++m) s = fma(w[m], col[(size_t)m * N], s);
    c[(size_t)r * N + j] = make_double2(s, 0.0);
}

__global__ __launch_bounds__(256) void r2g_phase_kernel(const double2* __restrict__ ch, int R, int N,
                                                        double2* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int r = blockIdx.y;
    const double2 v = ch[(size_t)r * N + k];
    const int ks = (k < (N + 1) / 2) ? k : k - N;      // signed wavenumber (Nyquist negative)
    const double sc = 1.0 / (double)N;
    const double vr = v.x * sc, vi = v.y * sc;
#pragma unroll
    for (int s = 0; s < R2G_UP; ++s) {
        double sn, cs;
        sincospi((double)(s * (long long)ks) / (double)(R2G_UP / 2 * (long long)N), &sn, &cs);
        out[((size_t)s * R + r) * N + k] = make_double2(vr * cs - vi * sn, vr * sn + vi * cs);
    }
}

__global__ __launch_bounds__(256) void r2g_interleave_kernel(const double2* __restrict__ fine, int R, int N,
                                                             double* __restrict__ cf) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const int r = blockIdx.y;
    double v[R2G_UP];
#pragma unroll
    for (int s = 0; s < R2G_UP; ++s) v[s] = fine[((size_t)s * R + r) * N + j].x;
    double4* o = (double4*)(cf + ((size_t)r * N + j) * R2G_UP);
#pragma unroll
    for (int q = 0; q < R2G_UP / 4; ++q) o[q] = make_double4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

__global__ __launch_bounds__(256) void r2g_gather_kernel(const double* __restrict__ cf, int nfld, int M, int nf,
                                                         BaryW bw, long long n, const double* __restrict__ xi,
                                                         const double* __restrict__ t,
                                                         const long long* __restrict__ idx, R2gOut out) {
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
    int ix[NLT];
#pragma unroll
    for (int j = 0; j < NLT; ++j) {
        int k = (int)(base + j);
        ix[j] = k >= nf ? k - nf : k;
        wt[j] *= inv;
    }
    const double x = xi[i];
    const long long o = idx ? idx[i] : i;
    for (int f = 0; f < nfld; ++f) {
        double T0 = 1.0, T1 = x, acc = 0.0;
        for (int m = 0; m < M; ++m) {
            const double* row = cf + ((size_t)f * M + m) * nf;
            double b = 0.0;
#pragma unroll
            for (int j = 0; j < NLT; ++j) b = fma(row[ix[j]], wt[j], b);
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
        out.p[f][o] = acc;
    }
}

int cheb_table(ipde_ctx* ctx, int M, const double** tab) {
    auto it = ctx->cheb_tab.find(M);
    if (it != ctx->cheb_tab.end()) {
        *tab = it->second;
        return IPDE_OK;
    }
    // nodes ascending: x_m = cos(theta_m), theta_m = pi (2 (M-1-m) + 1) / (2 M)
    std::vector<double> h((size_t)M * M);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k = 0; k < M; ++k)
        for (int m = 0; m < M; ++m) {
            const long double th = pi * (long double)(2 * (M - 1 - m) + 1) / (long double)(2 * M);
            h[(size_t)k * M + m] = (double)((k == 0 ? 1.0L : 2.0L) / (long double)M * cosl((long double)k * th));
        }
    double* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(double)) != hipSuccess) {
        IPDE_SET_ERR(ctx, "hipMalloc of the Chebyshev analysis table (M = %d) failed", M);
        return IPDE_ERR_ALLOC;
    }
    // (synchronous copy from the stack-owned vector: once per M and context)
    if (hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(d);
        IPDE_SET_ERR(ctx, "upload of the Chebyshev analysis table (M = %d) failed", M);
        return IPDE_ERR_HIP;
    }
    ctx->cheb_tab[M] = d;
    *tab = d;
    return IPDE_OK;
}

}  // namespace

int ipde_fft1_exec(ipde_ctx* ctx, int64_t batch, int64_t n, int direction, const void* in, void* out);  // spectral.hip

extern "C" int ipde_radial_to_grid(ipde_ctx* ctx, int loc, int64_t nfld, int64_t M, int64_t N, const double* fr,
                                   const double* bary_w, int64_t npts, const double* xi, const double* t,
                                   const int64_t* idx, double* const* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, nfld >= 1 && nfld <= R2G_MAXF && M >= 1 && M <= 512 && N >= NLT && N < (1 << 24));
    IPDE_CHECK_ARG(ctx, nfld * M <= 65535 && fr && bary_w && npts >= 0 && out);
    R2gOut o{};
    for (int f = 0; f < nfld; ++f) {
        IPDE_CHECK_ARG(ctx, out[f] != nullptr);
        o.p[f] = out[f];
    }
    if (npts == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, xi && t);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int R = (int)(nfld * M);
    const size_t rowsz = (size_t)R * N * sizeof(double2);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->r2g[0], rowsz));
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->r2g[1], rowsz));
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->r2g[2], R2G_UP * rowsz));
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->r2g[3], R2G_UP * rowsz));
    const double* tab;
    IPDE_TRY(cheb_table(ctx, (int)M, &tab));
    const double* d_fr;
    IPDE_TRY(ipde_stage_in(ctx, loc, 0, fr, (size_t)R * N, &d_fr));
    double2* c = (double2*)ctx->r2g[0].p;
    double2* ch = (double2*)ctx->r2g[1].p;
    double2* ph = (double2*)ctx->r2g[2].p;
    double2* fine = (double2*)ctx->r2g[3].p;
    double* cf = (double*)ctx->r2g[2].p;      // the phase-shifted spectra are dead after step 4
    const dim3 g((unsigned)((N + 255) / 256), (unsigned)R);
    hipLaunchKernelGGL(r2g_analysis_kernel, g, dim3(256), 0, ctx->stream, d_fr, tab, (int)M, (int)N, c);
    IPDE_TRY(ipde_fft1_exec(ctx, R, N, -1, c, ch));
    hipLaunchKernelGGL(r2g_phase_kernel, g, dim3(256), 0, ctx->stream, ch, R, (int)N, ph);
    IPDE_TRY(ipde_fft1_exec(ctx, (int64_t)R2G_UP * R, N, +1, ph, fine));
    hipLaunchKernelGGL(r2g_interleave_kernel, g, dim3(256), 0, ctx->stream, fine, R, (int)N, cf);
    BaryW bw;
    for (int j = 0; j < NLT; ++j) bw.w[j] = bary_w[j];
    hipLaunchKernelGGL(r2g_gather_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, ctx->stream, cf,
                       (int)nfld, (int)M, (int)(R2G_UP * N), bw, (long long)npts, xi, t, (const long long*)idx, o);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// Grid <-> list moves of the multi-boundary solvers (reference ipde/embedded_function.py:105-113,
// 135-138 and ipde/solvers/multi_boundary/scalar.py:72-117 do them with numpy masks): the forcing's
// physical values onto the zero-filled grid times the grid step function, list values added at
// listed grid points, listed grid points gathered.  One launch each, HBM bound.
namespace {

__global__ __launch_bounds__(256) void grid_scatter_kernel(long long n, const long long* __restrict__ idx,
                                                           const double* __restrict__ src,
                                                           const double* __restrict__ scale,
                                                           double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long k = idx[i];
    out[k] = scale ? src[i] * scale[k] : src[i];
}

__global__ __launch_bounds__(256) void grid_add_at_kernel(long long n, const long long* __restrict__ idx,
                                                          const double* __restrict__ src, double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long k = idx[i];
    out[k] = out[k] + src[i];
}

__global__ __launch_bounds__(256) void grid_gather_kernel(long long n, const long long* __restrict__ idx,
                                                          const double* __restrict__ in, double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = in[idx[i]];
}

}  // namespace

extern "C" int ipde_grid_scatter(ipde_ctx* ctx, int64_t ngrid, int64_t nidx, const int64_t* idx, const double* src,
                                 const double* scale, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ngrid >= 0 && nidx >= 0 && nidx <= ngrid && (ngrid == 0 || out));
    IPDE_CHECK_ARG(ctx, nidx == 0 || (idx && src));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ngrid) IPDE_HIP_CHECK(ctx, hipMemsetAsync(out, 0, (size_t)ngrid * sizeof(double), ctx->stream));
    if (nidx)
        hipLaunchKernelGGL(grid_scatter_kernel, dim3((unsigned)((nidx + 255) / 256)), dim3(256), 0, ctx->stream,
                           (long long)nidx, (const long long*)idx, src, scale, out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_grid_add_at(ipde_ctx* ctx, int64_t nidx, const int64_t* idx, const double* src, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, nidx >= 0);
    if (nidx == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, idx && src && out);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(grid_add_at_kernel, dim3((unsigned)((nidx + 255) / 256)), dim3(256), 0, ctx->stream,
                       (long long)nidx, (const long long*)idx, src, out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_grid_gather(ipde_ctx* ctx, int64_t nidx, const int64_t* idx, const double* in, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, nidx >= 0);
    if (nidx == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, idx && in && out);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(grid_gather_kernel, dim3((unsigned)((nidx + 255) / 256)), dim3(256), 0, ctx->stream,
                       (long long)nidx, (const long long*)idx, in, out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// Inside/outside classification of a whole grid from the near band alone (host form:
// ipde_amd/near.py grid_inside_curve; the reference classifies with the sign of r in the band
// and a polygon test elsewhere, ipde/embedded_boundary.py:185-214 + ebdy_collection.py:330-372).
// Band cells carry the sign of their r; every other cell of a row inherits the state of the
// last band cell before it (rows start outside).  A workgroup per row: each thread owns a
// contiguous chunk, publishes the state its chunk ends in, looks back over the chunks before
// it for its carry-in, then fills.  16.8 MB of bytes at 4096^2 - microseconds; the host scan
// (int64 running maximum + gather) took 0.13 s.
namespace {

constexpr unsigned char CELL_UNKNOWN = 0xFF;

__global__ __launch_bounds__(256) void band_sign_scatter_kernel(long long n, const long long* __restrict__ ix,
                                                                const long long* __restrict__ iy,
                                                                const double* __restrict__ r, long long ny,
                                                                unsigned char* __restrict__ state) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    state[ix[i] * ny + iy[i]] = r[i] < 0.0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void row_state_fill_kernel(long long ny, unsigned char* __restrict__ state) {
    __shared__ unsigned char last[256];
    unsigned char* row = state + (long long)blockIdx.x * ny;
    const long long chunk = (ny + 255) / 256;
    const long long j0 = (long long)threadIdx.x * chunk;
    const long long j1 = j0 + chunk < ny ? j0 + chunk : ny;
    unsigned char s = CELL_UNKNOWN;
    for (long long j = j0; j < j1; ++j) {
        const unsigned char v = row[j];
        if (v != CELL_UNKNOWN) s = v;
    }
    last[threadIdx.x] = s;
    __syncthreads();
    unsigned char carry = 0;                       // rows start outside
    for (int k = (int)threadIdx.x - 1; k >= 0; --k)
        if (last[k] != CELL_UNKNOWN) {
            carry = last[k];
            break;
        }
    for (long long j = j0; j < j1; ++j) {
        const unsigned char v = row[j];
        if (v != CELL_UNKNOWN) carry = v;
        else row[j] = carry;
    }
}

}  // namespace

extern "C" int ipde_grid_inside_scan(ipde_ctx* ctx, int64_t nx, int64_t ny, int64_t nband, const int64_t* ix,
                                     const int64_t* iy, const double* r, uint8_t* inside) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, nx >= 0 && ny >= 0 && nband >= 0 && nx < (1ll << 31) && ny < (1ll << 31));
    if (nx == 0 || ny == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, inside && (nband == 0 || (ix && iy && r)));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    IPDE_HIP_CHECK(ctx, hipMemsetAsync(inside, CELL_UNKNOWN, (size_t)nx * (size_t)ny, ctx->stream));
    if (nband)
        hipLaunchKernelGGL(band_sign_scatter_kernel, dim3((unsigned)((nband + 255) / 256)), dim3(256), 0,
                           ctx->stream, (long long)nband, (const long long*)ix, (const long long*)iy, r,
                           (long long)ny, inside);
    hipLaunchKernelGGL(row_state_fill_kernel, dim3((unsigned)nx), dim3(256), 0, ctx->stream, (long long)ny, inside);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}
