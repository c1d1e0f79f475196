// Modified Helmholtz (k^2 - Lap) single/double layer dense sums.
//
//   SLP: (1/2pi) K0(k r) w_sigma          DLP: (k/2pi) K1(k r) (n.d)/r w_tau
//
// Coordinates are scaled by k on the way in (pack kernel / target load), so that
// y = |t' - s'|^2 = (k r)^2 comes straight out of the distance computation and
//   SLP term = q' K0(sqrt y),                 q' = w_sigma/(2 pi)
//   DLP term = (a'.d') K1(sqrt y)/sqrt y,     a' = n w_tau k/(2 pi)
// Record rows: [0] k x [1] k y [2] q' [3] ax' [4] ay'.
//
// Table kernel (the fast path).  F0(y) = K0(sqrt y) and F1(y) = K1(sqrt y)/sqrt y are
// smooth on log-spaced intervals (the only singularity is y = 0), so each covered binade
// is cut into 2^6 intervals by the top 6 mantissa bits of y; per interval the LDS table
// holds the 6 coefficients of a degree-5 polynomial in z = y - c (c the interval's centre,
// |z| <= 2^-7 c), fitted at Chebyshev nodes from long-double std::cyl_bessel_k at context
// creation.  A table covers 32 binades; EIGHT windows [2^(-21+2w), 2^(11+2w)) are resident
// in HBM and the pack kernel picks, from the bounding box, the lowest one whose top covers
// (k * diameter)^2 — so for any k every pair farther apart than diameter * 2^-16 is a table
// hit (with the single window [2^-21, 2^11), k r > 45 sent whole lanes to the fallback:
// 75 ms instead of 10 ms at k = 100 on the 2048^2 x 4096 case).  Pointwise relative
// accuracy: 1e-15 for k r <= 11, 2e-14 at 20, 9e-14 at 30, 2e-12 at 40, 1e-10 at 45
// (where the value is 1e-20 of a near-field one); y below the window or r = 0 is detected
// per lane and that lane's targets are redone with the series / Chebyshev code below.
//   per pair: 4 (dx,dy,y) + 1 (z) + 5 (Horner) + 1 (accumulate) fp64 ops
//             (+2 for the DLP's a.d), 4 int32 ops, 3 ds_read_b128 (48-byte entry).
// Roofline: fp64 VALU / LDS-read co-bound; algorithmic HBM traffic 24 B per target.
#include "layer_pack.h"
#include "bessel_device.h"
#include <cmath>

namespace {

constexpr int MODE_SLP = 1, MODE_DLP = 2;

template <int MODE, bool SKIP, int R>
__device__ __forceinline__ void modhelm_generic_loop(const double* __restrict__ rec, int j0, int j1,
                                                     const double (&x)[R], const double (&y)[R],
                                                     double (&acc)[R]) {
    for (int j = j0; j < j1; ++j) {
        double sx = rec[ipde_rec_index(j, 0)], sy = rec[ipde_rec_index(j, 1)];
        double q = rec[ipde_rec_index(j, 2)];
        double ax = rec[ipde_rec_index(j, 3)], ay = rec[ipde_rec_index(j, 4)];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double dx = x[r] - sx, dy = y[r] - sy;
            double d2 = fma(dy, dy, dx * dx);
            if (SKIP && d2 == 0.0) continue;
            double k0 = 0.0, k1x = 0.0;
            bessel_k01<MODE>(d2, k0, k1x);
            if (MODE & MODE_SLP) acc[r] = fma(q, k0, acc[r]);
            if (MODE & MODE_DLP) acc[r] = fma(fma(ay, dy, ax * dx), k1x, acc[r]);
        }
    }
}

template <int R, int NT>
__device__ __forceinline__ void modhelm_store(const double (&acc)[R], int64_t base, int64_t nt,
                                              double* __restrict__ out, int accumulate) {
    double* o = out + (size_t)blockIdx.y * nt;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = base + (int64_t)r * NT;
        if (i < nt) o[i] = accumulate ? o[i] + acc[r] : acc[r];
    }
}

template <int MODE, bool SKIP, int R, int NT>
__global__ __launch_bounds__(NT) void modhelm_generic_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ out,
    const ApplyParams* __restrict__ prm, int accumulate) {
    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = prm->scale;
    double x[R], y[R], acc[R];
    int64_t base = (int64_t)blockIdx.x * (R * NT) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + (int64_t)r * NT, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = 0.0;
    }
    modhelm_generic_loop<MODE, SKIP, R>(rec, j0, j1, x, y, acc);
    modhelm_store<R, NT>(acc, base, nt, out, accumulate);
}

// ---------------------------------------------------------------------------
// table kernel
#define KT_B 6
#define KT_SHIFT (20 - KT_B)
#define KT_BINADES 32
#define KT_NKEYS (KT_BINADES << KT_B)    // 2048 entries
#define KT_NC 6                          // coefficients: degree 5 on 64 intervals per binade, in z = y - c
#define KT_ENTRY 6                       // doubles per entry: b0..b5 (48 B: THREE ds_read_b128)
#define KT_READS 3
// Round 1 had degree 7 on 32 intervals in 80-byte entries {R, a0..a7} (five reads per pair).  The
// kernel is bound by LDS reads — 4.7-5.3 LDS cycles per ds_read_b128 of random entries, against 15-16
// VALU issue cycles per pair and CU — so the entry is what to shrink:
//   * the interval's centre c comes out of the bits of y itself (top 6 mantissa bits kept, the next
//     one set: one v_and_or_b32), so no reciprocal is stored: z = y - c exactly, coefficients
//     b_j = a_j / c^j;
//   * twice the intervals pay for two dropped degrees: pointwise RELATIVE error 5e-15 for k r <= 5,
//     6e-14 at 10, 2e-12 at 20, 7e-12 at 30 (degree 7 / 32: 1e-15, 1e-14, 6e-14) — ABSOLUTE error below
//     2e-15 of the near-field values everywhere (the functions decay like exp(-k r)), which is what the
//     1e-12-of-max|u| parity bar and the solvers' tolerances see.
// Tried on the way (2048^2 x 4096, k = 10): degree 6 with R in packed 64-byte slots — chunk c of every
// entry in the same four bank groups, 12.8 ms; the same in 80-byte slots (160 KiB, all of a CU's LDS):
// 9.6 ms against 9.9 for round 1's table; 2, 3 or 4 targets per lane and two sources in flight: equal.
#define KT_EXP_LO (-21)                  // window 0 covers y in [2^-21, 2^11): k r in [7e-4, 45]
#define KT_NWIN 8                        // window w is shifted up by 2w binades (k r up to 5800)

template <int MODE, int R, int NT, int U>
__global__ __launch_bounds__(NT) void modhelm_table_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ out,
    const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab, int accumulate) {
    extern __shared__ double2 ltab[];  // KT_NKEYS * KT_ENTRY/2 double2
    // the pack kernel picked the window from the bounding box: (k * diameter)^2 < 2^(11 + 2w),
    // so every pair farther apart than diameter * 2^-16 is inside the table whatever k is
    // (window KT_NWIN: the pairs reach beyond the last one, or every pair is at k r >= 18 where the
    // table's relative error would show (layer_pack.h) — every lane takes the generic body)
    const bool nowin = prm->pad >= KT_NWIN;
    const int win = nowin ? KT_NWIN - 1 : prm->pad;
    gtab += (size_t)win * KT_NKEYS * (KT_ENTRY / 2);
    for (unsigned i = threadIdx.x; i < KT_NKEYS * (KT_ENTRY / 2); i += NT) ltab[i] = gtab[i];
    __syncthreads();
    const unsigned key_lo = (unsigned)((1023 + KT_EXP_LO + 2 * win) << KT_B);
    // only the LOWER end needs watching: the window's top covers the bounding box (pack kernel)
    unsigned hmin = 0xFFFFFFFFu;

    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = prm->scale;
    double x[R], y[R], acc[R];
    int64_t base = (int64_t)blockIdx.x * (R * NT) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + (int64_t)r * NT, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = 0.0;
    }
    // (no window: the table pass is skipped, every lane goes to the generic body below)
    const int b_end = nowin ? j0 / IPDE_SRC_PAD : j1 / IPDE_SRC_PAD;
    for (int b = j0 / IPDE_SRC_PAD; b < b_end; ++b) {
        SrcRow sx, sy, sq, sax, say;
        sx.load(rec, b, 0);
        sy.load(rec, b, 1);
        if (MODE & MODE_SLP) sq.load(rec, b, 2);
        if (MODE & MODE_DLP) {
            sax.load(rec, b, 3);
            say.load(rec, b, 4);
        }
#pragma unroll
        for (int u0 = 0; u0 < IPDE_SRC_PAD; u0 += U) {
            double dx[U][R], dy[U][R], d2[U][R];
            double2 e[U][R][KT_READS];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    dx[u][r] = x[r] - sx.v[u0 + u];
                    dy[u][r] = y[r] - sy.v[u0 + u];
                    d2[u][r] = fma(dy[u][r], dy[u][r], dx[u][r] * dx[u][r]);
                    unsigned hi = (unsigned)__double2hiint(d2[u][r]);
                    hmin = min(hmin, hi);
                    unsigned idx;
                    static_assert(KT_SHIFT == 14 && KT_B + 5 == 11, "literal operands below");
                    asm("v_bfe_u32 %0, %1, 14, 11" : "=v"(idx) : "v"(hi));
                    // 24-bit multiply-add is full rate (v_mul_lo_u32 is quarter rate)
                    const double2* ep = (const double2*)((const char*)ltab +
                                                         __umul24(idx, KT_ENTRY * 8u));
#pragma unroll
                    for (int c = 0; c < KT_READS; ++c) e[u][r][c] = ep[c];
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double2* c = e[u][r];
                    // the centre of y's interval: sign, exponent and 6 mantissa bits of y, then a one
                    static_assert(KT_B == 6 && KT_NC == 6 && KT_READS == 3, "entry = {b0, b1}, {b2, b3}, {b4, b5}");
                    const unsigned hi = (unsigned)__double2hiint(d2[u][r]);
                    const double ctr = __hiloint2double((int)((hi & 0xFFFFC000u) | 0x2000u), 0);
                    const double z = d2[u][r] - ctr;
                    double p = fma(c[2].y, z, c[2].x);   // b5 z + b4
                    p = fma(p, z, c[1].y);               // b3
                    p = fma(p, z, c[1].x);               // b2
                    p = fma(p, z, c[0].y);               // b1
                    p = fma(p, z, c[0].x);               // b0
                    if (MODE == MODE_SLP) {
                        acc[r] = fma(sq.v[u0 + u], p, acc[r]);
                    } else {
                        double ad = fma(say.v[u0 + u], dy[u][r], sax.v[u0 + u] * dx[u][r]);
                        acc[r] = fma(ad, p, acc[r]);
                    }
                }
        }
    }
    const bool inside = (hmin >> KT_SHIFT) >= key_lo && !nowin;
    if (!inside) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        modhelm_generic_loop<MODE, false, R>(rec, j0, j1, x, y, acc);
    }
    modhelm_store<R, NT>(acc, base, nt, out, accumulate);
}

__global__ __launch_bounds__(256) void reduce_partials_acc(const double* __restrict__ part,
                                                           int nchunk, int64_t nt,
                                                           double* __restrict__ out,
                                                           int accumulate) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    double s = accumulate ? out[i] : 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[(size_t)c * nt + i];
    out[i] = s;
}

// one pass (MODE_SLP or MODE_DLP); accumulate != 0 adds into `out`
template <int MODE>
int launch_modhelm(ipde_ctx* ctx, const double* rec, int64_t ns, const double* tx, const double* ty,
                   int64_t nt, double* out, const ApplyParams* prm, int flags, int accumulate) {
    const bool skip = (flags & IPDE_FLAG_SKIP_COINCIDENT) != 0;
    const bool generic = skip || (flags & IPDE_FLAG_GENERIC_MATH) != 0;
    constexpr int NT_T = 1024;
    constexpr int NT_G = 256, R_G = 2;
    // targets per lane x sources in flight of the table kernel (option "modhelm_variant")
    const int variant = ctx->opt_modhelm_variant;
    const int R_T = variant == 1 ? 4 : variant == 3 ? 3 : 2;
    const LayerGeom g = generic ? ipde_layer_geom(ns, nt, NT_G * R_G, 2 * ctx->num_cu)
                                : ipde_layer_geom(ns, nt, NT_T * R_T, ctx->num_cu);
    double* dst = out;
    int acc_main = accumulate;
    if (g.nchunk > 1) {
        IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, (size_t)g.nchunk * nt * sizeof(double)));
        dst = (double*)ctx->partial.p;
        acc_main = 0;
    }
    dim3 grid((unsigned)g.gx, (unsigned)g.nchunk);
    if (accumulate) ipde_time_continue(ctx);
    else ipde_time_begin(ctx);
    if (generic) {
        if (skip)
            hipLaunchKernelGGL((modhelm_generic_kernel<MODE, true, R_G, NT_G>), grid, dim3(NT_G), 0,
                               ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, dst, prm, acc_main);
        else
            hipLaunchKernelGGL((modhelm_generic_kernel<MODE, false, R_G, NT_G>), grid, dim3(NT_G), 0,
                               ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, dst, prm, acc_main);
    } else {
        const double* tab = ctx->d_ktab + (MODE == MODE_SLP ? 0 : (size_t)KT_NWIN * KT_NKEYS * KT_ENTRY);
        size_t lds = (size_t)KT_NKEYS * KT_ENTRY * sizeof(double);
        auto go = [&](auto kern) -> int {
            IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)lds));
            hipLaunchKernelGGL(kern, grid, dim3(NT_T), lds, ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, dst,
                               prm, (const double2*)tab, acc_main);
            return IPDE_OK;
        };
        int st;
        switch (variant) {
            case 1: st = go(modhelm_table_kernel<MODE, 4, NT_T, 1>); break;
            case 2: st = go(modhelm_table_kernel<MODE, 2, NT_T, 2>); break;
            case 3: st = go(modhelm_table_kernel<MODE, 3, NT_T, 1>); break;
            default: st = go(modhelm_table_kernel<MODE, 2, NT_T, 1>); break;
        }
        IPDE_TRY(st);
    }
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    if (g.nchunk > 1) {
        hipLaunchKernelGGL(reduce_partials_acc, dim3((unsigned)ceil_div64(nt, 256)), dim3(256), 0,
                           ctx->stream, (const double*)dst, g.nchunk, nt, out, accumulate);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    return IPDE_OK;
}

// ---- host: table construction ---------------------------------------------
// degree-(KT_NC-1) fit of F on x = c (1 + z), |z| <= a, at KT_NC Chebyshev nodes
template <class F>
void fit_interval(long double c, long double a, F f, double* coef /*KT_NC*/) {
    const int n = KT_NC;
    long double v[n], b[n];
    for (int i = 0; i < n; ++i) {
        long double u = cosl(M_PIl * (i + 0.5L) / n);
        v[i] = f(c * (1.0L + a * u));
    }
    for (int k = 0; k < n; ++k) {
        long double s = 0.0L;
        for (int i = 0; i < n; ++i) s += v[i] * cosl(M_PIl * k * (i + 0.5L) / n);
        b[k] = s * 2.0L / n;
    }
    b[0] *= 0.5L;
    // Chebyshev -> monomial in u via T_{k+1} = 2 u T_k - T_{k-1}
    long double mono[n] = {0}, Tkm1[n] = {0}, Tk[n] = {0};
    Tkm1[0] = 1.0L;  // T0
    Tk[1] = 1.0L;    // T1
    for (int j = 0; j < n; ++j) mono[j] += b[0] * Tkm1[j] + b[1] * Tk[j];
    for (int k = 2; k < n; ++k) {
        long double Tn[n] = {0};
        for (int j = 0; j + 1 < n; ++j) Tn[j + 1] += 2.0L * Tk[j];
        for (int j = 0; j < n; ++j) Tn[j] -= Tkm1[j];
        for (int j = 0; j < n; ++j) {
            mono[j] += b[k] * Tn[j];
            Tkm1[j] = Tk[j];
            Tk[j] = Tn[j];
        }
    }
    long double s = 1.0L;
    for (int j = 0; j < n; ++j) {
        coef[j] = (double)(mono[j] / s);  // u = z / a
        s *= a;
    }
}

}  // namespace

// One table set per DEVICE (3.4 MB; ~0.14 s of long-double Bessel evaluations): the solvers keep a
// context per boundary thread, and each used to build its own.
static std::mutex g_ktab_mutex;
static std::map<int, double*> g_ktab;

int ipde_build_k_table(ipde_ctx* ctx) {
    std::lock_guard<std::mutex> guard(g_ktab_mutex);
    auto it = g_ktab.find(ctx->device);
    if (it != g_ktab.end()) {
        ctx->d_ktab = it->second;
        return IPDE_OK;
    }
    IPDE_HIP_CHECK(ctx, ipde_bessel_upload());
    // piecewise tables of F0(y) = K0(sqrt y), F1(y) = K1(sqrt y)/sqrt y
    // layout: [kind (K0 | K1/x)][window][key][entry]
    std::vector<double> h((size_t)2 * KT_NWIN * KT_NKEYS * KT_ENTRY, 0.0);
    auto F0 = [](long double y) { return std::cyl_bessel_kl(0.0L, sqrtl(y)); };
    auto F1 = [](long double y) {
        long double x = sqrtl(y);
        return std::cyl_bessel_kl(1.0L, x) / x;
    };
    for (int w = 0; w < KT_NWIN; ++w) {
        const uint64_t key_lo = (uint64_t)((1023 + KT_EXP_LO + 2 * w) << KT_B);
        for (int n = 0; n < KT_NKEYS; ++n) {
            uint64_t key = key_lo + (uint64_t)n;
            size_t pos = (size_t)(key & (uint64_t)(KT_NKEYS - 1));
            uint64_t lo_bits = key << (52 - KT_B), hi_bits = (key + 1) << (52 - KT_B);
            double xlo, xhi;
            memcpy(&xlo, &lo_bits, 8);
            memcpy(&xhi, &hi_bits, 8);
            // the centre the kernel forms from the bits of y: exactly the midpoint
            const long double c = 0.5L * ((long double)xlo + (long double)xhi);
            const long double a = ((long double)xhi - (long double)xlo) / (2.0L * c);
            for (int t = 0; t < 2; ++t) {
                double* e = &h[(((size_t)t * KT_NWIN + w) * KT_NKEYS + pos) * KT_ENTRY];
                // beyond k r ~ 700 the functions underflow even in long double: zero entries
                if (xlo > 4.9e5) continue;
                if (t == 0)
                    fit_interval(c, a, F0, e);
                else
                    fit_interval(c, a, F1, e);
                // from the relative variable y / c - 1 to z = y - c
                long double s = 1.0L;
                for (int j = 0; j < KT_NC; ++j) {
                    e[j] = (double)((long double)e[j] / s);
                    s *= c;
                }
            }
        }
    }
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_ktab, h.size() * sizeof(double)));
    IPDE_HIP_CHECK(ctx, hipMemcpy(ctx->d_ktab, h.data(), h.size() * sizeof(double),
                                  hipMemcpyHostToDevice));
    g_ktab[ctx->device] = ctx->d_ktab;
    return IPDE_OK;
}

extern "C" int ipde_modhelm_apply(ipde_ctx* ctx, int loc, double k, int64_t ns, const double* sx,
                                  const double* sy, const double* w_sigma, const double* nx,
                                  const double* ny, const double* w_tau, int64_t nt,
                                  const double* tx, const double* ty, double* out, int flags) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, ns >= 0 && nt >= 0 && ns < (1LL << 30));
    IPDE_CHECK_ARG(ctx, k > 0.0);
    IPDE_CHECK_ARG(ctx, w_sigma != nullptr || w_tau != nullptr);
    IPDE_CHECK_ARG(ctx, w_tau == nullptr || (nx != nullptr && ny != nullptr));
    if (nt == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, tx && ty && out);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_ktab) IPDE_TRY(ipde_build_k_table(ctx));   // first use on this context
    double* d_out;
    IPDE_TRY(ipde_stage_out(ctx, loc, 7, out, nt, &d_out));
    if (ns == 0) {
        IPDE_HIP_CHECK(ctx, hipMemsetAsync(d_out, 0, nt * sizeof(double), ctx->stream));
        return ipde_stage_finish(ctx, loc, 7, out, nt);
    }
    IPDE_CHECK_ARG(ctx, sx && sy);
    const double *d_sx, *d_sy, *d_q, *d_nx, *d_ny, *d_tau, *d_tx, *d_ty;
    IPDE_TRY(ipde_stage_in(ctx, loc, 0, sx, ns, &d_sx));
    IPDE_TRY(ipde_stage_in(ctx, loc, 1, sy, ns, &d_sy));
    IPDE_TRY(ipde_stage_in(ctx, loc, 2, w_sigma, ns, &d_q));
    IPDE_TRY(ipde_stage_in(ctx, loc, 3, w_tau ? nx : nullptr, ns, &d_nx));
    IPDE_TRY(ipde_stage_in(ctx, loc, 4, w_tau ? ny : nullptr, ns, &d_ny));
    IPDE_TRY(ipde_stage_in(ctx, loc, 5, w_tau, ns, &d_tau));
    IPDE_TRY(ipde_stage_in(ctx, loc, 8, tx, nt, &d_tx));
    IPDE_TRY(ipde_stage_in(ctx, loc, 9, ty, nt, &d_ty));
    PackArgs pa{};
    pa.sx = d_sx;
    pa.sy = d_sy;
    pa.ch[0] = d_q;
    pa.mul[0] = 0.5 / M_PI;
    pa.ch[1] = d_nx;
    pa.mulby[1] = d_tau;
    pa.mul[1] = 0.5 * k / M_PI;
    pa.ch[2] = d_ny;
    pa.mulby[2] = d_tau;
    pa.mul[2] = 0.5 * k / M_PI;
    pa.corr_ch = -1;
    pa.corr2_ch = -1;
    pa.use_scale = 1;   // bounding box on the device: selects the table window
    pa.fixed_scale = k;
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, d_tx, d_ty, nt, &rec, &prm));
    int accumulate = 0;
    if (w_sigma) {
        IPDE_TRY(launch_modhelm<MODE_SLP>(ctx, rec, ns, d_tx, d_ty, nt, d_out, prm, flags, 0));
        accumulate = 1;
    }
    if (w_tau)
        IPDE_TRY(launch_modhelm<MODE_DLP>(ctx, rec, ns, d_tx, d_ty, nt, d_out, prm, flags, accumulate));
    return ipde_stage_finish(ctx, loc, 7, out, nt);
}
