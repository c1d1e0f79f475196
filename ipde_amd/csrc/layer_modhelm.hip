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

// ---------------------------------------------------------------------------
// Single-layer sums onto patch lists with the far sources of every 8 x 8 block of tiles in a local
// expansion (ipde_modhelm_apply_patches_far; the Laplace form, with the scheme, is in
// layer_laplace.hip).  In the scaled coordinates (k absorbed) Graf's addition theorem gives, for a
// source z_j = c + rho_j e^{i phi_j} and a target z = c + rho e^{i phi}, rho < rho_j,
//     K0(|z - z_j|) = sum_m eps_m K_m(rho_j) I_m(rho) cos(m (phi - phi_j)),   eps_0 = 1, eps_m = 2,
// so a block keeps C_m = eps_m sum_j q_j Kh_m(rho_j) e^{-i m phi_j}, m <= 26, with
// Kh_m = K_m s_m, s_m = (r/2)^m / m! (r the block's half-diagonal): bounded by (r/rho_j)^m / 2m, by the
// upward recurrence Kh_{m+1} = Kh_{m-1} (r^2/4)/(m (m+1)) + Kh_m (m/(m+1)) (r/rho_j) from K0 and K1.
// A target evaluates Re sum_m C_m zeta^m T_m(w), zeta = (z - c)/r, w = (rho/2)^2,
// T_m = I_m(rho) m! (2/rho)^m = sum_n w^n m!/(n! (m+n)!): T_27 and T_26 by their series (12 terms:
// 2e-20 at w = 4), the rest by T_{m-1} = T_m + w T_{m+1} / (m (m+1)) inside the Horner loop.  Blocks with
// r > 4 (k times the half-diagonal) use no expansion.  Sources beyond 4 r: ratio <= 1/4, truncation as
// for the Laplace form.  And whatever the block: a batch of sources all farther than r + 45 from the
// centre is DROPPED — K0(45) = 5e-21 against near-field values of O(1) (at k = 100 on a 3-unit domain
// that is most of the boundary for most blocks).
constexpr int MFAR_P = 26;
constexpr double MFAR_RHO = 0.25;
constexpr double MFAR_RMAX = 4.0;
constexpr double MFAR_DROP = 45.0;      // K0(45) = 5e-21
constexpr int MFAR_TN = 12;
constexpr int MFAR_NCOEF = 2 * (MFAR_P + 2);
constexpr int MFAR_HDR = 4;      // cx, cy, 1/r, r^2/4

// Two levels as in layer_laplace.hip: PPL = 16 first, `nslice` waves (slices of the sources) per parent
// block of sixteen consecutive blocks, `bits` = the batches the parent takes or drops; PPL = 1 then per
// block with `skip` = those bits, `bits` = its near batches.
// WHICH: MODE_SLP, MODE_DLP or both — ONE family of coefficients either way.  The double layer is the source-side
// directional derivative of the single layer, (a . grad_{z_j}) K0(|z - z_j|) = (a . d) K1(|d|) / |d|, and with
// (d_x +- i d_y)[K_m(rho) e^{i m phi}] = -K_{m+-1}(rho) e^{i (m+-1) phi} the derivative of Graf's coefficient
// function is a ladder step:  (a . grad)[K_m e^{-i m phi_j}] = -(a K_{m+1} e^{-i(m+1) phi_j} + conj(a) K_{m-1}
// e^{-i(m-1) phi_j}) / 2, a = a_x + i a_y.  In the scaled quantities Kh_m = K_m (r/2)^m / m! a source adds
//     C_{m-1} += -(2 m / r) a Kh_m e^{-i m phi_j}   (m = 1 .. P + 1; C_0: twice the real part, its own mirror term),
//     C_{m+1} += -(r / (2 (m + 1))) conj(a) Kh_m e^{-i m phi_j}   (m = 0 .. P - 1),
// one more step of the K recurrence than the single layer needs; the targets' side is unchanged.
template <int PPL, int WHICH>
__global__ __launch_bounds__(256) void modhelm_far_coeff_kernel(const double* __restrict__ rec, int ns_pad,
                                                               const double* __restrict__ pxy, int64_t np,
                                                               const ApplyParams* __restrict__ prm,
                                                               double* __restrict__ head, double* __restrict__ coef,
                                                               unsigned* __restrict__ near, int nch,
                                                               const unsigned* __restrict__ skip, int nslice) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t g = gw / nslice;
    const int slice = (int)(gw - g * nslice);
    if (g * 64 * PPL >= np) return;                    // (whole waves)
    const int trips = (ns_pad + 63) / 64, tps = (trips + nslice - 1) / nslice;
    const int jlo = slice * tps * 64, jhi = min(ns_pad, (slice + 1) * tps * 64);
    const double s1 = prm->scale;
    FarBlock blk;
    blk.init<PPL>(pxy, np, g, lane, s1);
    const double cx = blk.cx, cy = blk.cy, r = blk.r, r2 = blk.r2;
    const double r2q = 0.25 * r2;
    const bool block_ok = r <= MFAR_RMAX;
    const double thr = r2 * (1.0 / (MFAR_RHO * MFAR_RHO)) * (1.0 + 0x1p-40);
    const double drop2 = (r + MFAR_DROP) * (r + MFAR_DROP);
    double sre[MFAR_P + 1], sim[MFAR_P + 1];
#pragma unroll
    for (int k = 0; k <= MFAR_P; ++k) sre[k] = sim[k] = 0.0;
    for (int j0 = jlo; j0 < jhi; j0 += 64) {
        if (PPL == 1 && skip) {
            // (a trip whose eight batches the parent block took or dropped whole is nobody's here: see layer_laplace.hip)
            const unsigned t0 = skip[(g >> 4) * nch + (j0 >> 6)];
            if ((t0 & 0xFFu) == 0xFFu) {
                if (lane == 0) near[g * nch + (j0 >> 6)] = 0u;
                continue;
            }
        }
        const int j = j0 + lane;
        const bool valid = j < ns_pad;
        const int jj = valid ? j : ns_pad - 1;
        const double dx = rec[ipde_rec_index(jj, 0)] - cx, dy = rec[ipde_rec_index(jj, 1)] - cy;
        const double d2 = fma(dy, dy, dx * dx);
        // a batch of eight sources goes one way as a whole: dropped (all eight beyond r + 45), else
        // into the expansion (all eight beyond 4 r, block narrow enough), else pair by pair; batches the
        // parent block took or dropped are nobody's here
        const unsigned taken = (PPL == 1 && skip) ? skip[(g >> 4) * nch + (j0 >> 6)] : 0u;
        const bool mine = valid && !((taken >> (lane >> 3)) & 1u);
        const bool negligible = d2 >= drop2;
        const unsigned long long mk = __ballot(mine && !negligible);       // lanes that matter at all
        const unsigned long long m = __ballot(mine && !(block_ok && d2 >= thr));
        const unsigned long long mv = __ballot(valid);
        const int sh = lane & ~7;
        const bool kept = ((mk >> sh) & 0xFFull) != 0;
        const bool far = mine && kept && ((m >> sh) & 0xFFull) == 0;
        if (lane == 0) {
            const unsigned nearbits = far_batch_bits(m) & far_batch_bits(mk);
            // parent level: everything it does not leave to its blocks (taken into its expansion, or dropped)
            near[g * nch + (j0 >> 6)] = PPL == 1 ? nearbits : (far_batch_bits(mv) & ~nearbits);
        }
        if (__ballot(far && !negligible) == 0) continue;              // (wave-uniform)
        const double y = far ? d2 : 1.0;
        const bool live = far && !negligible;
        const double q = ((WHICH & MODE_SLP) && live) ? rec[ipde_rec_index(jj, 2)] : 0.0;
        const double ax = ((WHICH & MODE_DLP) && live) ? rec[ipde_rec_index(jj, 3)] : 0.0;
        const double ay = ((WHICH & MODE_DLP) && live) ? rec[ipde_rec_index(jj, 4)] : 0.0;
        double k0 = 0.0, k1x = 0.0;
        bessel_k01<3>(y, k0, k1x);
        const double rho = sqrt(y), irho = 1.0 / rho;
        const double ere = dx * irho, eim = -dy * irho;        // e^{-i phi_j} (near lanes: weight 0)
        const double rr = r * irho;
        double km1 = k0, km = k1x * y * (0.5 * r) * irho;      // Kh_0, Kh_1 = K1(rho) r / 2
        if (WHICH & MODE_SLP) sre[0] = fma(q, km1, sre[0]);
        // -(2 / r) a and -(r / 2) conj(a)
        const double apx = -2.0 / r * ax, apy = -2.0 / r * ay, aqx = -0.5 * r * ax, aqy = 0.5 * r * ay;
        if (WHICH & MODE_DLP) {                                 // m = 0: Kh_0, phase 1, into C_1
            sre[1] = fma(aqx, km1, sre[1]);
            sim[1] = fma(aqy, km1, sim[1]);
        }
        double pre = ere, pim = eim;                            // e^{-i m phi_j}
        const double q2 = 2.0 * q;
        constexpr int MTOP = (WHICH & MODE_DLP) ? MFAR_P + 1 : MFAR_P;
#pragma unroll
        for (int mm = 1; mm <= MTOP; ++mm) {
            if ((WHICH & MODE_SLP) && mm <= MFAR_P) {
                const double wq = q2 * km;
                sre[mm] = fma(wq, pre, sre[mm]);
                sim[mm] = fma(wq, pim, sim[mm]);
            }
            if (WHICH & MODE_DLP) {
                const double vr = km * pre, vi = km * pim;      // Kh_m e^{-i m phi_j}
                const double tr = apx * vr - apy * vi, ti = fma(apx, vi, apy * vr);
                sre[mm - 1] = fma((double)mm, tr, sre[mm - 1]);
                sim[mm - 1] = fma((double)mm, ti, sim[mm - 1]);
                if (mm + 1 <= MFAR_P) {
                    const double ur = aqx * vr - aqy * vi, ui = fma(aqx, vi, aqy * vr);
                    sre[mm + 1] = fma(1.0 / (double)(mm + 1), ur, sre[mm + 1]);
                    sim[mm + 1] = fma(1.0 / (double)(mm + 1), ui, sim[mm + 1]);
                }
            }
            if (mm < MTOP) {
                const double kn = fma(km1, r2q * (1.0 / ((double)mm * (mm + 1))), km * (((double)mm / (mm + 1)) * rr));
                km1 = km;
                km = kn;
                const double nre = pre * ere - pim * eim;
                pim = fma(pre, eim, pim * ere);
                pre = nre;
            }
        }
    }
#pragma unroll
    for (int k = 0; k <= MFAR_P; ++k) {
        sre[k] = wave_sum(sre[k]);
        sim[k] = wave_sum(sim[k]);
    }
    if (lane == 0) {
        if (slice == 0) {
            head[g * MFAR_HDR + 0] = cx;
            head[g * MFAR_HDR + 1] = cy;
            head[g * MFAR_HDR + 2] = 1.0 / r;
            head[g * MFAR_HDR + 3] = r2q;
        }
        double* c = coef + gw * MFAR_NCOEF;
#pragma unroll
        for (int k = 0; k <= MFAR_P; ++k) {
            c[2 * k] = sre[k];
            c[2 * k + 1] = sim[k];
        }
    }
}

// MODE: the near pairs' kernel (single or double layer; the table of that kind in LDS); FAR: add the block's
// expansion (whose coefficients hold whatever layers the coefficient pass was given); ACC: add to `out`.  Both
// layers in one apply: <SLP, true, false> then <DLP, false, true>.
template <int NT, int MODE, bool FAR, bool ACC>
__global__ __launch_bounds__(NT) void modhelm_patch_far_kernel(
    const double* __restrict__ rec, int ns_pad, const double* __restrict__ pxy, int64_t np,
    const int* __restrict__ pout, double* __restrict__ out, const ApplyParams* __restrict__ prm,
    const double2* __restrict__ gtab, const double* __restrict__ head, const double* __restrict__ coef,
    const unsigned* __restrict__ near, int nch) {
    extern __shared__ double2 ltab[];
    const bool nowin = prm->pad >= KT_NWIN;
    const int win = nowin ? KT_NWIN - 1 : prm->pad;
    gtab += (size_t)win * KT_NKEYS * (KT_ENTRY / 2);
    for (unsigned i = threadIdx.x; i < KT_NKEYS * (KT_ENTRY / 2); i += NT) ltab[i] = gtab[i];
    __syncthreads();
    const unsigned key_lo = (unsigned)((1023 + KT_EXP_LO + 2 * win) << KT_B);
    unsigned hmin = 0xFFFFFFFFu;
    const double s1 = prm->scale;
    // wave w of workgroup b takes block w * gridDim.x + b (see laplace_patch_far_kernel)
    const int64_t g = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x));
    if (g * 64 >= np) return;                          // (whole waves, after the only barrier)
    const int64_t lane = g * 64 + (threadIdx.x & 63);
    const int64_t t = min(lane, np - 1);
    double xs[4], ys[4], acc[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        xs[a] = pxy[(int64_t)a * np + t] * s1;
        ys[a] = pxy[(int64_t)(4 + a) * np + t] * s1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0;
    const unsigned* nm = near + g * nch;
    if (!nowin) {
        for (int c = 0; c < nch; ++c) {
            unsigned m = nm[c];
            while (m) {
                const int bt = __builtin_ctz(m);
                m &= m - 1;
                SrcRow sx, sy, sq, sa;
                sx.load(rec, 8 * c + bt, 0);
                sy.load(rec, 8 * c + bt, 1);
                sq.load(rec, 8 * c + bt, MODE == MODE_SLP ? 2 : 3);      // q, or a_x
                if (MODE == MODE_DLP) sa.load(rec, 8 * c + bt, 4);       // a_y
#pragma unroll
                for (int u = 0; u < IPDE_SRC_PAD; ++u) {
                    double dx2[4], dy2[4], adx[4], ady[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const double dx = xs[a] - sx.v[u], dy = ys[a] - sy.v[u];
                        dx2[a] = dx * dx;
                        dy2[a] = dy * dy;
                        if (MODE == MODE_DLP) {
                            adx[a] = sq.v[u] * dx;
                            ady[a] = sa.v[u] * dy;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        double d2[4];
                        double2 e[4][KT_READS];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            d2[b] = dx2[a] + dy2[b];
                            const unsigned hi = (unsigned)__double2hiint(d2[b]);
                            hmin = min(hmin, hi);
                            unsigned idx;
                            asm("v_bfe_u32 %0, %1, 14, 11" : "=v"(idx) : "v"(hi));
                            const double2* ep = (const double2*)((const char*)ltab + __umul24(idx, KT_ENTRY * 8u));
#pragma unroll
                            for (int cc = 0; cc < KT_READS; ++cc) e[b][cc] = ep[cc];
                        }
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const unsigned hi = (unsigned)__double2hiint(d2[b]);
                            const double ctr = __hiloint2double((int)((hi & 0xFFFFC000u) | 0x2000u), 0);
                            const double z = d2[b] - ctr;
                            double p = fma(e[b][2].y, z, e[b][2].x);
                            p = fma(p, z, e[b][1].y);
                            p = fma(p, z, e[b][1].x);
                            p = fma(p, z, e[b][0].y);
                            p = fma(p, z, e[b][0].x);
                            acc[4 * a + b] = fma(MODE == MODE_SLP ? sq.v[u] : adx[a] + ady[b], p, acc[4 * a + b]);
                        }
                    }
                }
            }
        }
    }
    if (nowin || (hmin >> KT_SHIFT) < key_lo) {
        // no table window for this launch, or a near pair below the window: the near batches of this
        // patch again with the series / Chebyshev code
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double xa[4] = {xs[a], xs[a], xs[a], xs[a]};
            double gs[4] = {0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < nch; ++c) {
                unsigned m = nm[c];
                while (m) {
                    const int bt = __builtin_ctz(m);
                    m &= m - 1;
                    modhelm_generic_loop<MODE, false, 4>(rec, 8 * (8 * c + bt), 8 * (8 * c + bt) + 8, xa, ys, gs);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[4 * a + b] = gs[b];
        }
    }
    // far sources: Re sum_m C_m zeta^m T_m(w), Horner downwards with T's recurrence alongside
    if (FAR) {
        const double* h = head + g * MFAR_HDR;
        const double cx = h[0], cy = h[1], rinv = h[2], r2q = h[3];
        const double* C = coef + g * MFAR_NCOEF;
        double zy[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) zy[b] = (ys[b] - cy) * rinv;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double zx = (xs[a] - cx) * rinv;
            double w[4], tm[4], tp[4], vr[4], vi[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                w[b] = r2q * (zx * zx + zy[b] * zy[b]);        // (rho / 2)^2
                // T_m = sum_n w^n m! / (n! (m + n)!), MFAR_TN terms, m = P + 1 and P
                double s1p = 1.0, s0p = 1.0;
#pragma unroll
                for (int n = MFAR_TN; n >= 1; --n) {
                    s1p = fma(s1p, w[b] * (1.0 / ((double)n * (MFAR_P + 1 + n))), 1.0);
                    s0p = fma(s0p, w[b] * (1.0 / ((double)n * (MFAR_P + n))), 1.0);
                }
                tp[b] = s1p;
                tm[b] = s0p;
                vr[b] = vi[b] = 0.0;
            }
            for (int m = MFAR_P; m >= 0; --m) {
                const double cr = C[2 * m], ci = C[2 * m + 1];
                const double f = m >= 1 ? 1.0 / ((double)m * (m + 1)) : 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    // v = v zeta + C_m T_m
                    const double nr = vr[b] * zx - vi[b] * zy[b];
                    vi[b] = fma(vr[b], zy[b], vi[b] * zx) + ci * tm[b];
                    vr[b] = nr + cr * tm[b];
                    // T_{m-1} = T_m + w T_{m+1} / (m (m + 1))
                    const double tn = fma(w[b] * f, tp[b], tm[b]);
                    tp[b] = tm[b];
                    tm[b] = tn;
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[4 * a + b] += vr[b];
        }
    }
    if (lane >= np) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = pout[(int64_t)r * np + t];
        if (i >= 0) out[i] = ACC ? out[i] + acc[r] : acc[r];
    }
}

// The parent blocks' expansions added to the stored values (a workgroup per parent: its 27 coefficients,
// summed over the source slices, staged in LDS; a lane per patch).
__global__ __launch_bounds__(1024) void modhelm_far_parent_kernel(const double* __restrict__ pxy, int64_t np,
                                                                  const int* __restrict__ pout, double* __restrict__ out,
                                                                  const ApplyParams* __restrict__ prm,
                                                                  const double* __restrict__ head2,
                                                                  const double* __restrict__ coef2, int nslice) {
    __shared__ double2 C[MFAR_P + 1];
    const int64_t par = blockIdx.x;                    // patches [1024 par, 1024 par + 1024)
    if (threadIdx.x <= MFAR_P) {
        const int k = threadIdx.x;
        double cr = 0.0, ci = 0.0;
        for (int sl = 0; sl < nslice; ++sl) {
            cr += coef2[(par * nslice + sl) * MFAR_NCOEF + 2 * k];
            ci += coef2[(par * nslice + sl) * MFAR_NCOEF + 2 * k + 1];
        }
        C[k] = double2{cr, ci};
    }
    __syncthreads();
    const int64_t t = par * 1024 + threadIdx.x;
    if (t >= np) return;
    const double s1 = prm->scale;
    const double* h = head2 + par * MFAR_HDR;
    const double cx = h[0], cy = h[1], rinv = h[2], r2q = h[3];
    double zy[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) zy[b] = (pxy[(int64_t)(4 + b) * np + t] * s1 - cy) * rinv;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double zx = (pxy[(int64_t)a * np + t] * s1 - cx) * rinv;
        double w[4], tm[4], tp[4], vr[4], vi[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            w[b] = r2q * (zx * zx + zy[b] * zy[b]);
            double s1p = 1.0, s0p = 1.0;
#pragma unroll
            for (int n = MFAR_TN; n >= 1; --n) {
                s1p = fma(s1p, w[b] * (1.0 / ((double)n * (MFAR_P + 1 + n))), 1.0);
                s0p = fma(s0p, w[b] * (1.0 / ((double)n * (MFAR_P + n))), 1.0);
            }
            tp[b] = s1p;
            tm[b] = s0p;
            vr[b] = vi[b] = 0.0;
        }
#pragma unroll 2
        for (int m = MFAR_P; m >= 0; --m) {
            const double2 c = C[m];
            const double f = m >= 1 ? 1.0 / ((double)m * (m + 1)) : 0.0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double nr = vr[b] * zx - vi[b] * zy[b];
                vi[b] = fma(vr[b], zy[b], vi[b] * zx) + c.y * tm[b];
                vr[b] = nr + c.x * tm[b];
                const double tn = fma(w[b] * f, tp[b], tm[b]);
                tp[b] = tm[b];
                tm[b] = tn;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = pout[(int64_t)(4 * a + b) * np + t];
            if (i >= 0) out[i] += vr[b];
        }
    }
}

// which: MODE_SLP, MODE_DLP or both (the densities present in the records)
int launch_modhelm_patches_far(ipde_ctx* ctx, const double* rec, int64_t ns, const double* pxy, int64_t np,
                               const int* pout, double* out, const ApplyParams* prm, int which) {
    constexpr int NT = 1024;
    constexpr int NSL = 8;                             // waves (slices of the sources) per parent block
    const int ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    const int64_t ng = ceil_div64(np, 64);
    const int64_t ng2 = ceil_div64(ng, 16);
    const int nch = (int)ceil_div64(ns_pad, 64);
    const size_t nd = (size_t)ng * (MFAR_HDR + MFAR_NCOEF) + (size_t)ng2 * (MFAR_HDR + NSL * MFAR_NCOEF);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial,
                                 nd * sizeof(double) + (size_t)(ng + ng2) * nch * sizeof(unsigned)));
    double* head = (double*)ctx->partial.p;
    double* coef = head + (size_t)ng * MFAR_HDR;
    double* head2 = coef + (size_t)ng * MFAR_NCOEF;
    double* coef2 = head2 + (size_t)ng2 * MFAR_HDR;
    unsigned* near = (unsigned*)(coef2 + (size_t)ng2 * NSL * MFAR_NCOEF);
    unsigned* taken = near + (size_t)ng * nch;
    ipde_time_begin(ctx);
    auto coeffs = [&](auto parent, auto block) {
        hipLaunchKernelGGL(parent, dim3((unsigned)ceil_div64(ng2 * NSL, 4)), dim3(256), 0, ctx->stream, rec, ns_pad, pxy, np,
                           prm, head2, coef2, taken, nch, (const unsigned*)nullptr, NSL);
        hipLaunchKernelGGL(block, dim3((unsigned)ceil_div64(ng, 4)), dim3(256), 0, ctx->stream, rec, ns_pad, pxy, np, prm,
                           head, coef, near, nch, (const unsigned*)taken, 1);
    };
    if (which == MODE_SLP)
        coeffs(modhelm_far_coeff_kernel<16, MODE_SLP>, modhelm_far_coeff_kernel<1, MODE_SLP>);
    else if (which == MODE_DLP)
        coeffs(modhelm_far_coeff_kernel<16, MODE_DLP>, modhelm_far_coeff_kernel<1, MODE_DLP>);
    else
        coeffs(modhelm_far_coeff_kernel<16, MODE_SLP | MODE_DLP>, modhelm_far_coeff_kernel<1, MODE_SLP | MODE_DLP>);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    const size_t lds = (size_t)KT_NKEYS * KT_ENTRY * sizeof(double);
    const double* tab_s = ctx->d_ktab;
    const double* tab_d = ctx->d_ktab + (size_t)KT_NWIN * KT_NKEYS * KT_ENTRY;
    auto patches = [&](auto kern, const double* tab) -> int {
        IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(64 * ng, NT)), dim3(NT), lds, ctx->stream, rec, ns_pad, pxy, np,
                           pout, out, prm, (const double2*)tab, (const double*)head, (const double*)coef,
                           (const unsigned*)near, nch);
        return IPDE_OK;
    };
    if (which & MODE_SLP) {
        IPDE_TRY(patches(modhelm_patch_far_kernel<NT, MODE_SLP, true, false>, tab_s));
        if (which & MODE_DLP) IPDE_TRY(patches(modhelm_patch_far_kernel<NT, MODE_DLP, false, true>, tab_d));
    } else {
        IPDE_TRY(patches(modhelm_patch_far_kernel<NT, MODE_DLP, true, false>, tab_d));
    }
    hipLaunchKernelGGL(modhelm_far_parent_kernel, dim3((unsigned)ng2), dim3(1024), 0, ctx->stream, pxy, np, pout, out,
                       prm, (const double*)head2, (const double*)coef2, NSL);
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// The far-field form for the radial grids of the annuli (ipde_modhelm_apply_columns_far): targets
// (M, N) row-major, column j = the M points of one radial line.  No lattice, but a group of 64
// neighbouring columns is compact (an arc of the annulus): the coefficient kernel above runs unchanged
// on a stand-in patch list (patch j: xs = {min_r x_rj, max_r x_rj, ...}, ys likewise — the column's bounding
// box over all its rows, so ANY (M, N) array gives correct sums; compact blocks make them fast).  One level (64 columns = a
// block; eight source slices per block: there are only N / 64 blocks); a lane owns a column and walks its
// rows four at a time.
__global__ __launch_bounds__(256) void columns_as_patches_kernel(const double* __restrict__ tx,
                                                                 const double* __restrict__ ty, int M, int64_t N,
                                                                 double* __restrict__ pxy) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    // the column's bounding box over ALL its rows (round 3 took rows 0 and M - 1 only: right for a straight
    // radial line, silently wrong for any other (M, N) array — a target outside the block's disc breaks the
    // truncation bound); M * N extra reads, nothing next to the sum
    double x0 = tx[j], x1 = x0, y0 = ty[j], y1 = y0;
    for (int r = 1; r < M; ++r) {
        const double x = tx[(int64_t)r * N + j], y = ty[(int64_t)r * N + j];
        x0 = fmin(x0, x);
        x1 = fmax(x1, x);
        y0 = fmin(y0, y);
        y1 = fmax(y1, y);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        pxy[(int64_t)a * N + j] = (a & 1) ? x1 : x0;
        pxy[(int64_t)(4 + a) * N + j] = (a & 1) ? y1 : y0;
    }
}

// (MODE, FAR, ACC as for modhelm_patch_far_kernel)
template <int NT, int MODE, bool FAR, bool ACC>
__global__ __launch_bounds__(NT) void modhelm_cols_far_kernel(
    const double* __restrict__ rec, int ns_pad, const double* __restrict__ tx, const double* __restrict__ ty, int M,
    int64_t N, double* __restrict__ out, const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab,
    const double* __restrict__ head, const double* __restrict__ coef, int nslice,
    const unsigned* __restrict__ near, int nch) {
    extern __shared__ double2 ltab[];
    const bool nowin = prm->pad >= KT_NWIN;
    const int win = nowin ? KT_NWIN - 1 : prm->pad;
    gtab += (size_t)win * KT_NKEYS * (KT_ENTRY / 2);
    for (unsigned i = threadIdx.x; i < KT_NKEYS * (KT_ENTRY / 2); i += NT) ltab[i] = gtab[i];
    __syncthreads();
    const unsigned key_lo = (unsigned)((1023 + KT_EXP_LO + 2 * win) << KT_B);
    const double s1 = prm->scale;
    // a wave takes four rows of one block of 64 columns (N / 64 blocks alone would leave most of the GPU idle)
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int nrc = (M + 3) / 4;
    const int64_t wid = __builtin_amdgcn_readfirstlane((int)(wv * gridDim.x + blockIdx.x));
    const int64_t g = wid / nrc;
    const int r0 = 4 * (int)(wid - g * nrc);
    if (g * 64 >= N) return;                           // (whole waves, after the only barrier)
    // the block's coefficients (sums over the source slices), parked in LDS behind the table
    double2* C = ltab + KT_NKEYS * (KT_ENTRY / 2) + wv * (MFAR_P + 1);
    if (ln <= MFAR_P) {
        double cr = 0.0, ci = 0.0;
        for (int sl = 0; sl < nslice; ++sl) {
            cr += coef[(g * nslice + sl) * MFAR_NCOEF + 2 * ln];
            ci += coef[(g * nslice + sl) * MFAR_NCOEF + 2 * ln + 1];
        }
        C[ln] = double2{cr, ci};
    }
    __builtin_amdgcn_wave_barrier();
    const double* h = head + g * MFAR_HDR;
    const double cx = h[0], cy = h[1], rinv = h[2], r2q = h[3];
    const int64_t j = g * 64 + ln, jj = min(j, N - 1);
    const unsigned* nm = near + g * nch;
    {
        double x[4], y[4], acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t t = (int64_t)min(r0 + i, M - 1) * N + jj;
            x[i] = tx[t] * s1;
            y[i] = ty[t] * s1;
            acc[i] = 0.0;
        }
        unsigned hmin = 0xFFFFFFFFu;
        if (!nowin) {
            for (int c = 0; c < nch; ++c) {
                unsigned m = nm[c];
                while (m) {
                    const int bt = __builtin_ctz(m);
                    m &= m - 1;
                    SrcRow sx, sy, sq, sa;
                    sx.load(rec, 8 * c + bt, 0);
                    sy.load(rec, 8 * c + bt, 1);
                    sq.load(rec, 8 * c + bt, MODE == MODE_SLP ? 2 : 3);      // q, or a_x
                    if (MODE == MODE_DLP) sa.load(rec, 8 * c + bt, 4);       // a_y
#pragma unroll
                    for (int u = 0; u < IPDE_SRC_PAD; ++u) {
                        double d2[4], ad[4];
                        double2 e[4][KT_READS];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double dx = x[i] - sx.v[u], dy = y[i] - sy.v[u];
                            d2[i] = fma(dy, dy, dx * dx);
                            if (MODE == MODE_DLP) ad[i] = fma(sa.v[u], dy, sq.v[u] * dx);
                            const unsigned hi = (unsigned)__double2hiint(d2[i]);
                            hmin = min(hmin, hi);
                            unsigned idx;
                            asm("v_bfe_u32 %0, %1, 14, 11" : "=v"(idx) : "v"(hi));
                            const double2* ep = (const double2*)((const char*)ltab + __umul24(idx, KT_ENTRY * 8u));
#pragma unroll
                            for (int cc = 0; cc < KT_READS; ++cc) e[i][cc] = ep[cc];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned hi = (unsigned)__double2hiint(d2[i]);
                            const double ctr = __hiloint2double((int)((hi & 0xFFFFC000u) | 0x2000u), 0);
                            const double z = d2[i] - ctr;
                            double p = fma(e[i][2].y, z, e[i][2].x);
                            p = fma(p, z, e[i][1].y);
                            p = fma(p, z, e[i][1].x);
                            p = fma(p, z, e[i][0].y);
                            p = fma(p, z, e[i][0].x);
                            acc[i] = fma(MODE == MODE_SLP ? sq.v[u] : ad[i], p, acc[i]);
                        }
                    }
                }
            }
        }
        if (nowin || (hmin >> KT_SHIFT) < key_lo) {
            double gs[4] = {0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < nch; ++c) {
                unsigned m = nm[c];
                while (m) {
                    const int bt = __builtin_ctz(m);
                    m &= m - 1;
                    modhelm_generic_loop<MODE, false, 4>(rec, 8 * (8 * c + bt), 8 * (8 * c + bt) + 8, x, y, gs);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = gs[i];
        }
        // far sources: Re sum_m C_m zeta^m T_m(w)
        double zx[4], zy[4], w[4], tm[4], tp[4], vr[4], vi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            zx[i] = (x[i] - cx) * rinv;
            zy[i] = (y[i] - cy) * rinv;
            w[i] = r2q * (zx[i] * zx[i] + zy[i] * zy[i]);
            double s1p = 1.0, s0p = 1.0;
#pragma unroll
            for (int n = MFAR_TN; n >= 1; --n) {
                s1p = fma(s1p, w[i] * (1.0 / ((double)n * (MFAR_P + 1 + n))), 1.0);
                s0p = fma(s0p, w[i] * (1.0 / ((double)n * (MFAR_P + n))), 1.0);
            }
            tp[i] = s1p;
            tm[i] = s0p;
            vr[i] = vi[i] = 0.0;
        }
        if (FAR) {
#pragma unroll 2
            for (int m = MFAR_P; m >= 0; --m) {
                const double2 c = C[m];
                const double f = m >= 1 ? 1.0 / ((double)m * (m + 1)) : 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double nr = vr[i] * zx[i] - vi[i] * zy[i];
                    vi[i] = fma(vr[i], zy[i], vi[i] * zx[i]) + c.y * tm[i];
                    vr[i] = nr + c.x * tm[i];
                    const double tn = fma(w[i] * f, tp[i], tm[i]);
                    tp[i] = tm[i];
                    tm[i] = tn;
                }
            }
        }
        if (j < N) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (r0 + i < M) {
                    double* o = out + (int64_t)(r0 + i) * N + j;
                    *o = ACC ? *o + acc[i] + vr[i] : acc[i] + vr[i];
                }
        }
    }
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

// The records of an apply: q' = w_sigma / 2 pi, a' = n w_tau k / 2 pi, coordinates scaled by k (see the file's head).
static void modhelm_pack_args(PackArgs& pa, double k, const double* sx, const double* sy, const double* w_sigma,
                              const double* nx, const double* ny, const double* w_tau) {
    pa.sx = sx;
    pa.sy = sy;
    pa.ch[0] = w_sigma;
    pa.mul[0] = 0.5 / M_PI;
    pa.ch[1] = w_tau ? nx : nullptr;
    pa.mulby[1] = w_tau;
    pa.mul[1] = 0.5 * k / M_PI;
    pa.ch[2] = w_tau ? ny : nullptr;
    pa.mulby[2] = w_tau;
    pa.mul[2] = 0.5 * k / M_PI;
    pa.corr_ch = -1;
    pa.corr2_ch = -1;
    pa.use_scale = 1;
    pa.fixed_scale = k;
}

// Single- and / or double-layer sums (K0(k r) w_sigma + k K1(k r) (n . d) / r w_tau) / (2 pi) onto a patch list whose
// 64-patch groups are 8 x 8 blocks of tiles (ipde_target_plan_build_blocks, pad_blocks = 1): far sources block by
// block in local expansions (Graf's addition theorem; the double layer by the ladder relations of K_m), near
// batches through the tables.
extern "C" int ipde_modhelm_apply_patches_far(ipde_ctx* ctx, double k, int64_t ns, const double* sx,
                                              const double* sy, const double* w_sigma, const double* nx,
                                              const double* ny, const double* w_tau, int64_t np,
                                              const double* pxy, const int32_t* pout, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ns >= 0 && np >= 0 && ns < (1LL << 30) && np < (1LL << 27));
    IPDE_CHECK_ARG(ctx, k > 0.0);
    if (np == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, pxy && pout && out);
    IPDE_CHECK_ARG(ctx, ns > 0 && sx && sy && (w_sigma || w_tau));
    IPDE_CHECK_ARG(ctx, w_tau == nullptr || (nx != nullptr && ny != nullptr));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_ktab) IPDE_TRY(ipde_build_k_table(ctx));
    PackArgs pa{};
    modhelm_pack_args(pa, k, sx, sy, w_sigma, nx, ny, w_tau);
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, pxy, pxy + 4 * np, 4 * np, &rec, &prm));
    return launch_modhelm_patches_far(ctx, rec, ns, pxy, np, pout, out, prm,
                                      (w_sigma ? MODE_SLP : 0) | (w_tau ? MODE_DLP : 0));
}

// The same sums onto an (M, N) radial grid (row-major DEVICE arrays tx, ty: column j = the M points of
// one radial line, neighbouring columns neighbouring lines): blocks of 64 columns, far sources in the
// block's local expansion, near batches through the tables — the radial sums of the solvers' correct()
// (reference ipde/solvers/internals/scalar.py:113-114).
extern "C" int ipde_modhelm_apply_columns_far(ipde_ctx* ctx, double k, int64_t ns, const double* sx,
                                              const double* sy, const double* w_sigma, const double* nx,
                                              const double* ny, const double* w_tau, int M, int64_t N,
                                              const double* tx, const double* ty, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ns >= 0 && N >= 0 && M >= 1 && ns < (1LL << 30) && N < (1LL << 30) && (int64_t)M * N < (1LL << 40));
    IPDE_CHECK_ARG(ctx, k > 0.0);
    if (N == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, tx && ty && out);
    IPDE_CHECK_ARG(ctx, ns > 0 && sx && sy && (w_sigma || w_tau));
    IPDE_CHECK_ARG(ctx, w_tau == nullptr || (nx != nullptr && ny != nullptr));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_ktab) IPDE_TRY(ipde_build_k_table(ctx));
    PackArgs pa{};
    modhelm_pack_args(pa, k, sx, sy, w_sigma, nx, ny, w_tau);
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, tx, ty, (int64_t)M * N, &rec, &prm));
    const int which = (w_sigma ? MODE_SLP : 0) | (w_tau ? MODE_DLP : 0);
    constexpr int NT = 256, NSL = 8;
    const int ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    const int64_t ng = ceil_div64(N, 64);
    const int nch = (int)ceil_div64(ns_pad, 64);
    const size_t nd = (size_t)8 * N + (size_t)ng * (MFAR_HDR + NSL * MFAR_NCOEF);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, nd * sizeof(double) + (size_t)ng * nch * sizeof(unsigned)));
    double* pxy = (double*)ctx->partial.p;
    double* head = pxy + (size_t)8 * N;
    double* coef = head + (size_t)ng * MFAR_HDR;
    unsigned* near = (unsigned*)(coef + (size_t)ng * NSL * MFAR_NCOEF);
    ipde_time_begin(ctx);
    hipLaunchKernelGGL(columns_as_patches_kernel, dim3((unsigned)ceil_div64(N, 256)), dim3(256), 0, ctx->stream, tx, ty,
                       M, N, pxy);
    auto coeffs = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(ng * NSL, 4)), dim3(256), 0, ctx->stream, rec, ns_pad,
                           (const double*)pxy, N, prm, head, coef, near, nch, (const unsigned*)nullptr, NSL);
    };
    if (which == MODE_SLP)
        coeffs(modhelm_far_coeff_kernel<1, MODE_SLP>);
    else if (which == MODE_DLP)
        coeffs(modhelm_far_coeff_kernel<1, MODE_DLP>);
    else
        coeffs(modhelm_far_coeff_kernel<1, MODE_SLP | MODE_DLP>);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    const size_t lds = (size_t)KT_NKEYS * KT_ENTRY * sizeof(double) + (size_t)(NT / 64) * (MFAR_P + 1) * sizeof(double2);
    const double* tab_s = ctx->d_ktab;
    const double* tab_d = ctx->d_ktab + (size_t)KT_NWIN * KT_NKEYS * KT_ENTRY;
    auto cols = [&](auto kern, const double* tab) -> int {
        IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(64 * ng * ((M + 3) / 4), NT)), dim3(NT), lds, ctx->stream, rec,
                           ns_pad, tx, ty, M, N, out, prm, (const double2*)tab, (const double*)head, (const double*)coef,
                           NSL, (const unsigned*)near, nch);
        return IPDE_OK;
    };
    if (which & MODE_SLP) {
        IPDE_TRY(cols(modhelm_cols_far_kernel<NT, MODE_SLP, true, false>, tab_s));
        if (which & MODE_DLP) IPDE_TRY(cols(modhelm_cols_far_kernel<NT, MODE_DLP, false, true>, tab_d));
    } else {
        IPDE_TRY(cols(modhelm_cols_far_kernel<NT, MODE_DLP, true, false>, tab_d));
    }
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}
