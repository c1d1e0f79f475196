// Shared device-side pieces of the layer-potential kernels (gfx950).
//
// Data layout
//   sources : "batch-SoA" records.  Sources are grouped in batches of
//             IPDE_SRC_PAD = 8; a batch holds IPDE_SRC_NCH channel rows of 8
//             doubles each (row 0 = x, row 1 = y, rows 2.. = kernel-specific
//             densities):  rec[(batch*NCH + ch)*8 + u].
//             The source index is wave-uniform, so one channel row of a batch is
//             ONE s_load_dwordx16 into SGPRs: no LDS / VGPR traffic for the source
//             side, and no SMEM in flight while the LDS table lookups of the batch
//             are being consumed (SMEM shares lgkmcnt with the LDS and returns out
//             of order, which would force lgkmcnt(0) before every lookup).
//             The pack kernel pads to whole batches (zero density, coordinates of
//             the last real source) so hot loops have no tail.
//   targets : SoA tx[], ty[]; lane l of block b owns targets
//             (b*R + r)*NT + l, r = 0..R-1  -> fully coalesced loads/stores.
//   LDS     : the log/reciprocal table, one copy per workgroup.
#pragma once
#include "ipde_common.h"

#define IPDE_SRC_NCH 10
#define IPDE_SRC_PAD 8

__host__ __device__ __forceinline__ size_t ipde_rec_index(int64_t j, int ch) {
    return ((size_t)(j >> 3) * IPDE_SRC_NCH + ch) * IPDE_SRC_PAD + (size_t)(j & 7);
}

// Per-call parameters produced on the device by the pack kernel (no host sync).
struct ApplyParams {
    int sh;        // coordinates are scaled by 2^sh so that all d^2 < 2^exp_hi
    int pad;       // modified Helmholtz: index of the table window (layer_modhelm.hip);
                   // Laplace / Stokes: 1 = the scaling could not bring the pairs under the top of
                   // the table (coordinates beyond 2^400): use the generic kernel body
    double corr;   // constant added to every output (undoes the log scaling)
    double corr2;  // second constant (Stokes v component)
    double inv_s;  // 2^-sh
    double inv_s2; // 2^-2sh
    double scale;  // coordinate scale actually applied (2^sh, or PackArgs::fixed_scale)
};

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------
// Shared by the far-field forms (ipde_{laplace,modhelm,stokes}_apply_patches_far): the block of
// patches a wave owns — patches [64 g, 64 g + 64) of pxy[8][np], coordinates times `scale` — as a
// disc: centre of the bounding box, half-diagonal r.  Whatever the grouping of the plan, every
// target of the wave lies within r of the centre.
struct FarBlock {
    double cx, cy, r, r2;
    // PPL patches per lane: the wave owns patches [64 PPL g, 64 PPL (g + 1)) (PPL = 16: a parent block
    // of sixteen consecutive blocks)
    template <int PPL = 1>
    __device__ __forceinline__ void init(const double* __restrict__ pxy, int64_t np, int64_t g, int lane,
                                         double scale) {
        double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const int64_t t = min((g * PPL + i) * 64 + lane, np - 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const double x = pxy[(int64_t)a * np + t] * scale, y = pxy[(int64_t)(4 + a) * np + t] * scale;
                xlo = fmin(xlo, x);
                xhi = fmax(xhi, x);
                ylo = fmin(ylo, y);
                yhi = fmax(yhi, y);
            }
        }
        xlo = wave_min(xlo);
        xhi = wave_max(xhi);
        ylo = wave_min(ylo);
        yhi = wave_max(yhi);
        cx = 0.5 * (xlo + xhi);
        cy = 0.5 * (ylo + yhi);
        const double hx = 0.5 * (xhi - xlo), hy = 0.5 * (yhi - ylo);
        r2 = hx * hx + hy * hy;
        r = sqrt(r2);
    }
};
// one bit per batch of eight lanes: set where the 64-bit lane mask has any lane of the batch
__device__ __forceinline__ unsigned far_batch_bits(unsigned long long m) {
    unsigned bits = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) bits |= ((m >> (8 * b)) & 0xFFull) ? (1u << b) : 0u;
    return bits;
}

// ---------------------------------------------------------------------------
// Table-driven log / reciprocal.  For every key = hi32(x) >> 12 (sign, exponent and
// the top IPDE_TAB_B = 8 mantissa bits of x) of the 32 covered binades the table
// holds {Rh, T} with R ~ 1/centre(interval), Rh = R/2, T = -log R, stored at
// position (key mod 8192).  With y = x*Rh - 1/2 (= z/2, z = x*R - 1, |z| <= 2^-9):
//   log(x) = T + log1p(2y),   1/x = 2 Rh / (1 + 2y).
// log1p is the degree-5 Taylor polynomial economised to degree 4 on |z| <= a = 2^-9
// (z^5 -> (20 a^2 z^3 - 5 a^4 z)/16; max error a^5/80 = 3.6e-16 absolute, one
// rounding of a log value of O(1..10)):
//   log1p(z) ~ z (1 - a^4/16) - z^2/2 + z^3 (1/3 + a^2/4) - z^4/4
// written in y so that every Horner step is ONE v_fma_f64 whose constants are
// inline constants (-4, -2, -0.5) or a single SGPR pair (gfx9 constant-bus limit 1).
#define IPDE_TAB_B 8
#define IPDE_TAB_SHIFT (20 - IPDE_TAB_B)
#define IPDE_TAB_BINADES 32
#define IPDE_TAB_NKEYS (IPDE_TAB_BINADES << IPDE_TAB_B)

__device__ __forceinline__ double tab_y(double x, double Rh) { return fma(x, Rh, -0.5); }

#define IPDE_LOG_K3 (8.0 * (1.0 / 3.0 + 0x1p-20))
#define IPDE_LOG_K1 (2.0 * (1.0 - 0x1p-40))

__device__ __forceinline__ double log_from_y(double y, double T) {
    double p = fma(y, -4.0, IPDE_LOG_K3);
    p = fma(y, p, -2.0);
    p = fma(y, p, IPDE_LOG_K1);
    return fma(y, p, T);
}

// 1/x = 2 Rh (1 - z)(1 + z^2)(1 + z^4) + O(z^8), z = 2y
__device__ __forceinline__ double rcp_from_y(double Rh, double y) {
    double g = fma(y, -4.0, 2.0);  // 2 (1 - z)
    double z = y + y;
    double z2 = z * z;
    g = fma(z2, g, g);
    double z4 = z2 * z2;
    g = fma(z4, g, g);
    return Rh * g;
}

// 1/x to 3.5e-15 relative in 6 instructions: 2 Rh (1 - z)(1 + z^2 + z^4), error z^6 <= 2^-48.
// Enough wherever the result carries a 1e-10..1e-12 tolerance (the Stokes kernels).
__device__ __forceinline__ double rcp_from_y_fast(double Rh, double y) {
    double g = fma(y, -4.0, 2.0);  // 2 (1 - z)
    double z = y + y;
    double z2 = z * z;
    double h = fma(z2, z2, z2);    // z^2 + z^4
    g = fma(g, h, g);
    return Rh * g;
}

// Table addressing: one v_bfe_u32 + one v_lshl_add_u32, plus min3/max3 tracking of
// hi32(x) so that the covered range is validated ONCE per lane after the source loop.
struct TabAddr {
    // Only the LOWER end of the table needs watching: the pack kernel scales the coordinates
    // so that the bounding-box diagonal — hence every pair — stays below the top covered
    // binade (with a 2^-30 relative margin for the roundings), and flags the launch for the
    // generic kernel when it cannot (ApplyParams::pad).  Tracking a minimum alone is one
    // v_min3_u32 per TWO pairs.
    unsigned hmin;
    __device__ __forceinline__ TabAddr() : hmin(0xFFFFFFFFu) {}
    __device__ __forceinline__ double2 lookup(const double2* ltab, double x) {
        unsigned hi = (unsigned)__double2hiint(x);
        hmin = min(hmin, hi);
        // hipcc lowers the C form to v_lshrrev + v_and (+ v_lshl_add for the address);
        // the bit-field extract is one instruction.  Plain VALU op: no wait counters,
        // VALU->VALU dependencies are hardware-interlocked.
        unsigned idx;
        static_assert(IPDE_TAB_SHIFT == 12 && IPDE_TAB_B + 5 == 13, "literal operands below");
        asm("v_bfe_u32 %0, %1, 12, 13" : "=v"(idx) : "v"(hi));
        return ltab[idx];
    }
    // the same read without the range tracking (callers that bound x themselves)
    __device__ __forceinline__ double2 lookup_untracked(const double2* ltab, double x) const {
        unsigned idx;
        asm("v_bfe_u32 %0, %1, 12, 13" : "=v"(idx) : "v"((unsigned)__double2hiint(x)));
        return ltab[idx];
    }
    // every x seen so far was at or above the lowest covered binade?  (NaN / Inf inputs give
    // NaN / Inf results through the arithmetic itself; the 13-bit index keeps any read
    // inside the table)
    __device__ __forceinline__ bool all_inside(unsigned key_lo) const {
        return (hmin >> IPDE_TAB_SHIFT) >= key_lo;
    }
};

// Launch geometry shared by the three kernel families: big target sets give every
// lane R targets and every block all sources; small target sets split the
// sources over blockIdx.y (partials are summed in a fixed order afterwards).
struct LayerGeom {
    int64_t gx;
    int nchunk;
    int chunk;   // sources per chunk, multiple of IPDE_SRC_PAD
    int ns_pad;  // ns rounded up to whole batches
};
static inline LayerGeom ipde_layer_geom(int64_t ns, int64_t nt, int nt_per_block, int num_cu) {
    LayerGeom g;
    g.gx = ceil_div64(nt, nt_per_block);
    // One block per CU at a time (the LDS table), so a launch runs in ceil(blocks / CUs) rounds
    // of (sources per block) each.  Few target blocks — interface-sized lists, or 1/8 of the
    // 2048^2 list on one of 8 GPUs: 126 blocks on 256 CUs — leave CUs idle or end in a nearly
    // empty round; splitting the sources nchunk ways turns that into rounds of 1/nchunk length.
    // Pick the split with the least (rounds x length), 0.5 % per extra block for its table load
    // and the partial-sum pass.
    int nchunk = 1;
    if (g.gx < 8 * (int64_t)num_cu) {
        const int cmax = (int)std::min<int64_t>(512, std::max<int64_t>(1, ceil_div64(ns, 64)));
        double best = 1e300;
        for (int c = 1; c <= cmax; ++c) {
            const double rounds = (double)ceil_div64(g.gx * c, num_cu);
            const double cost = rounds * (1.0 / c + 0.005);
            if (cost < best * (1.0 - 1e-9)) {
                best = cost;
                nchunk = c;
            }
        }
    }
    g.ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    g.chunk = (int)(ceil_div64(ceil_div64(ns, nchunk), IPDE_SRC_PAD) * IPDE_SRC_PAD);
    g.nchunk = (int)ceil_div64(ns, g.chunk);
    return g;
}
