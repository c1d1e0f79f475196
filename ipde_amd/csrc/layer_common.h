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
    int pad;
    double corr;   // constant added to every output (undoes the log scaling)
    double corr2;  // second constant (Stokes v component)
    double inv_s;  // 2^-sh
    double inv_s2; // 2^-2sh
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
// Table-driven log: log(x) = T + log1p(z), z = x*R - 1, |z| <= 2^-(B+1).
// The entry {R, T} is one ds_read_b128.  Degree-5 Taylor of log1p(z)/z is
// accurate to z^6/6 <= 2^-54/6 for B = 8.
__device__ __forceinline__ double log1p_poly5(double z, double T) {
    double p = fma(z, 0.2, -0.25);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, -0.5);
    p = fma(p, z, 1.0);
    return fma(p, z, T);
}

// 1/x from the same table entry: 1/x = R/(1+z), |z| <= 2^-9:
//   1/(1+z) = (1-z)(1+z^2)(1+z^4) + O(z^8)
__device__ __forceinline__ double tab_rcp_from(double R, double z) {
    double g = fma(z, -1.0, 1.0);
    double z2 = z * z;
    g = fma(z2, g, g);
    double z4 = z2 * z2;
    g = fma(z4, g, g);
    return R * g;
}

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
    int nchunk = 1;
    const int64_t want_blocks = 2 * (int64_t)num_cu;
    if (g.gx < want_blocks) {
        nchunk = (int)std::min<int64_t>(ceil_div64(want_blocks, g.gx), ceil_div64(ns, 64));
        if (nchunk < 1) nchunk = 1;
        if (nchunk > 1024) nchunk = 1024;
    }
    g.ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    g.chunk = (int)(ceil_div64(ceil_div64(ns, nchunk), IPDE_SRC_PAD) * IPDE_SRC_PAD);
    g.nchunk = (int)ceil_div64(ns, g.chunk);
    return g;
}
