// Laplace single/double layer dense sums for gfx950.
//
// Record rows: [0] x*s [1] y*s [2] q' = -w_sigma/(4 pi)
//              [3] ax' = s*nx*w_tau/(2 pi) [4] ay' = s*ny*w_tau/(2 pi)
//   out_i = sum_j q'_j log(d2_ij) + (a'_j . d_ij)/d2_ij   (+ corr for the scaling)
//
// Roofline: fp64 VALU bound.  Per source/target pair the table kernel issues
//   4 (dx,dy,d2) + 1 (z) + 4 (log1p poly + T) + 1 (accumulate) fp64 ops for the
//   SLP, + 2 (a.d) + 6 (1/d2 from the same table entry) + 1 for the DLP,
// plus 3 int32 ops (shift, mask, min3/max3 range tracking) and one ds_read_b128
// (the {R, -log R} table entry).  Measured: every VALU instruction, int32 or fp64,
// costs one quad-cycle of the SIMD here, so instruction count is what matters.
// Algorithmic HBM traffic is 24 B per target (read x,y, write u) — irrelevant.
#include "layer_pack.h"

namespace {

constexpr int MODE_SLP = 1, MODE_DLP = 2, MODE_BOTH = 3;

// ---------------------------------------------------------------------------
// generic (libdevice log / IEEE division, no table): reference-grade path, the
// SKIP_COINCIDENT path, and the fallback for table misses
template <int MODE, bool SKIP>
__device__ __forceinline__ double laplace_pair_generic(double dx, double dy, double q, double ax,
                                                       double ay, double acc) {
    double d2 = fma(dy, dy, dx * dx);
    if (SKIP && d2 == 0.0) return acc;
    if (MODE & MODE_SLP) acc = fma(q, log(d2), acc);
    if (MODE & MODE_DLP) {
        double ad = fma(ay, dy, ax * dx);
        acc = fma(ad, 1.0 / d2, acc);
    }
    return acc;
}

template <int MODE, bool SKIP, int R>
__device__ __forceinline__ void laplace_generic_loop(const double* __restrict__ rec, int j0, int j1,
                                                     const double (&x)[R], const double (&y)[R],
                                                     double (&acc)[R]) {
    for (int j = j0; j < j1; ++j) {
        double sx = rec[ipde_rec_index(j, 0)], sy = rec[ipde_rec_index(j, 1)];
        double q = rec[ipde_rec_index(j, 2)];
        double ax = rec[ipde_rec_index(j, 3)], ay = rec[ipde_rec_index(j, 4)];
#pragma unroll
        for (int r = 0; r < R; ++r)
            acc[r] = laplace_pair_generic<MODE, SKIP>(x[r] - sx, y[r] - sy, q, ax, ay, acc[r]);
    }
}

template <int MODE, bool SKIP, int R, int NT>
__global__ __launch_bounds__(NT) void laplace_generic_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ out,
    const ApplyParams* __restrict__ prm) {
    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = ldexp(1.0, prm->sh);
    double x[R], y[R], acc[R];
    int64_t base = (int64_t)blockIdx.x * (R * NT) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + (int64_t)r * NT, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = 0.0;
    }
    laplace_generic_loop<MODE, SKIP, R>(rec, j0, j1, x, y, acc);
    const double corr = (blockIdx.y == 0) ? prm->corr : 0.0;
    double* o = out + (size_t)blockIdx.y * nt;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = base + (int64_t)r * NT;
        if (i < nt) o[i] = acc[r] + corr;
    }
}

// ---------------------------------------------------------------------------
// table kernel: R targets per lane, U sources of a batch in flight -> R*U
// independent lookup + polynomial chains, written phase by phase so that the
// ds_reads of a group are issued back to back and the fp64 chains interleave.
template <int MODE, int R, int NT, int U>
__global__ __launch_bounds__(NT) void laplace_table_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ out,
    const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab, unsigned key_lo,
    unsigned nkeys, int shift) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    TabAddr ta;

    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = ldexp(1.0, prm->sh);
    double x[R], y[R], acc[R];
    int64_t base = (int64_t)blockIdx.x * (R * NT) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + (int64_t)r * NT, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = 0.0;
    }
    for (int b = j0 / IPDE_SRC_PAD; b < j1 / IPDE_SRC_PAD; ++b) {
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
            double dx[U][R], dy[U][R], d2[U][R], z[U][R];
            double2 e[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    dx[u][r] = x[r] - sx.v[u0 + u];
                    dy[u][r] = y[r] - sy.v[u0 + u];
                    d2[u][r] = fma(dy[u][r], dy[u][r], dx[u][r] * dx[u][r]);
                    e[u][r] = ta.lookup(ltab, d2[u][r]);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) z[u][r] = tab_y(d2[u][r], e[u][r].x);
            if (MODE & MODE_SLP) {
                // log_from_y, written phase by phase (z[][] holds y = z/2 here)
                double p[U][R];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r) p[u][r] = fma(z[u][r], -4.0, IPDE_LOG_K3);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r) p[u][r] = fma(z[u][r], p[u][r], -2.0);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r) p[u][r] = fma(z[u][r], p[u][r], IPDE_LOG_K1);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r) p[u][r] = fma(z[u][r], p[u][r], e[u][r].y);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = fma(sq.v[u0 + u], p[u][r], acc[r]);
            }
            if (MODE & MODE_DLP) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        double ad = fma(say.v[u0 + u], dy[u][r], sax.v[u0 + u] * dx[u][r]);
                        acc[r] = fma(ad, rcp_from_y(e[u][r].x, z[u][r]), acc[r]);
                    }
            }
        }
    }
    if (!ta.all_inside(key_lo) || prm->pad) {
        // some pair of this lane fell outside the table (d^2 == 0, tiny or huge):
        // redo the lane's targets with the generic math.  Rare by construction.
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        laplace_generic_loop<MODE, false, R>(rec, j0, j1, x, y, acc);
    }
    const double corr = (blockIdx.y == 0) ? prm->corr : 0.0;
    double* o = out + (size_t)blockIdx.y * nt;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = base + (int64_t)r * NT;
        if (i < nt) o[i] = acc[r] + corr;
    }
}

// Row-run variant of the table kernel (all three modes): every lane owns R CONSECUTIVE entries
// of the target list.  The solver's target lists are grid points in C order, so the R
// targets of a lane normally sit in one grid row and share x: then (x - sx)^2 is formed
// once per source and lane and a pair costs dy, d2 = fma(dy, dy, dx2) — 2 fp64
// instructions instead of 4 (13.1 -> 11.6 VALU instructions per pair at R = 4).  Whether
// all lanes of a wave have that property is decided once per wave (a row boundary or an
// unstructured list takes the general body); results are bitwise those of the strided
// kernel's arithmetic (same dx, dy, d2 values).
template <int MODE, int R, bool SHARED>
__device__ __forceinline__ void laplace_rowrun_loop(const double* __restrict__ rec, int j0, int j1,
                                                    const double2* ltab, TabAddr& ta,
                                                    const double (&x)[R], const double (&y)[R],
                                                    double (&acc)[R]) {
    for (int b = j0 / IPDE_SRC_PAD; b < j1 / IPDE_SRC_PAD; ++b) {
        SrcRow sx, sy, sq, sax, say;
        sx.load(rec, b, 0);
        sy.load(rec, b, 1);
        if (MODE & MODE_SLP) sq.load(rec, b, 2);
        if (MODE & MODE_DLP) {
            sax.load(rec, b, 3);
            say.load(rec, b, 4);
        }
#pragma unroll
        for (int u = 0; u < IPDE_SRC_PAD; ++u) {
            double d2[R], z[R], ad[R];
            double2 e[R];
            if (SHARED) {
                const double dx = x[0] - sx.v[u];
                const double dx2 = dx * dx;
                const double axdx = (MODE & MODE_DLP) ? sax.v[u] * dx : 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double dy = y[r] - sy.v[u];
                    d2[r] = fma(dy, dy, dx2);
                    if (MODE & MODE_DLP) ad[r] = fma(say.v[u], dy, axdx);
                    e[r] = ta.lookup(ltab, d2[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double dx = x[r] - sx.v[u];
                    const double dy = y[r] - sy.v[u];
                    d2[r] = fma(dy, dy, dx * dx);
                    if (MODE & MODE_DLP) ad[r] = fma(say.v[u], dy, sax.v[u] * dx);
                    e[r] = ta.lookup(ltab, d2[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) z[r] = tab_y(d2[r], e[r].x);
            if (MODE & MODE_SLP) {
                double p[R];
#pragma unroll
                for (int r = 0; r < R; ++r) p[r] = fma(z[r], -4.0, IPDE_LOG_K3);
#pragma unroll
                for (int r = 0; r < R; ++r) p[r] = fma(z[r], p[r], -2.0);
#pragma unroll
                for (int r = 0; r < R; ++r) p[r] = fma(z[r], p[r], IPDE_LOG_K1);
#pragma unroll
                for (int r = 0; r < R; ++r) p[r] = fma(z[r], p[r], e[r].y);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fma(sq.v[u], p[r], acc[r]);
            }
            if (MODE & MODE_DLP) {
                // 1/d2 from the same table entry in 6 instructions (3.5e-15, layer_common.h)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fma(ad[r], rcp_from_y_fast(e[r].x, z[r]), acc[r]);
            }
        }
    }
}

template <int MODE, int R, int NT>
__global__ __launch_bounds__(NT) void laplace_rowrun_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ out,
    const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab, unsigned key_lo,
    unsigned nkeys, int shift) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    TabAddr ta;
    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = ldexp(1.0, prm->sh);
    double x[R], y[R], acc[R];
    const int64_t base = (int64_t)blockIdx.x * (R * NT) + (int64_t)threadIdx.x * R;
    bool same = true;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + r, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = 0.0;
        same = same && (x[r] == x[0]);
    }
    if (__all(same))
        laplace_rowrun_loop<MODE, R, true>(rec, j0, j1, ltab, ta, x, y, acc);
    else
        laplace_rowrun_loop<MODE, R, false>(rec, j0, j1, ltab, ta, x, y, acc);
    if (!ta.all_inside(key_lo) || prm->pad) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        laplace_generic_loop<MODE, false, R>(rec, j0, j1, x, y, acc);
    }
    const double corr = (blockIdx.y == 0) ? prm->corr : 0.0;
    double* o = out + (size_t)blockIdx.y * nt;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = base + r;
        if (i < nt) o[i] = acc[r] + corr;
    }
}

// ---------------------------------------------------------------------------
// Patch variant (ipde_laplace_apply_patches): a lane owns a 4 x 4 TENSOR patch of targets,
// (xs[a], ys[b]).  Then d2 = dx2[a] + dy2[b]: the eight squares are formed once per source and
// lane (a sub and a mul each: 1.0 instruction per pair) and a pair's d2 is ONE more instruction
// (row-run: 2.5 in all) — 10.4 VALU instructions per pair against 11.15.  The solver's target
// lists are grids in C order with a band around the curve removed; the host cuts them into 4 x 4
// tiles (ipde_amd/target_plan.py).  A tile the band cut into still runs as a patch: its missing
// points are computed and not stored (pout < 0; ~1 % of the pairs of a grid_pnai list).
// Table range: min over the patch of d2 = min_a dx2 + min_b dy2, tracked through the high
// words (two v_min3/v_min per axis, an add, a min: 6 per 16 pairs).
// Layout: pxy[8][np] (rows 0-3 the xs, 4-7 the ys), pout[16][np] the positions in `out` of
// target (a, b) at row 4a + b: every load and the partial stores are coalesced.
// (one batch of IPDE_SRC_PAD sources against the lane's patch)
template <int MODE>
__device__ __forceinline__ void laplace_patch_block(const double* __restrict__ rec, int b,
                                                    const double2* ltab, TabAddr& ta,
                                                    const double (&xs)[4], const double (&ys)[4],
                                                    double (&acc)[16]) {
    {
        SrcRow sx, sy, sq, sax, say;
        sx.load(rec, b, 0);
        sy.load(rec, b, 1);
        if (MODE & MODE_SLP) sq.load(rec, b, 2);
        if (MODE & MODE_DLP) {
            sax.load(rec, b, 3);
            say.load(rec, b, 4);
        }
#pragma unroll
        for (int u = 0; u < IPDE_SRC_PAD; ++u) {
            double dx2[4], dy2[4], axdx[4], aydy[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const double dx = xs[a] - sx.v[u];
                const double dy = ys[a] - sy.v[u];
                dx2[a] = dx * dx;
                dy2[a] = dy * dy;
                if (MODE & MODE_DLP) {
                    axdx[a] = sax.v[u] * dx;
                    aydy[a] = say.v[u] * dy;
                }
            }
            {
                const unsigned hx = min(min((unsigned)__double2hiint(dx2[0]), (unsigned)__double2hiint(dx2[1])),
                                        min((unsigned)__double2hiint(dx2[2]), (unsigned)__double2hiint(dx2[3])));
                const unsigned hy = min(min((unsigned)__double2hiint(dy2[0]), (unsigned)__double2hiint(dy2[1])),
                                        min((unsigned)__double2hiint(dy2[2]), (unsigned)__double2hiint(dy2[3])));
                const double lo = __hiloint2double((int)hx, 0) + __hiloint2double((int)hy, 0);
                ta.hmin = min(ta.hmin, (unsigned)__double2hiint(lo));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double d2[4], z[4];
                double2 e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d2[r] = dx2[a] + dy2[r];
                    e[r] = ta.lookup_untracked(ltab, d2[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = tab_y(d2[r], e[r].x);
                if (MODE & MODE_SLP) {
                    double p[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = fma(z[r], -4.0, IPDE_LOG_K3);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = fma(z[r], p[r], -2.0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = fma(z[r], p[r], IPDE_LOG_K1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = fma(z[r], p[r], e[r].y);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[4 * a + r] = fma(sq.v[u], p[r], acc[4 * a + r]);
                }
                if (MODE & MODE_DLP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[4 * a + r] = fma(axdx[a] + aydy[r], rcp_from_y_fast(e[r].x, z[r]), acc[4 * a + r]);
                }
            }
        }
    }
}

template <int MODE>
__device__ __forceinline__ void laplace_patch_loop(const double* __restrict__ rec, int j0, int j1,
                                                   const double2* ltab, TabAddr& ta,
                                                   const double (&xs)[4], const double (&ys)[4],
                                                   double (&acc)[16]) {
    for (int b = j0 / IPDE_SRC_PAD; b < j1 / IPDE_SRC_PAD; ++b) laplace_patch_block<MODE>(rec, b, ltab, ta, xs, ys, acc);
}

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void laplace_patch_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ pxy, int64_t np,
    const int* __restrict__ pout, double* __restrict__ out, double* __restrict__ partial,
    const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab, unsigned key_lo,
    unsigned nkeys) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    TabAddr ta;
    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = ldexp(1.0, prm->sh);
    const int64_t lane = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int64_t t = min(lane, np - 1);
    double xs[4], ys[4], acc[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        xs[a] = pxy[(int64_t)a * np + t] * s1;
        ys[a] = pxy[(int64_t)(4 + a) * np + t] * s1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0;
    laplace_patch_loop<MODE>(rec, j0, j1, ltab, ta, xs, ys, acc);
    if (!ta.all_inside(key_lo) || prm->pad) {
        // a pair of this patch may have left the table: its rows again with the generic math
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double xa[4] = {xs[a], xs[a], xs[a], xs[a]};
            double g[4] = {0.0, 0.0, 0.0, 0.0};
            laplace_generic_loop<MODE, false, 4>(rec, j0, j1, xa, ys, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[4 * a + r] = g[r];
        }
    }
    if (lane >= np) return;
    const double corr = (blockIdx.y == 0) ? prm->corr : 0.0;
    if (partial) {
#pragma unroll
        for (int r = 0; r < 16; ++r) partial[((int64_t)blockIdx.y * 16 + r) * np + t] = acc[r] + corr;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = pout[(int64_t)r * np + t];
            if (i >= 0) out[i] = acc[r] + corr;
        }
    }
}

// out[pout[i]] = sum over the source chunks, in chunk order
__global__ __launch_bounds__(256) void laplace_patch_reduce(const double* __restrict__ partial, int nchunk,
                                                            int64_t n16, const int* __restrict__ pout,
                                                            double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    double s = partial[i];
    for (int c = 1; c < nchunk; ++c) s += partial[(int64_t)c * n16 + i];
    if (pout[i] >= 0) out[pout[i]] = s;
}

template <int MODE>
int launch_laplace_patches(ipde_ctx* ctx, const double* rec, int64_t ns, const double* pxy, int64_t np,
                           const int* pout, double* out, const ApplyParams* prm) {
    constexpr int NT = 1024;
    const LogTable& lt = ctx->logtab;
    const LayerGeom g = ipde_layer_geom(ns, 16 * np, 16 * NT, ctx->num_cu);
    double* partial = nullptr;
    if (g.nchunk > 1) {
        IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, (size_t)g.nchunk * 16 * np * sizeof(double)));
        partial = (double*)ctx->partial.p;
    }
    const size_t lds = (size_t)lt.nkeys * sizeof(double2);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)laplace_patch_kernel<MODE, NT>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ipde_time_begin(ctx);
    hipLaunchKernelGGL((laplace_patch_kernel<MODE, NT>), dim3((unsigned)g.gx, (unsigned)g.nchunk), dim3(NT), lds,
                       ctx->stream, rec, g.ns_pad, g.chunk, pxy, np, pout, out, partial, prm,
                       (const double2*)lt.d_tab, (unsigned)lt.key_lo, (unsigned)lt.nkeys);
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    if (partial) {
        hipLaunchKernelGGL(laplace_patch_reduce, dim3((unsigned)ceil_div64(16 * np, 256)), dim3(256), 0,
                           ctx->stream, (const double*)partial, g.nchunk, 16 * np, pout, out);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// Patches with the far sources in local expansions (ipde_laplace_apply_patches_far).
//
// The 64 patches of a wave are one 8 x 8 block of tiles (a plan built with padded blocks,
// target_plan.hip): 32 x 32 grid points with centre c and half-diagonal r.  For a source z_j with
// |z_j - c| >= r / FAR_RHO every target z of the block has |t| <= FAR_RHO, t = (z - c) v,
// v = 1 / (z_j - c), and with delta = z - z_j = (c - z_j)(1 - t):
//     q log|delta|^2        = q log|c - z_j|^2 - 2 q Re sum_{k>=1} t^k / k
//     (a . d) / |d|^2       = Re(alpha / delta) = -Re sum_{k>=0} alpha v^(k+1) (z - c)^k
// so the block's far sources collapse into FAR_P + 1 complex coefficients
//     B_k = -(2 / k) sum_j q_j v_j^k - sum_j alpha_j v_j^(k+1)        (B_0: sum_j q_j log|c - z_j|^2 - ...)
// and a target costs FAR_P complex Horner steps instead of one table logarithm per source.  The
// truncation: FAR_RHO^(FAR_P+1) / ((FAR_P + 1)(1 - FAR_RHO)) = 3e-18 of sum|q_j| (0.25, 26 terms) —
// below the rounding of the direct sum.  (Everything is kept in the block's own units: vt = r v,
// zeta = (z - c) / r, so no power over- or underflows whatever the coordinate scale is.)
// Sources nearer than that go through the table kernel's body, batch by batch: the coefficient
// kernel leaves a bit per batch of eight sources and block.  2048^2 targets x 4096 sources: ~1 % of
// the pairs are near ones; 3.3 ms -> see DESIGN.md.
constexpr int FAR_P = 26;
constexpr double FAR_RHO = 0.25;
constexpr int FAR_NCOEF = 2 * (FAR_P + 2);      // doubles per block and kind: k = 0 .. FAR_P + 1, complex
constexpr int FAR_HDR = 4;                      // cx, cy, 1 / r, (spare)

// WHICH = MODE_SLP: head[g] = {cx, cy, 1/r, 0}; cs[g][2k], cs[g][2k+1] = sum q vt^k (k >= 1), cs[g][0] = sum q log d^2
// WHICH = MODE_DLP: cd[g][2k], [2k+1] = sum (alpha / r) vt^k, k = 1 .. FAR_P + 1
// near[g][chunk]: bit b set = batch 8 chunk + b has a source nearer than r / FAR_RHO: the whole batch is summed directly
// Two levels: PPL = 16 runs first, a wave per PARENT block (sixteen consecutive blocks: a 4 x 4 group in
// the plan's Z order) — the batches all of whose sources are beyond 4 parent radii enter the parent's
// coefficients and get a bit in `bits`; PPL = 1 then runs per block with `skip` = those bits (parent g / 16):
// such batches are neither near nor far for the block, the rest as before (`bits` = the near batches).
template <int WHICH, int PPL>
__global__ __launch_bounds__(256) void laplace_far_coeff_kernel(const double* __restrict__ rec, int ns_pad,
                                                               const double* __restrict__ pxy, int64_t np,
                                                               const ApplyParams* __restrict__ prm,
                                                               double* __restrict__ head, double* __restrict__ coef,
                                                               unsigned* __restrict__ near, int nch, int write_near,
                                                               const unsigned* __restrict__ skip, int nslice) {
    // (parent level: `nslice` waves per parent, each over a slice of the sources — 256 parents alone
    // would leave most of the GPU idle; the slices' sums are added up, in order, where they are used)
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t g = gw / nslice;
    const int slice = (int)(gw - g * nslice);
    if (g * 64 * PPL >= np) return;                    // (whole waves)
    const int trips = (ns_pad + 127) / 128, tps = (trips + nslice - 1) / nslice;
    const int jlo = slice * tps * 128, jhi = min(ns_pad, (slice + 1) * tps * 128);
    const double s1 = ldexp(1.0, prm->sh);
    FarBlock blk;
    blk.init<PPL>(pxy, np, g, lane, s1);
    const double cx = blk.cx, cy = blk.cy, r = blk.r, r2 = blk.r2;
    const double thr = r2 * (1.0 / (FAR_RHO * FAR_RHO)) * (1.0 + 0x1p-40);
    constexpr int K1 = WHICH == MODE_SLP ? FAR_P : FAR_P + 1;
    double sre[K1 + 1], sim[K1 + 1];
#pragma unroll
    for (int k = 0; k <= K1; ++k) sre[k] = sim[k] = 0.0;
    // two sources per lane and trip: two independent power chains in flight
    for (int j0 = jlo; j0 < jhi; j0 += 128) {
        if (PPL == 1 && skip) {
            // a trip whose sixteen batches the parent block took whole is nobody's here: the far arcs of the curve
            // are long, so most trips of most blocks end at these two scalar loads (the blocks' pass visited every
            // source once per block: O(N_s N_blocks) loads and distance tests for ~1 % of them in an expansion)
            const unsigned t0 = skip[(g >> 4) * nch + (j0 >> 6)];
            const unsigned t1 = j0 + 64 < ns_pad ? skip[(g >> 4) * nch + (j0 >> 6) + 1] : 0xFFu;
            if ((t0 & 0xFFu) == 0xFFu && (t1 & 0xFFu) == 0xFFu) {
                if (write_near && lane == 0) {
                    near[g * nch + (j0 >> 6)] = 0u;
                    if (j0 + 64 < ns_pad) near[g * nch + (j0 >> 6) + 1] = 0u;
                }
                continue;
            }
        }
        double vre[2], vim[2], wre[2], wim[2];
        bool anyfar = false;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int jb = j0 + 64 * h;                    // (wave-uniform)
            const int j = jb + lane;
            const bool valid = j < ns_pad;
            const int jj = valid ? j : ns_pad - 1;
            const double dx = rec[ipde_rec_index(jj, 0)] - cx, dy = rec[ipde_rec_index(jj, 1)] - cy;
            const double d2 = fma(dy, dy, dx * dx);
            // a batch of eight sources goes one way as a whole: into the expansion only if all eight are far
            // (batches the parent block took are nobody's here)
            const unsigned taken = (PPL == 1 && skip && jb < ns_pad) ? skip[(g >> 4) * nch + (jb >> 6)] : 0u;
            const bool mine = valid && !((taken >> (lane >> 3)) & 1u);
            const unsigned long long m = __ballot(mine && !(d2 >= thr && !prm->pad));
            const unsigned long long mv = __ballot(valid);
            const bool far = mine && ((m >> (lane & ~7)) & 0xFFull) == 0;
            anyfar = anyfar || far;
            if (write_near && lane == 0 && jb < ns_pad) {
                const unsigned nearbits = far_batch_bits(m);
                // parent level: the bits of the batches it takes (all of a batch's sources far)
                near[g * nch + (jb >> 6)] = PPL == 1 ? nearbits : (far_batch_bits(mv) & ~nearbits);
            }
            // near sources ride along with zero weight at a harmless position
            const double inv = far ? r / d2 : 0.0;
            vre[h] = dx * inv;
            vim[h] = -dy * inv;                            // vt = r / (z_j - c)
            if (WHICH == MODE_SLP) {
                wre[h] = far ? rec[ipde_rec_index(jj, 2)] : 0.0;
                wim[h] = 0.0;
                if (__ballot(far)) sre[0] = fma(wre[h], log(far ? d2 : 1.0), sre[0]);
            } else {
                wre[h] = far ? rec[ipde_rec_index(jj, 3)] / r : 0.0;
                wim[h] = far ? rec[ipde_rec_index(jj, 4)] / r : 0.0;
            }
        }
        if (__ballot(anyfar) == 0) continue;               // nothing of this trip enters the expansion (wave-uniform)
        double pre[2] = {vre[0], vre[1]}, pim[2] = {vim[0], vim[1]};      // vt^k
#pragma unroll
        for (int k = 1; k <= K1; ++k) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (WHICH == MODE_SLP) {
                    sre[k] = fma(wre[h], pre[h], sre[k]);
                    sim[k] = fma(wre[h], pim[h], sim[k]);
                } else {
                    sre[k] += wre[h] * pre[h] - wim[h] * pim[h];
                    sim[k] += wre[h] * pim[h] + wim[h] * pre[h];
                }
                const double nre = pre[h] * vre[h] - pim[h] * vim[h];
                pim[h] = fma(pre[h], vim[h], pim[h] * vre[h]);
                pre[h] = nre;
            }
        }
    }
#pragma unroll
    for (int k = 0; k <= K1; ++k) {
        sre[k] = wave_sum(sre[k]);
        sim[k] = wave_sum(sim[k]);
    }
    if (lane == 0) {
        if ((WHICH == MODE_SLP || write_near) && slice == 0) {
            head[g * FAR_HDR + 0] = cx;
            head[g * FAR_HDR + 1] = cy;
            head[g * FAR_HDR + 2] = 1.0 / r;
            head[g * FAR_HDR + 3] = 0.0;
        }
        double* c = coef + gw * FAR_NCOEF;
#pragma unroll
        for (int k = 0; k <= K1; ++k) {
            c[2 * k] = sre[k];
            c[2 * k + 1] = sim[k];
        }
    }
}

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void laplace_patch_far_kernel(
    const double* __restrict__ rec, int ns_pad, const double* __restrict__ pxy, int64_t np,
    const int* __restrict__ pout, double* __restrict__ out, const ApplyParams* __restrict__ prm,
    const double2* __restrict__ gtab, unsigned key_lo, unsigned nkeys, const double* __restrict__ head,
    const double* __restrict__ cs, const double* __restrict__ cdl, const unsigned* __restrict__ near, int nch,
    const double* __restrict__ head2, const double* __restrict__ cs2, const double* __restrict__ cdl2,
    int nslice2) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    TabAddr ta;
    const double s1 = ldexp(1.0, prm->sh);
    // wave w of workgroup b takes block w * gridDim.x + b: the sixteen waves of a workgroup (one CU)
    // are spread over the list — consecutive blocks are neighbours in space, and a workgroup of
    // sixteen neighbours next to the curve would carry sixteen times the near work of one far from it
    const int wv = threadIdx.x >> 6;
    const int64_t g = __builtin_amdgcn_readfirstlane((int)(wv * gridDim.x + blockIdx.x));
    if (g * 64 >= np) return;                          // (whole waves, after the only barrier)
    const int64_t lane = g * 64 + (threadIdx.x & 63);
    const int64_t t = min(lane, np - 1);
    // the wave's expansion coefficients, combined (B_k = -(2/k) S_k - D_(k+1), B_0 real) and parked in LDS
    // behind the table: the Horner loops read them by broadcast ds_read_b128 (as scalar loads from HBM each
    // step was an exposed round trip: +100 us at two levels)
    double2* wc = ltab + nkeys + wv * (2 * (FAR_P + 1));
    {
        const int k = threadIdx.x & 63;
#pragma unroll
        for (int level = 0; level < 2; ++level) {
            const int ns = level == 0 ? 1 : nslice2;          // (the parent's sums come in slices)
            const int64_t gl = level == 0 ? g : (g >> 4) * ns;
            const double* S = (level == 0 ? cs : cs2) + gl * FAR_NCOEF;
            const double* D = (level == 0 ? cdl : cdl2) + gl * FAR_NCOEF;
            if (k <= FAR_P) {
                double bre = 0.0, bim = 0.0;
                for (int sl = 0; sl < ns; ++sl) {
                    const double* Ss = S + (size_t)sl * FAR_NCOEF;
                    const double* Ds = D + (size_t)sl * FAR_NCOEF;
                    if (k == 0) {
                        if (MODE & MODE_SLP) bre += Ss[0];
                        if (MODE & MODE_DLP) bre -= Ds[2];
                    } else {
                        if (MODE & MODE_SLP) {
                            const double f = -2.0 / (double)k;
                            bre += f * Ss[2 * k];
                            bim += f * Ss[2 * k + 1];
                        }
                        if (MODE & MODE_DLP) {
                            bre -= Ds[2 * (k + 1)];
                            bim -= Ds[2 * (k + 1) + 1];
                        }
                    }
                }
                wc[level * (FAR_P + 1) + k] = double2{bre, bim};
            }
        }
    }
    double xs[4], ys[4], acc[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        xs[a] = pxy[(int64_t)a * np + t] * s1;
        ys[a] = pxy[(int64_t)(4 + a) * np + t] * s1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0;
    // near sources: the table body, batch by batch
    const unsigned* nm = near + g * nch;
    for (int c = 0; c < nch; ++c) {
        unsigned m = nm[c];
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            laplace_patch_block<MODE>(rec, 8 * c + b, ltab, ta, xs, ys, acc);
        }
    }
    if (!ta.all_inside(key_lo) || prm->pad) {
        // a pair of this patch may have left the table (or the scaling failed: then no source is in
        // an expansion either): all its sources again with the generic math
        const bool everything = prm->pad != 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double xa[4] = {xs[a], xs[a], xs[a], xs[a]};
            double gsum[4] = {0.0, 0.0, 0.0, 0.0};
            if (everything) {
                laplace_generic_loop<MODE, false, 4>(rec, 0, ns_pad, xa, ys, gsum);
            } else {
                for (int c = 0; c < nch; ++c) {
                    unsigned m = nm[c];
                    while (m) {
                        const int b = __builtin_ctz(m);
                        m &= m - 1;
                        laplace_generic_loop<MODE, false, 4>(rec, 8 * (8 * c + b), 8 * (8 * c + b) + 8, xa, ys, gsum);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[4 * a + r] = gsum[r];
        }
    }
    // far sources: Re sum_k B_k zeta^k by Horner — the block's own expansion (level 0) and its parent's
    // (level 1: the sources beyond four parent radii)
    __builtin_amdgcn_wave_barrier();
    if (!prm->pad) {
#pragma unroll 1
        for (int level = 0; level < 2; ++level) {
            const int64_t gl = level == 0 ? g : (g >> 4);
            const double* h = (level == 0 ? head : head2) + gl * FAR_HDR;
            const double cx = h[0], cy = h[1], rinv = h[2];
            const double2* B = wc + level * (FAR_P + 1);
            const double b0 = B[0].x;
            double zy[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) zy[bb] = (ys[bb] - cy) * rinv;
            // a row of the patch at a time: eight chain registers instead of thirty-two (the kernel runs at
            // 128 VGPRs: sixteen waves share the table)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const double zx = (xs[a] - cx) * rinv;
                double vre[4] = {0.0, 0.0, 0.0, 0.0}, vim[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
                for (int k = FAR_P; k >= 1; --k) {
                    const double2 bk = B[k];
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        // v = (v + B_k) zeta
                        const double ure = vre[bb] + bk.x, uim = vim[bb] + bk.y;
                        vre[bb] = ure * zx - uim * zy[bb];
                        vim[bb] = fma(ure, zy[bb], uim * zx);
                    }
                }
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) acc[4 * a + bb] += vre[bb] + b0;
            }
        }
    }
    if (lane >= np) return;
    const double corr = prm->corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = pout[(int64_t)r * np + t];
        if (i >= 0) out[i] = acc[r] + corr;
    }
}

template <int MODE>
int launch_laplace_patches_far(ipde_ctx* ctx, const double* rec, int64_t ns, const double* pxy, int64_t np,
                               const int* pout, double* out, const ApplyParams* prm) {
    constexpr int NT = 1024;
    const LogTable& lt = ctx->logtab;
    const int ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    const int64_t ng = ceil_div64(np, 64);
    const int64_t ng2 = ceil_div64(ng, 16);            // parent blocks
    const int nch = (int)ceil_div64(ns_pad, 64);
    constexpr int NSL = 8;                             // waves (slices of the sources) per parent
    // workspace: blocks: head | S | D; parents: head | S, D per slice; then the near bits and the parents' bits
    const size_t nd = (size_t)ng * (FAR_HDR + 2 * FAR_NCOEF) + (size_t)ng2 * (FAR_HDR + 2 * NSL * FAR_NCOEF);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial,
                                 nd * sizeof(double) + (size_t)(ng + ng2) * nch * sizeof(unsigned)));
    double* head = (double*)ctx->partial.p;
    double* cs = head + (size_t)ng * FAR_HDR;
    double* cdl = cs + (size_t)ng * FAR_NCOEF;
    double* head2 = cdl + (size_t)ng * FAR_NCOEF;
    double* cs2 = head2 + (size_t)ng2 * FAR_HDR;
    double* cdl2 = cs2 + (size_t)ng2 * NSL * FAR_NCOEF;
    unsigned* near = (unsigned*)(cdl2 + (size_t)ng2 * NSL * FAR_NCOEF);
    unsigned* taken = near + (size_t)ng * nch;
    const unsigned gb = (unsigned)ceil_div64(ng, 4), gb2 = (unsigned)ceil_div64(ng2 * NSL, 4);
    // (option "timing_split": an event pair per stage — parents' coefficients, blocks' coefficients, patches —
    // instead of one around the three; bench.py's per-kernel roofline figures)
    const bool split = ctx->opt_timing_split != 0;
    ipde_time_begin(ctx);
    // parents first (their bits steer the blocks' pass)
    if (MODE & MODE_SLP)
        hipLaunchKernelGGL((laplace_far_coeff_kernel<MODE_SLP, 16>), dim3(gb2), dim3(256), 0, ctx->stream, rec, ns_pad,
                           pxy, np, prm, head2, cs2, taken, nch, 1, (const unsigned*)nullptr, NSL);
    if (MODE & MODE_DLP)
        hipLaunchKernelGGL((laplace_far_coeff_kernel<MODE_DLP, 16>), dim3(gb2), dim3(256), 0, ctx->stream, rec, ns_pad,
                           pxy, np, prm, head2, cdl2, taken, nch, (MODE & MODE_SLP) ? 0 : 1, (const unsigned*)nullptr,
                           NSL);
    if (split) {
        ipde_time_end(ctx);
        ipde_time_begin(ctx);
    }
    if (MODE & MODE_SLP)
        hipLaunchKernelGGL((laplace_far_coeff_kernel<MODE_SLP, 1>), dim3(gb), dim3(256), 0, ctx->stream, rec, ns_pad,
                           pxy, np, prm, head, cs, near, nch, 1, (const unsigned*)taken, 1);
    if (MODE & MODE_DLP)
        hipLaunchKernelGGL((laplace_far_coeff_kernel<MODE_DLP, 1>), dim3(gb), dim3(256), 0, ctx->stream, rec, ns_pad,
                           pxy, np, prm, head, cdl, near, nch, (MODE & MODE_SLP) ? 0 : 1, (const unsigned*)taken, 1);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    if (split) {
        ipde_time_end(ctx);
        ipde_time_begin(ctx);
    }
    // the table, then 2 levels x 27 combined coefficients for each of the workgroup's waves
    const size_t lds = ((size_t)lt.nkeys + (size_t)(NT / 64) * 2 * (FAR_P + 1)) * sizeof(double2);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)laplace_patch_far_kernel<MODE, NT>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((laplace_patch_far_kernel<MODE, NT>), dim3((unsigned)ceil_div64(64 * ng, NT)), dim3(NT), lds,
                       ctx->stream, rec, ns_pad, pxy, np, pout, out, prm, (const double2*)lt.d_tab,
                       (unsigned)lt.key_lo, (unsigned)lt.nkeys, (const double*)head, (const double*)cs,
                       (const double*)cdl, (const unsigned*)near, nch, (const double*)head2, (const double*)cs2,
                       (const double*)cdl2, NSL);
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// The far-field form for the radial grids of the annuli (ipde_laplace_apply_columns_far; the scheme and
// the stand-in patch list are described at modhelm_cols_far_kernel in layer_modhelm.hip): targets (M, N)
// row-major, column j = one radial line; blocks of 64 columns, one level, eight source slices per block; a
// wave takes four rows of a block.  Single layer only (what the solvers' correct() sums).
__global__ __launch_bounds__(256) void laplace_columns_as_patches_kernel(const double* __restrict__ tx,
                                                                         const double* __restrict__ ty, int M,
                                                                         int64_t N, double* __restrict__ pxy) {
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

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void laplace_cols_far_kernel(
    const double* __restrict__ rec, int ns_pad, const double* __restrict__ tx, const double* __restrict__ ty, int M,
    int64_t N, double* __restrict__ out, const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab,
    unsigned key_lo, unsigned nkeys, const double* __restrict__ head, const double* __restrict__ cs,
    const double* __restrict__ cdl, int nslice, const unsigned* __restrict__ near, int nch) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    const double s1 = ldexp(1.0, prm->sh);
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int nrc = (M + 3) / 4;
    const int64_t wid = __builtin_amdgcn_readfirstlane((int)(wv * gridDim.x + blockIdx.x));
    const int64_t g = wid / nrc;
    const int r0 = 4 * (int)(wid - g * nrc);
    if (g * 64 >= N) return;                           // (whole waves, after the only barrier)
    // B_k = -(2/k) S_k - D_(k+1) (B_0 = S_0 - Re D_1), the slices' sums added up, parked in LDS behind the table
    double2* B = ltab + nkeys + wv * (FAR_P + 1);
    if (ln <= FAR_P) {
        double br = 0.0, bi = 0.0;
        for (int sl = 0; sl < nslice; ++sl) {
            if (MODE & MODE_SLP) {
                const double f = ln == 0 ? 1.0 : -2.0 / (double)ln;
                br += f * cs[(g * nslice + sl) * FAR_NCOEF + 2 * ln];
                if (ln) bi += f * cs[(g * nslice + sl) * FAR_NCOEF + 2 * ln + 1];
            }
            if (MODE & MODE_DLP) {
                br -= cdl[(g * nslice + sl) * FAR_NCOEF + 2 * (ln + 1)];
                if (ln) bi -= cdl[(g * nslice + sl) * FAR_NCOEF + 2 * (ln + 1) + 1];
            }
        }
        B[ln] = double2{br, bi};
    }
    __builtin_amdgcn_wave_barrier();
    const double* h = head + g * FAR_HDR;
    const double cx = h[0], cy = h[1], rinv = h[2];
    const int64_t j = g * 64 + ln, jj = min(j, N - 1);
    double x[4], y[4], acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = (int64_t)min(r0 + i, M - 1) * N + jj;
        x[i] = tx[t] * s1;
        y[i] = ty[t] * s1;
    }
    TabAddr ta;
    const unsigned* nm = near + g * nch;
    for (int c = 0; c < nch; ++c) {
        unsigned m = nm[c];
        while (m) {
            const int bt = __builtin_ctz(m);
            m &= m - 1;
            SrcRow sx, sy, sq, sax, say;
            sx.load(rec, 8 * c + bt, 0);
            sy.load(rec, 8 * c + bt, 1);
            if (MODE & MODE_SLP) sq.load(rec, 8 * c + bt, 2);
            if (MODE & MODE_DLP) {
                sax.load(rec, 8 * c + bt, 3);
                say.load(rec, 8 * c + bt, 4);
            }
#pragma unroll
            for (int u = 0; u < IPDE_SRC_PAD; ++u) {
                double d2[4], ad[4];
                double2 e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double dx = x[i] - sx.v[u], dy = y[i] - sy.v[u];
                    d2[i] = fma(dy, dy, dx * dx);
                    if (MODE & MODE_DLP) ad[i] = fma(say.v[u], dy, sax.v[u] * dx);
                    e[i] = ta.lookup(ltab, d2[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double yy = tab_y(d2[i], e[i].x);
                    if (MODE & MODE_SLP) acc[i] = fma(sq.v[u], log_from_y(yy, e[i].y), acc[i]);
                    if (MODE & MODE_DLP) acc[i] = fma(ad[i], rcp_from_y_fast(e[i].x, yy), acc[i]);
                }
            }
        }
    }
    if (!ta.all_inside(key_lo) || prm->pad) {
        double gs[4] = {0.0, 0.0, 0.0, 0.0};
        if (prm->pad) {
            laplace_generic_loop<MODE, false, 4>(rec, 0, ns_pad, x, y, gs);
        } else {
            for (int c = 0; c < nch; ++c) {
                unsigned m = nm[c];
                while (m) {
                    const int bt = __builtin_ctz(m);
                    m &= m - 1;
                    laplace_generic_loop<MODE, false, 4>(rec, 8 * (8 * c + bt), 8 * (8 * c + bt) + 8, x, y, gs);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = gs[i];
    }
    if (!prm->pad) {
        double zx[4], zy[4], vre[4] = {0.0, 0.0, 0.0, 0.0}, vim[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            zx[i] = (x[i] - cx) * rinv;
            zy[i] = (y[i] - cy) * rinv;
        }
#pragma unroll 2
        for (int k = FAR_P; k >= 1; --k) {
            const double2 bk = B[k];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double ure = vre[i] + bk.x, uim = vim[i] + bk.y;
                vre[i] = ure * zx[i] - uim * zy[i];
                vim[i] = fma(ure, zy[i], uim * zx[i]);
            }
        }
        const double b0 = B[0].x;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += vre[i] + b0;
    }
    if (j < N) {
        const double corr = (MODE & MODE_SLP) ? prm->corr : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + i < M) out[(int64_t)(r0 + i) * N + j] = acc[i] + corr;
    }
}

template <int MODE, int R, int NT>
int launch_rowrun_variant(ipde_ctx* ctx, dim3 grid, const double* rec, const LayerGeom& g,
                          const double* tx, const double* ty, int64_t nt, double* dst,
                          const ApplyParams* prm) {
    const LogTable& lt = ctx->logtab;
    size_t lds = (size_t)lt.nkeys * sizeof(double2);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)laplace_rowrun_kernel<MODE, R, NT>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((laplace_rowrun_kernel<MODE, R, NT>), grid, dim3(NT), lds, ctx->stream, rec, g.ns_pad,
                       g.chunk, tx, ty, nt, dst, prm, (const double2*)lt.d_tab, (unsigned)lt.key_lo,
                       (unsigned)lt.nkeys, 20 - lt.mant_bits);
    return IPDE_OK;
}

template <int MODE, int R, int NT, int U>
int launch_table_variant(ipde_ctx* ctx, dim3 grid, const double* rec, const LayerGeom& g,
                         const double* tx, const double* ty, int64_t nt, double* dst,
                         const ApplyParams* prm) {
    const LogTable& lt = ctx->logtab;
    size_t lds = (size_t)lt.nkeys * sizeof(double2);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)laplace_table_kernel<MODE, R, NT, U>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((laplace_table_kernel<MODE, R, NT, U>), grid, dim3(NT), lds, ctx->stream, rec,
                       g.ns_pad, g.chunk, tx, ty, nt, dst, prm, (const double2*)lt.d_tab,
                       (unsigned)lt.key_lo, (unsigned)lt.nkeys, 20 - lt.mant_bits);
    return IPDE_OK;
}

template <int MODE>
int launch_laplace(ipde_ctx* ctx, const double* rec, int64_t ns, const double* tx,
                   const double* ty, int64_t nt, double* out, const ApplyParams* prm, int flags) {
    const bool generic = (flags & (IPDE_FLAG_GENERIC_MATH | IPDE_FLAG_SKIP_COINCIDENT)) != 0;
    const bool skip = (flags & IPDE_FLAG_SKIP_COINCIDENT) != 0;
    int variant = generic ? -1 : ctx->opt_laplace_variant;
    int NTv = 256, Rv = 2;
    switch (variant) {
        case 0: NTv = 512; Rv = 4; break;
        case 1: NTv = 1024; Rv = 4; break;
        case 2: NTv = 512; Rv = 8; break;
        case 3: NTv = 1024; Rv = 2; break;
        case 4: NTv = 768; Rv = 4; break;
        case 5: NTv = 512; Rv = 4; break;
        case 6: NTv = 1024; Rv = 4; break;
        case 7: NTv = 1024; Rv = 2; break;
        case 8: NTv = 1024; Rv = 3; break;
        case 9: NTv = 1024; Rv = 4; break;   // row-run
        case 10: NTv = 512; Rv = 8; break;   // row-run, 8 per lane
        case 11: NTv = 1024; Rv = 6; break;  // row-run, 6 per lane
        case 12: NTv = 512; Rv = 4; break;
        case 13: NTv = 1024; Rv = 3; break;
        case 14: NTv = 1024; Rv = 5; break;
        case 15: NTv = 768; Rv = 4; break;
        default: break;
    }
    const LayerGeom g = ipde_layer_geom(ns, nt, NTv * Rv, ctx->num_cu);
    double* dst = out;
    if (g.nchunk > 1) {
        IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, (size_t)g.nchunk * nt * sizeof(double)));
        dst = (double*)ctx->partial.p;
    }
    dim3 grid((unsigned)g.gx, (unsigned)g.nchunk);
    ipde_time_begin(ctx);
    if (generic) {
        if (skip)
            hipLaunchKernelGGL((laplace_generic_kernel<MODE, true, 2, 256>), grid, dim3(256), 0,
                               ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, dst, prm);
        else
            hipLaunchKernelGGL((laplace_generic_kernel<MODE, false, 2, 256>), grid, dim3(256), 0,
                               ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, dst, prm);
    } else {
        int st;
        switch (variant) {
            case 1: st = launch_table_variant<MODE, 4, 1024, 1>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 2: st = launch_table_variant<MODE, 8, 512, 1>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 3: st = launch_table_variant<MODE, 2, 1024, 2>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 4: st = launch_table_variant<MODE, 4, 768, 2>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 5: st = launch_table_variant<MODE, 4, 512, 1>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 6: st = launch_table_variant<MODE, 4, 1024, 2>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 7: st = launch_table_variant<MODE, 2, 1024, 4>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 8: st = launch_table_variant<MODE, 3, 1024, 2>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 9: st = launch_rowrun_variant<MODE, 4, 1024>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 10: st = launch_rowrun_variant<MODE, 8, 512>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 11: st = launch_rowrun_variant<MODE, 6, 1024>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 12: st = launch_rowrun_variant<MODE, 4, 512>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 13: st = launch_rowrun_variant<MODE, 3, 1024>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 14: st = launch_rowrun_variant<MODE, 5, 1024>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            case 15: st = launch_rowrun_variant<MODE, 4, 768>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
            default: st = launch_table_variant<MODE, 4, 512, 2>(ctx, grid, rec, g, tx, ty, nt, dst, prm); break;
        }
        IPDE_TRY(st);
    }
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    if (g.nchunk > 1) {
        hipLaunchKernelGGL(ipde_reduce_partials, dim3((unsigned)ceil_div64(nt, 256)), dim3(256), 0,
                           ctx->stream, (const double*)dst, g.nchunk, nt, out);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    return IPDE_OK;
}

}  // namespace

extern "C" int ipde_laplace_apply(ipde_ctx* ctx, int loc, int64_t ns, const double* sx,
                                  const double* sy, const double* w_sigma, const double* nx,
                                  const double* ny, const double* w_tau, int64_t nt,
                                  const double* tx, const double* ty, double* out, int flags) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, ns >= 0 && nt >= 0 && ns < (1LL << 30));
    IPDE_CHECK_ARG(ctx, w_sigma != nullptr || w_tau != nullptr);
    IPDE_CHECK_ARG(ctx, w_tau == nullptr || (nx != nullptr && ny != nullptr));
    if (nt == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, tx && ty && out);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
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
    pa.mul[0] = -0.25 / M_PI;
    pa.ch[1] = d_nx;
    pa.mulby[1] = d_tau;
    pa.mul[1] = 0.5 / M_PI;
    pa.pw[1] = 1;
    pa.ch[2] = d_ny;
    pa.mulby[2] = d_tau;
    pa.mul[2] = 0.5 / M_PI;
    pa.pw[2] = 1;
    pa.corr_ch = 0;
    pa.corr2_ch = -1;
    const bool generic = (flags & (IPDE_FLAG_GENERIC_MATH | IPDE_FLAG_SKIP_COINCIDENT)) != 0;
    pa.use_scale = generic ? 0 : 1;
    pa.exp_hi = ctx->logtab.exp_hi;
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, d_tx, d_ty, nt, &rec, &prm));
    int mode = (w_sigma ? MODE_SLP : 0) | (w_tau ? MODE_DLP : 0);
    int st;
    if (mode == MODE_SLP)
        st = launch_laplace<MODE_SLP>(ctx, rec, ns, d_tx, d_ty, nt, d_out, prm, flags);
    else if (mode == MODE_DLP)
        st = launch_laplace<MODE_DLP>(ctx, rec, ns, d_tx, d_ty, nt, d_out, prm, flags);
    else
        st = launch_laplace<MODE_BOTH>(ctx, rec, ns, d_tx, d_ty, nt, d_out, prm, flags);
    IPDE_TRY(st);
    return ipde_stage_finish(ctx, loc, 7, out, nt);
}

extern "C" int ipde_laplace_apply_patches(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                          const double* w_sigma, const double* nx, const double* ny,
                                          const double* w_tau, int64_t np, const double* pxy,
                                          const int32_t* pout, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ns >= 0 && np >= 0 && ns < (1LL << 30) && np < (1LL << 27));
    IPDE_CHECK_ARG(ctx, w_sigma != nullptr || w_tau != nullptr);
    IPDE_CHECK_ARG(ctx, w_tau == nullptr || (nx != nullptr && ny != nullptr));
    if (np == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, pxy && pout && out);
    IPDE_CHECK_ARG(ctx, ns > 0 && sx && sy);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    PackArgs pa{};
    pa.sx = sx;
    pa.sy = sy;
    pa.ch[0] = w_sigma;
    pa.mul[0] = -0.25 / M_PI;
    pa.ch[1] = w_tau ? nx : nullptr;
    pa.mulby[1] = w_tau;
    pa.mul[1] = 0.5 / M_PI;
    pa.pw[1] = 1;
    pa.ch[2] = w_tau ? ny : nullptr;
    pa.mulby[2] = w_tau;
    pa.mul[2] = 0.5 / M_PI;
    pa.pw[2] = 1;
    pa.corr_ch = 0;
    pa.corr2_ch = -1;
    pa.use_scale = 1;
    pa.exp_hi = ctx->logtab.exp_hi;
    const double* rec;
    const ApplyParams* prm;
    // the patches' xs and ys are rows 0-3 and 4-7 of pxy: the bounding box of the 4 np + 4 np
    // coordinates is the bounding box of the targets
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, pxy, pxy + 4 * np, 4 * np, &rec, &prm));
    const int mode = (w_sigma ? MODE_SLP : 0) | (w_tau ? MODE_DLP : 0);
    if (mode == MODE_SLP) return launch_laplace_patches<MODE_SLP>(ctx, rec, ns, pxy, np, pout, out, prm);
    if (mode == MODE_DLP) return launch_laplace_patches<MODE_DLP>(ctx, rec, ns, pxy, np, pout, out, prm);
    return launch_laplace_patches<MODE_BOTH>(ctx, rec, ns, pxy, np, pout, out, prm);
}

// The same sum with the far sources of every 64-patch block in a local expansion (see
// laplace_far_coeff_kernel): the plan must come from ipde_target_plan_build_blocks(..., 8, 8, ...,
// pad_blocks = 1), 64 consecutive patches = one block of tiles.
extern "C" int ipde_laplace_apply_patches_far(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                              const double* w_sigma, const double* nx, const double* ny,
                                              const double* w_tau, int64_t np, const double* pxy,
                                              const int32_t* pout, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ns >= 0 && np >= 0 && ns < (1LL << 30) && np < (1LL << 27));
    IPDE_CHECK_ARG(ctx, w_sigma != nullptr || w_tau != nullptr);
    IPDE_CHECK_ARG(ctx, w_tau == nullptr || (nx != nullptr && ny != nullptr));
    if (np == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, pxy && pout && out);
    IPDE_CHECK_ARG(ctx, ns > 0 && sx && sy);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    PackArgs pa{};
    pa.sx = sx;
    pa.sy = sy;
    pa.ch[0] = w_sigma;
    pa.mul[0] = -0.25 / M_PI;
    pa.ch[1] = w_tau ? nx : nullptr;
    pa.mulby[1] = w_tau;
    pa.mul[1] = 0.5 / M_PI;
    pa.pw[1] = 1;
    pa.ch[2] = w_tau ? ny : nullptr;
    pa.mulby[2] = w_tau;
    pa.mul[2] = 0.5 / M_PI;
    pa.pw[2] = 1;
    pa.corr_ch = 0;
    pa.corr2_ch = -1;
    pa.use_scale = 1;
    pa.exp_hi = ctx->logtab.exp_hi;
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, pxy, pxy + 4 * np, 4 * np, &rec, &prm));
    const int mode = (w_sigma ? MODE_SLP : 0) | (w_tau ? MODE_DLP : 0);
    if (mode == MODE_SLP) return launch_laplace_patches_far<MODE_SLP>(ctx, rec, ns, pxy, np, pout, out, prm);
    if (mode == MODE_DLP) return launch_laplace_patches_far<MODE_DLP>(ctx, rec, ns, pxy, np, pout, out, prm);
    return launch_laplace_patches_far<MODE_BOTH>(ctx, rec, ns, pxy, np, pout, out, prm);
}

// the records of an apply: q = -w_sigma / 4 pi (times log d^2), a = n w_tau / 2 pi (scaled with the coordinates)
static void laplace_pack_args(ipde_ctx* ctx, PackArgs& pa, const double* sx, const double* sy, const double* w_sigma,
                              const double* nx, const double* ny, const double* w_tau) {
    pa.sx = sx;
    pa.sy = sy;
    pa.ch[0] = w_sigma;
    pa.mul[0] = -0.25 / M_PI;
    pa.ch[1] = w_tau ? nx : nullptr;
    pa.mulby[1] = w_tau;
    pa.mul[1] = 0.5 / M_PI;
    pa.pw[1] = 1;
    pa.ch[2] = w_tau ? ny : nullptr;
    pa.mulby[2] = w_tau;
    pa.mul[2] = 0.5 / M_PI;
    pa.pw[2] = 1;
    pa.corr_ch = 0;
    pa.corr2_ch = -1;
    pa.use_scale = 1;
    pa.exp_hi = ctx->logtab.exp_hi;
}

// Single- and / or double-layer sums onto an (M, N) radial grid (row-major DEVICE arrays; column j = one radial
// line): the radial sums of the solvers' correct() (reference ipde/solvers/internals/scalar.py:113-114) with the far
// sources of every block of 64 lines in a local expansion.
template <int MODE>
static int launch_laplace_columns_far(ipde_ctx* ctx, const double* rec, int64_t ns, int M, int64_t N, const double* tx,
                                      const double* ty, double* out, const ApplyParams* prm) {
    constexpr int NT = 256, NSL = 8;
    const LogTable& lt = ctx->logtab;
    const int ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    const int64_t ng = ceil_div64(N, 64);
    const int nch = (int)ceil_div64(ns_pad, 64);
    const size_t nd = (size_t)8 * N + (size_t)ng * (FAR_HDR + 2 * NSL * FAR_NCOEF);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, nd * sizeof(double) + (size_t)ng * nch * sizeof(unsigned)));
    double* pxy = (double*)ctx->partial.p;
    double* head = pxy + (size_t)8 * N;
    double* cs = head + (size_t)ng * FAR_HDR;
    double* cdl = cs + (size_t)ng * NSL * FAR_NCOEF;
    unsigned* near = (unsigned*)(cdl + (size_t)ng * NSL * FAR_NCOEF);
    ipde_time_begin(ctx);
    hipLaunchKernelGGL(laplace_columns_as_patches_kernel, dim3((unsigned)ceil_div64(N, 256)), dim3(256), 0, ctx->stream,
                       tx, ty, M, N, pxy);
    const unsigned gb = (unsigned)ceil_div64(ng * NSL, 4);
    if (MODE & MODE_SLP)
        hipLaunchKernelGGL((laplace_far_coeff_kernel<MODE_SLP, 1>), dim3(gb), dim3(256), 0, ctx->stream, rec, ns_pad,
                           (const double*)pxy, N, prm, head, cs, near, nch, 1, (const unsigned*)nullptr, NSL);
    if (MODE & MODE_DLP)
        hipLaunchKernelGGL((laplace_far_coeff_kernel<MODE_DLP, 1>), dim3(gb), dim3(256), 0, ctx->stream, rec, ns_pad,
                           (const double*)pxy, N, prm, head, cdl, near, nch, (MODE & MODE_SLP) ? 0 : 1,
                           (const unsigned*)nullptr, NSL);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    const size_t lds = ((size_t)lt.nkeys + (size_t)(NT / 64) * (FAR_P + 1)) * sizeof(double2);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)laplace_cols_far_kernel<MODE, NT>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((laplace_cols_far_kernel<MODE, NT>), dim3((unsigned)ceil_div64(64 * ng * ((M + 3) / 4), NT)), dim3(NT),
                       lds, ctx->stream, rec, ns_pad, tx, ty, M, N, out, prm, (const double2*)lt.d_tab,
                       (unsigned)lt.key_lo, (unsigned)lt.nkeys, (const double*)head, (const double*)cs, (const double*)cdl,
                       NSL, (const unsigned*)near, nch);
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_laplace_apply_columns_far(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                              const double* w_sigma, const double* nx, const double* ny,
                                              const double* w_tau, int M, int64_t N, const double* tx,
                                              const double* ty, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ns >= 0 && N >= 0 && M >= 1 && ns < (1LL << 30) && N < (1LL << 30) && (int64_t)M * N < (1LL << 40));
    if (N == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, tx && ty && out);
    IPDE_CHECK_ARG(ctx, ns > 0 && sx && sy && (w_sigma || w_tau));
    IPDE_CHECK_ARG(ctx, w_tau == nullptr || (nx != nullptr && ny != nullptr));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    PackArgs pa{};
    laplace_pack_args(ctx, pa, sx, sy, w_sigma, nx, ny, w_tau);
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, tx, ty, (int64_t)M * N, &rec, &prm));
    const int mode = (w_sigma ? MODE_SLP : 0) | (w_tau ? MODE_DLP : 0);
    if (mode == MODE_SLP) return launch_laplace_columns_far<MODE_SLP>(ctx, rec, ns, M, N, tx, ty, out, prm);
    if (mode == MODE_DLP) return launch_laplace_columns_far<MODE_DLP>(ctx, rec, ns, M, N, tx, ty, out, prm);
    return launch_laplace_columns_far<MODE_BOTH>(ctx, rec, ns, M, N, tx, ty, out, prm);
}
