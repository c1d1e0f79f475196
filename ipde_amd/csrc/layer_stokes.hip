// Stokes (mu = 1) Stokeslet / stresslet dense sums with pressure.
//
// Record rows:
//   [0] x*s  [1] y*s
//   [2] fx' = wfx/(4 pi)   [3] fy' = wfy/(4 pi)                (Stokeslet density)
//   [4] gx' = s*wdx/pi     [5] gy' = s*wdy/pi                  (stresslet density)
//   [6] nx   [7] ny        [8] n.g'
// With coordinates scaled by s = 2^sh the velocities are scale invariant (given
// g' carries one factor s), the pressures come out divided by s, and the log
// part picks up the constant handled through ApplyParams::corr/corr2.
//
// fp64 VALU bound: ~25 (SLP) / ~24 (DLP) fp64 ops per pair + one ds_read_b128.
#include "layer_pack.h"

namespace {

constexpr int MODE_SLP = 1, MODE_DLP = 2;

struct StokesAcc {
    double uL, vL;  // sum f' * log(d2)
    double u, v, p;
};

struct StokesSrc {
    double fx, fy, gx, gy, nx, ny, ng;
};

template <int MODE>
__device__ __forceinline__ void stokes_pair(double dx, double dy, double L, double rinv,
                                            const StokesSrc& s, StokesAcc& a) {
    if (MODE & MODE_SLP) {
        a.uL = fma(s.fx, L, a.uL);
        a.vL = fma(s.fy, L, a.vL);
        double t = fma(s.fy, dy, s.fx * dx) * rinv;
        a.u = fma(t, dx, a.u);
        a.v = fma(t, dy, a.v);
        a.p += t;  // times 2s at the end: (1/2pi) = 2 * (1/4pi)
    }
    if (MODE & MODE_DLP) {
        double dn = fma(s.ny, dy, s.nx * dx);
        double dg = fma(s.gy, dy, s.gx * dx);
        double w = dn * dg * rinv * rinv;
        a.u = fma(w, dx, a.u);
        a.v = fma(w, dy, a.v);
        // p/2 accumulated: -(n.g)/(2 d2) + (d.n)(d.g)/d2^2
        a.p += fma(-0.5 * s.ng, rinv, w);
    }
}

template <int MODE, bool SKIP, int R>
__device__ __forceinline__ void stokes_generic_loop(const double* __restrict__ rec, int j0, int j1,
                                                    const double (&x)[R], const double (&y)[R],
                                                    StokesAcc (&acc)[R]) {
    for (int j = j0; j < j1; ++j) {
        double sx = rec[ipde_rec_index(j, 0)], sy = rec[ipde_rec_index(j, 1)];
        StokesSrc s;
        s.fx = rec[ipde_rec_index(j, 2)];
        s.fy = rec[ipde_rec_index(j, 3)];
        s.gx = rec[ipde_rec_index(j, 4)];
        s.gy = rec[ipde_rec_index(j, 5)];
        s.nx = rec[ipde_rec_index(j, 6)];
        s.ny = rec[ipde_rec_index(j, 7)];
        s.ng = rec[ipde_rec_index(j, 8)];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double dx = x[r] - sx, dy = y[r] - sy;
            double d2 = fma(dy, dy, dx * dx);
            if (SKIP && d2 == 0.0) continue;
            double L = (MODE & MODE_SLP) ? log(d2) : 0.0;
            stokes_pair<MODE>(dx, dy, L, 1.0 / d2, s, acc[r]);
        }
    }
}

template <int R, int NT>
__device__ __forceinline__ void stokes_store(const StokesAcc (&acc)[R], int64_t base, int64_t nt,
                                             const ApplyParams* __restrict__ prm, double s1,
                                             double* __restrict__ ou, double* __restrict__ ov,
                                             double* __restrict__ op) {
    const bool first = blockIdx.y == 0;
    const double cu = first ? prm->corr : 0.0, cv = first ? prm->corr2 : 0.0;
    const double ps = 2.0 * s1;
    size_t off = (size_t)blockIdx.y * nt;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = base + (int64_t)r * NT;
        if (i < nt) {
            ou[off + i] = fma(-0.5, acc[r].uL + cu, acc[r].u);
            ov[off + i] = fma(-0.5, acc[r].vL + cv, acc[r].v);
            if (op) op[off + i] = ps * acc[r].p;
        }
    }
}

template <int MODE, bool SKIP, int R, int NT>
__global__ __launch_bounds__(NT) void stokes_generic_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ ou, double* __restrict__ ov,
    double* __restrict__ op, const ApplyParams* __restrict__ prm) {
    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = ldexp(1.0, prm->sh);
    double x[R], y[R];
    StokesAcc acc[R];
    int64_t base = (int64_t)blockIdx.x * (R * NT) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + (int64_t)r * NT, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = StokesAcc{0, 0, 0, 0, 0};
    }
    stokes_generic_loop<MODE, SKIP, R>(rec, j0, j1, x, y, acc);
    stokes_store<R, NT>(acc, base, nt, prm, s1, ou, ov, op);
}

template <int MODE, int R, int NT, int U>
__global__ __launch_bounds__(NT) void stokes_table_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ ou, double* __restrict__ ov,
    double* __restrict__ op, const ApplyParams* __restrict__ prm,
    const double2* __restrict__ gtab, unsigned key_lo, unsigned nkeys, int shift) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    TabAddr ta;
    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = ldexp(1.0, prm->sh);
    double x[R], y[R];
    StokesAcc acc[R];
    int64_t base = (int64_t)blockIdx.x * (R * NT) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + (int64_t)r * NT, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = StokesAcc{0, 0, 0, 0, 0};
    }
    // half batches (4 sources): every channel row is one s_load_dwordx8, all issued
    // before the first table lookup of the group (see layer_common.h)
    for (int hb = j0 / 4; hb < j1 / 4; ++hb) {
        const double* row = rec + ((size_t)(hb >> 1) * IPDE_SRC_NCH) * IPDE_SRC_PAD + 4 * (hb & 1);
        double sx[4], sy[4];
        StokesSrc s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sx[u] = row[u];
            sy[u] = row[IPDE_SRC_PAD + u];
            s[u] = StokesSrc{};
            if (MODE & MODE_SLP) {
                s[u].fx = row[2 * IPDE_SRC_PAD + u];
                s[u].fy = row[3 * IPDE_SRC_PAD + u];
            }
            if (MODE & MODE_DLP) {
                s[u].gx = row[4 * IPDE_SRC_PAD + u];
                s[u].gy = row[5 * IPDE_SRC_PAD + u];
                s[u].nx = row[6 * IPDE_SRC_PAD + u];
                s[u].ny = row[7 * IPDE_SRC_PAD + u];
                s[u].ng = row[8 * IPDE_SRC_PAD + u];
            }
        }
#pragma unroll
        for (int u0 = 0; u0 < 4; u0 += U) {
            double dx[U][R], dy[U][R], d2[U][R];
            double2 e[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    dx[u][r] = x[r] - sx[u0 + u];
                    dy[u][r] = y[r] - sy[u0 + u];
                    d2[u][r] = fma(dy[u][r], dy[u][r], dx[u][r] * dx[u][r]);
                    e[u][r] = ta.lookup(ltab, d2[u][r]);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double yy = tab_y(d2[u][r], e[u][r].x);
                    double rinv = rcp_from_y(e[u][r].x, yy);
                    double L = (MODE & MODE_SLP) ? log_from_y(yy, e[u][r].y) : 0.0;
                    stokes_pair<MODE>(dx[u][r], dy[u][r], L, rinv, s[u0 + u], acc[r]);
                }
        }
    }
    if (!ta.all_inside(key_lo) || prm->pad) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = StokesAcc{0, 0, 0, 0, 0};
        stokes_generic_loop<MODE, false, R>(rec, j0, j1, x, y, acc);
    }
    stokes_store<R, NT>(acc, base, nt, prm, s1, ou, ov, op);
}

// Row-run variant of the table kernel (stokeslet, stresslet or both, with pressure), as in
// layer_laplace.hip:
// a lane owns R consecutive targets; when they share x (grid rows; decided per wave)
// dx, dx^2 and f_x dx are formed once per source and lane.  With the 6-instruction
// reciprocal: 24 VALU instructions per pair instead of 27.25.
template <int MODE, int R, bool SHARED>
__device__ __forceinline__ void stokes_rowrun_loop(const double* __restrict__ rec, int j0, int j1,
                                                   const double2* ltab, TabAddr& ta,
                                                   const double (&x)[R], const double (&y)[R],
                                                   StokesAcc (&acc)[R]) {
    for (int hb = j0 / 4; hb < j1 / 4; ++hb) {
        const double* row = rec + ((size_t)(hb >> 1) * IPDE_SRC_NCH) * IPDE_SRC_PAD + 4 * (hb & 1);
        double sx[4], sy[4];
        StokesSrc s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sx[u] = row[u];
            sy[u] = row[IPDE_SRC_PAD + u];
            s[u] = StokesSrc{};
            if (MODE & MODE_SLP) {
                s[u].fx = row[2 * IPDE_SRC_PAD + u];
                s[u].fy = row[3 * IPDE_SRC_PAD + u];
            }
            if (MODE & MODE_DLP) {
                s[u].gx = row[4 * IPDE_SRC_PAD + u];
                s[u].gy = row[5 * IPDE_SRC_PAD + u];
                s[u].nx = row[6 * IPDE_SRC_PAD + u];
                s[u].ny = row[7 * IPDE_SRC_PAD + u];
                s[u].ng = row[8 * IPDE_SRC_PAD + u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double dx[R], dy[R], d2[R], fd[R], dn[R], dg[R];
            double2 e[R];
            if (SHARED) {
                const double dxs = x[0] - sx[u];
                const double dx2 = dxs * dxs;
                const double fxdx = (MODE & MODE_SLP) ? s[u].fx * dxs : 0.0;
                const double nxdx = (MODE & MODE_DLP) ? s[u].nx * dxs : 0.0;
                const double gxdx = (MODE & MODE_DLP) ? s[u].gx * dxs : 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    dx[r] = dxs;
                    dy[r] = y[r] - sy[u];
                    d2[r] = fma(dy[r], dy[r], dx2);
                    if (MODE & MODE_SLP) fd[r] = fma(s[u].fy, dy[r], fxdx);
                    if (MODE & MODE_DLP) {
                        dn[r] = fma(s[u].ny, dy[r], nxdx);
                        dg[r] = fma(s[u].gy, dy[r], gxdx);
                    }
                    e[r] = ta.lookup(ltab, d2[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    dx[r] = x[r] - sx[u];
                    dy[r] = y[r] - sy[u];
                    d2[r] = fma(dy[r], dy[r], dx[r] * dx[r]);
                    if (MODE & MODE_SLP) fd[r] = fma(s[u].fy, dy[r], s[u].fx * dx[r]);
                    if (MODE & MODE_DLP) {
                        dn[r] = fma(s[u].ny, dy[r], s[u].nx * dx[r]);
                        dg[r] = fma(s[u].gy, dy[r], s[u].gx * dx[r]);
                    }
                    e[r] = ta.lookup(ltab, d2[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double yy = tab_y(d2[r], e[r].x);
                const double rinv = rcp_from_y_fast(e[r].x, yy);
                if (MODE & MODE_SLP) {
                    const double t = fd[r] * rinv;
                    const double L = log_from_y(yy, e[r].y);
                    acc[r].uL = fma(s[u].fx, L, acc[r].uL);
                    acc[r].vL = fma(s[u].fy, L, acc[r].vL);
                    acc[r].u = fma(t, dx[r], acc[r].u);
                    acc[r].v = fma(t, dy[r], acc[r].v);
                    acc[r].p += t;
                }
                if (MODE & MODE_DLP) {
                    const double w = dn[r] * dg[r] * rinv * rinv;
                    acc[r].u = fma(w, dx[r], acc[r].u);
                    acc[r].v = fma(w, dy[r], acc[r].v);
                    acc[r].p += fma(-0.5 * s[u].ng, rinv, w);
                }
            }
        }
    }
}

template <int MODE, int R, int NT>
__global__ __launch_bounds__(NT) void stokes_rowrun_kernel(
    const double* __restrict__ rec, int ns_pad, int chunk, const double* __restrict__ tx,
    const double* __restrict__ ty, int64_t nt, double* __restrict__ ou, double* __restrict__ ov,
    double* __restrict__ op, const ApplyParams* __restrict__ prm,
    const double2* __restrict__ gtab, unsigned key_lo, unsigned nkeys, int shift) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    TabAddr ta;
    const int j0 = blockIdx.y * chunk;
    const int j1 = min(ns_pad, j0 + chunk);
    const double s1 = ldexp(1.0, prm->sh);
    double x[R], y[R];
    StokesAcc acc[R];
    const int64_t base = (int64_t)blockIdx.x * (R * NT) + (int64_t)threadIdx.x * R;
    bool same = true;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = min(base + r, nt - 1);
        x[r] = tx[i] * s1;
        y[r] = ty[i] * s1;
        acc[r] = StokesAcc{0, 0, 0, 0, 0};
        same = same && (x[r] == x[0]);
    }
    if (__all(same))
        stokes_rowrun_loop<MODE, R, true>(rec, j0, j1, ltab, ta, x, y, acc);
    else
        stokes_rowrun_loop<MODE, R, false>(rec, j0, j1, ltab, ta, x, y, acc);
    if (!ta.all_inside(key_lo) || prm->pad) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = StokesAcc{0, 0, 0, 0, 0};
        stokes_generic_loop<MODE, false, R>(rec, j0, j1, x, y, acc);
    }
    // consecutive targets per lane: the store helper's strided indexing does not apply
    const bool first = blockIdx.y == 0;
    const double cu = first ? prm->corr : 0.0, cv = first ? prm->corr2 : 0.0;
    const double ps = 2.0 * s1;
    const size_t off = (size_t)blockIdx.y * nt;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t i = base + r;
        if (i < nt) {
            ou[off + i] = fma(-0.5, acc[r].uL + cu, acc[r].u);
            ov[off + i] = fma(-0.5, acc[r].vL + cv, acc[r].v);
            if (op) op[off + i] = ps * acc[r].p;
        }
    }
}

template <int MODE>
int launch_stokes(ipde_ctx* ctx, const double* rec, int64_t ns, const double* tx, const double* ty,
                  int64_t nt, double* ou, double* ov, double* op, const ApplyParams* prm,
                  int flags) {
    const bool generic = (flags & (IPDE_FLAG_GENERIC_MATH | IPDE_FLAG_SKIP_COINCIDENT)) != 0;
    const bool skip = (flags & IPDE_FLAG_SKIP_COINCIDENT) != 0;
    constexpr int NT_TAB = 512, R_TAB = 2, U_TAB = 2;
    const bool rowrun = !generic && ctx->opt_stokes_variant == 1;
    constexpr int NT_RR = 512, R_RR = 4;
    constexpr int NT_GEN = 256, R_GEN = 2;
    const LayerGeom g = ipde_layer_geom(
        ns, nt, generic ? NT_GEN * R_GEN : (rowrun ? NT_RR * R_RR : NT_TAB * R_TAB), ctx->num_cu);
    double *du = ou, *dv = ov, *dp = op;
    if (g.nchunk > 1) {
        size_t per = (size_t)g.nchunk * nt;
        IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, 3 * per * sizeof(double)));
        du = (double*)ctx->partial.p;
        dv = du + per;
        dp = op ? dv + per : nullptr;
    }
    dim3 grid((unsigned)g.gx, (unsigned)g.nchunk);
    ipde_time_begin(ctx);
    if (generic) {
        if (skip)
            hipLaunchKernelGGL((stokes_generic_kernel<MODE, true, R_GEN, NT_GEN>), grid, dim3(NT_GEN), 0,
                               ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, du, dv, dp, prm);
        else
            hipLaunchKernelGGL((stokes_generic_kernel<MODE, false, R_GEN, NT_GEN>), grid, dim3(NT_GEN),
                               0, ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, du, dv, dp, prm);
    } else if (rowrun) {
        const LogTable& lt = ctx->logtab;
        size_t lds = (size_t)lt.nkeys * sizeof(double2);
        IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stokes_rowrun_kernel<MODE, R_RR, NT_RR>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((stokes_rowrun_kernel<MODE, R_RR, NT_RR>), grid, dim3(NT_RR), lds, ctx->stream,
                           rec, g.ns_pad, g.chunk, tx, ty, nt, du, dv, dp, prm,
                           (const double2*)lt.d_tab, (unsigned)lt.key_lo, (unsigned)lt.nkeys,
                           20 - lt.mant_bits);
    } else {
        const LogTable& lt = ctx->logtab;
        size_t lds = (size_t)lt.nkeys * sizeof(double2);
        IPDE_HIP_CHECK(ctx, hipFuncSetAttribute(
                                (const void*)stokes_table_kernel<MODE, R_TAB, NT_TAB, U_TAB>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((stokes_table_kernel<MODE, R_TAB, NT_TAB, U_TAB>), grid, dim3(NT_TAB), lds,
                           ctx->stream, rec, g.ns_pad, g.chunk, tx, ty, nt, du, dv, dp, prm,
                           (const double2*)lt.d_tab, (unsigned)lt.key_lo, (unsigned)lt.nkeys,
                           20 - lt.mant_bits);
    }
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    if (g.nchunk > 1) {
        dim3 g2((unsigned)ceil_div64(nt, 256));
        hipLaunchKernelGGL(ipde_reduce_partials, g2, dim3(256), 0, ctx->stream, (const double*)du,
                           g.nchunk, nt, ou);
        hipLaunchKernelGGL(ipde_reduce_partials, g2, dim3(256), 0, ctx->stream, (const double*)dv,
                           g.nchunk, nt, ov);
        if (op)
            hipLaunchKernelGGL(ipde_reduce_partials, g2, dim3(256), 0, ctx->stream,
                               (const double*)dp, g.nchunk, nt, op);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    return IPDE_OK;
}

}  // namespace

extern "C" int ipde_stokes_apply(ipde_ctx* ctx, int loc, int64_t ns, const double* sx,
                                 const double* sy, const double* wfx, const double* wfy,
                                 const double* nx, const double* ny, const double* wdx,
                                 const double* wdy, int64_t nt, const double* tx, const double* ty,
                                 double* out_u, double* out_v, double* out_p, int flags) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, ns >= 0 && nt >= 0 && ns < (1LL << 30));
    const bool slp = wfx != nullptr && wfy != nullptr;
    const bool dlp = wdx != nullptr && wdy != nullptr;
    IPDE_CHECK_ARG(ctx, (wfx == nullptr) == (wfy == nullptr));
    IPDE_CHECK_ARG(ctx, (wdx == nullptr) == (wdy == nullptr));
    IPDE_CHECK_ARG(ctx, slp || dlp);
    IPDE_CHECK_ARG(ctx, !dlp || (nx != nullptr && ny != nullptr));
    if (nt == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, tx && ty && out_u && out_v);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    double *d_u, *d_v, *d_p;
    IPDE_TRY(ipde_stage_out(ctx, loc, 8, out_u, nt, &d_u));
    IPDE_TRY(ipde_stage_out(ctx, loc, 9, out_v, nt, &d_v));
    IPDE_TRY(ipde_stage_out(ctx, loc, 10, out_p, nt, &d_p));
    if (ns == 0) {
        IPDE_HIP_CHECK(ctx, hipMemsetAsync(d_u, 0, nt * sizeof(double), ctx->stream));
        IPDE_HIP_CHECK(ctx, hipMemsetAsync(d_v, 0, nt * sizeof(double), ctx->stream));
        if (d_p) IPDE_HIP_CHECK(ctx, hipMemsetAsync(d_p, 0, nt * sizeof(double), ctx->stream));
    } else {
        IPDE_CHECK_ARG(ctx, sx && sy);
        const double *d_sx, *d_sy, *d_fx, *d_fy, *d_nx, *d_ny, *d_gx, *d_gy, *d_tx, *d_ty;
        IPDE_TRY(ipde_stage_in(ctx, loc, 0, sx, ns, &d_sx));
        IPDE_TRY(ipde_stage_in(ctx, loc, 1, sy, ns, &d_sy));
        IPDE_TRY(ipde_stage_in(ctx, loc, 2, slp ? wfx : nullptr, ns, &d_fx));
        IPDE_TRY(ipde_stage_in(ctx, loc, 3, slp ? wfy : nullptr, ns, &d_fy));
        IPDE_TRY(ipde_stage_in(ctx, loc, 4, dlp ? nx : nullptr, ns, &d_nx));
        IPDE_TRY(ipde_stage_in(ctx, loc, 5, dlp ? ny : nullptr, ns, &d_ny));
        IPDE_TRY(ipde_stage_in(ctx, loc, 6, dlp ? wdx : nullptr, ns, &d_gx));
        IPDE_TRY(ipde_stage_in(ctx, loc, 7, dlp ? wdy : nullptr, ns, &d_gy));
        IPDE_TRY(ipde_stage_in(ctx, loc, 11, tx, nt, &d_tx));
        IPDE_TRY(ipde_stage_in(ctx, loc, 12, ty, nt, &d_ty));
        PackArgs pa{};
        pa.sx = d_sx;
        pa.sy = d_sy;
        pa.ch[0] = d_fx;
        pa.mul[0] = 0.25 / M_PI;
        pa.ch[1] = d_fy;
        pa.mul[1] = 0.25 / M_PI;
        pa.ch[2] = d_gx;
        pa.mul[2] = 1.0 / M_PI;
        pa.pw[2] = 1;
        pa.ch[3] = d_gy;
        pa.mul[3] = 1.0 / M_PI;
        pa.pw[3] = 1;
        pa.ch[4] = d_nx;
        pa.mul[4] = 1.0;
        pa.ch[5] = d_ny;
        pa.mul[5] = 1.0;
        pa.stokes_ng = dlp ? 1 : 0;
        pa.corr_ch = slp ? 0 : -1;
        pa.corr2_ch = slp ? 1 : -1;
        const bool generic = (flags & (IPDE_FLAG_GENERIC_MATH | IPDE_FLAG_SKIP_COINCIDENT)) != 0;
        pa.use_scale = generic ? 0 : 1;
        pa.exp_hi = ctx->logtab.exp_hi;
        const double* rec;
        const ApplyParams* prm;
        IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, d_tx, d_ty, nt, &rec, &prm));
        int st;
        if (slp && dlp)
            st = launch_stokes<3>(ctx, rec, ns, d_tx, d_ty, nt, d_u, d_v, d_p, prm, flags);
        else if (slp)
            st = launch_stokes<MODE_SLP>(ctx, rec, ns, d_tx, d_ty, nt, d_u, d_v, d_p, prm, flags);
        else
            st = launch_stokes<MODE_DLP>(ctx, rec, ns, d_tx, d_ty, nt, d_u, d_v, d_p, prm, flags);
        IPDE_TRY(st);
    }
    IPDE_TRY(ipde_stage_finish(ctx, loc, 8, out_u, nt));
    IPDE_TRY(ipde_stage_finish(ctx, loc, 9, out_v, nt));
    IPDE_TRY(ipde_stage_finish(ctx, loc, 10, out_p, nt));
    return IPDE_OK;
}
