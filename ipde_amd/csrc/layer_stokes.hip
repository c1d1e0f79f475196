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

// ---------------------------------------------------------------------------
// Stokeslet sums onto patch lists with the far sources of every 8 x 8 block of tiles in a local
// expansion (ipde_stokes_apply_patches_far; the Laplace form of this is in layer_laplace.hip, where
// the scheme is described).  With F = f'_x + i f'_y, delta = z - z_j, U = u + i v:
//     U = sum_j  -F log|delta| + F/2 + (conj F / 2) delta / conj(delta),     sum_j (f'.d)/d^2 = Re sum_j F / delta
// and for a source beyond 4 block radii r of the block's centre c, vt = r / (z_j - c), zeta = (z - c)/r:
//     log|delta| = log|c - z_j| - Re sum_{k>=1} (zeta vt)^k / k,     1/delta = -(vt/r) sum_{k>=0} (zeta vt)^k
// which leaves three families of coefficients X_k = sum_j W_j vt_j^k per block,
//     X1: W = F (k = 1..P+1),   X2: W = conj F (k = 1..P),   X3: W = F vt / conj(vt) (k = 0..P),
// C0 = sum_j (F/2 - F log|c - z_j|), and with S2 = sum_{k>=0} X1_{k+1} zeta^k
//     U = C0 + sum_{k>=1} X1_k zeta^k / (2k) + sum_{k>=0} [conj(X3_k)/2 + conj(X2_k)/(2k)] conj(zeta)^k - zeta conj(S2) / 2
//     sum_j (f'.d)/d^2 = -Re S2 / r.
// Three coefficient launches (a family fills the register file), then the patch kernel: expansion
// values first, then the near batches pair by pair through the table.
//
// The stresslet (round 4).  With N = n_x + i n_y, G = g'_x + i g'_y:  (d.n)(d.g') = [delta^2 conj(N G) + conj(delta)^2
// N G + |delta|^2 2 (n.g')] / 4, so with A = N G and the real C = 2 n.g'
//     U = sum_j (d.n)(d.g') delta / |delta|^4 = (1/4) sum_j [ A / delta + C / conj(delta) + conj(A) delta / conj(delta)^2 ],
//     p / 2 = sum_j [ -(n.g') / (2 |delta|^2) + (d.n)(d.g') / |delta|^4 ] = (1/2) Re sum_j A / delta^2,
// and with 1/delta^2 = (vt^2 / r^2) sum_k (k + 1) (zeta vt)^k three more families, stored by slot k = power - 1,
//     Y1_k = sum_j A vt^(k+1) (k = 0..P+1),   Y2_k = sum_j C vt^(k+1),   Y4_k = sum_j (A vt / conj(vt)) vt^(k+1)   (k = 0..P),
// give, T2 = sum_k (k + 1) Y1_(k+1) zeta^k,
//     U = -(1/4r) sum_k Y1_k zeta^k - (1/4r) sum_k [conj(Y2_k) + (k + 1) conj(Y4_k)] conj(zeta)^k + (1/4r) zeta conj(T2),
//     p / 2 = Re T2 / (2 r^2)
// — the stokeslet's three chains with other coefficients: a1_k += -Y1_k / 4r, a2_k += -(k + 1) Y1_(k+1) / 2r,
// b_k += -[conj(Y2_k) + (k + 1) conj(Y4_k)] / 4r.  The targets' side is the stokeslet's, unchanged; both layers in one
// apply share it.
constexpr int SFAR_P = 26;
constexpr double SFAR_RHO = 0.25;
constexpr int SFAR_NCOEF = 2 * (SFAR_P + 2);
constexpr int SFAR_HDR = 4;

// Two levels as in layer_laplace.hip: PPL = 16 first, `nslice` waves (slices of the sources) per parent
// block of sixteen consecutive blocks, `bits` = the batches the parent takes; PPL = 1 then per block with
// `skip` = those bits, `bits` = its near batches.  WHICH = 1, 2, 3: the stokeslet's families X1, X2, X3;
// 4, 5, 6: the stresslet's Y1, Y2, Y4 (slot k = power k + 1).  `lead`: this launch writes the block headers and
// the batch bits (the first family of an apply).
template <int WHICH, int PPL>
__global__ __launch_bounds__(256) void stokes_far_coeff_kernel(const double* __restrict__ rec, int ns_pad,
                                                              const double* __restrict__ pxy, int64_t np,
                                                              const ApplyParams* __restrict__ prm,
                                                              double* __restrict__ head, double* __restrict__ coef,
                                                              unsigned* __restrict__ near, int nch,
                                                              const unsigned* __restrict__ skip, int nslice,
                                                              int lead) {
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
    const double thr = r2 * (1.0 / (SFAR_RHO * SFAR_RHO)) * (1.0 + 0x1p-40);
    constexpr bool DL = WHICH >= 4;
    constexpr int K0 = (WHICH == 3 || DL) ? 0 : 1;
    constexpr int K1 = (WHICH == 1 || WHICH == 4) ? SFAR_P + 1 : SFAR_P;
    double sre[K1 + 1], sim[K1 + 1];
#pragma unroll
    for (int k = 0; k <= K1; ++k) sre[k] = sim[k] = 0.0;
    // two sources per lane and trip: two independent power chains in flight (the chain of complex
    // products is the critical path at two waves per SIMD)
    for (int j0 = jlo; j0 < jhi; j0 += 128) {
        if (PPL == 1 && skip) {
            // (a trip whose sixteen batches the parent block took whole is nobody's here: see layer_laplace.hip)
            const unsigned t0 = skip[(g >> 4) * nch + (j0 >> 6)];
            const unsigned t1 = j0 + 64 < ns_pad ? skip[(g >> 4) * nch + (j0 >> 6) + 1] : 0xFFu;
            if ((t0 & 0xFFu) == 0xFFu && (t1 & 0xFFu) == 0xFFu) {
                if (lead && lane == 0) {
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
            // a batch of eight sources goes one way as a whole (batches the parent took are nobody's here)
            const unsigned taken = (PPL == 1 && skip && jb < ns_pad) ? skip[(g >> 4) * nch + (jb >> 6)] : 0u;
            const bool mine = valid && !((taken >> (lane >> 3)) & 1u);
            const unsigned long long m = __ballot(mine && !(d2 >= thr && !prm->pad));
            const unsigned long long mv = __ballot(valid);
            const bool far = mine && ((m >> (lane & ~7)) & 0xFFull) == 0;
            anyfar = anyfar || far;
            if (lead && lane == 0 && jb < ns_pad) {
                const unsigned nearbits = far_batch_bits(m);
                near[g * nch + (jb >> 6)] = PPL == 1 ? nearbits : (far_batch_bits(mv) & ~nearbits);
            }
            const double inv = far ? r / d2 : 0.0;
            vre[h] = dx * inv;
            vim[h] = -dy * inv;                            // vt = r / (z_j - c)
            double fx, fy;
            if (!DL) {
                fx = far ? rec[ipde_rec_index(jj, 2)] : 0.0;
                fy = far ? rec[ipde_rec_index(jj, 3)] : 0.0;
            } else if (WHICH == 5) {
                fx = far ? 2.0 * rec[ipde_rec_index(jj, 8)] : 0.0;        // C = 2 n.g'
                fy = 0.0;
            } else {
                const double gx = far ? rec[ipde_rec_index(jj, 4)] : 0.0, gy = far ? rec[ipde_rec_index(jj, 5)] : 0.0;
                const double nx = rec[ipde_rec_index(jj, 6)], ny = rec[ipde_rec_index(jj, 7)];
                fx = nx * gx - ny * gy;                                     // A = N G
                fy = nx * gy + ny * gx;
            }
            if (WHICH == 4 || WHICH == 5) {
                wre[h] = fx;
                wim[h] = fy;
            } else if (WHICH == 1) {
                wre[h] = fx;
                wim[h] = fy;
                const double hl = fma(-0.5, log(far ? d2 : 1.0), 0.5);     // 1/2 - log|c - z_j|
                sre[0] = fma(fx, hl, sre[0]);
                sim[0] = fma(fy, hl, sim[0]);
            } else if (WHICH == 2) {
                wre[h] = fx;
                wim[h] = -fy;
            } else {
                // F vt / conj(vt) = F vt^2 / |vt|^2 = F (dx - i dy)^2 / d2
                const double id = far ? 1.0 / d2 : 0.0;
                const double ure = (dx * dx - dy * dy) * id, uim = -2.0 * dx * dy * id;
                wre[h] = fx * ure - fy * uim;
                wim[h] = fx * uim + fy * ure;
            }
        }
        if (__ballot(anyfar) == 0) continue;               // nothing of this trip enters the expansion (wave-uniform)
        double pre[2], pim[2];                             // vt^k
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pre[h] = (K0 == 0 && !DL) ? 1.0 : vre[h];
            pim[h] = (K0 == 0 && !DL) ? 0.0 : vim[h];
        }
#pragma unroll
        for (int k = K0; k <= K1; ++k) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                sre[k] += wre[h] * pre[h] - wim[h] * pim[h];
                sim[k] += wre[h] * pim[h] + wim[h] * pre[h];
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
        if (lead && slice == 0) {
            head[g * SFAR_HDR + 0] = cx;
            head[g * SFAR_HDR + 1] = cy;
            head[g * SFAR_HDR + 2] = 1.0 / r;
            head[g * SFAR_HDR + 3] = 0.0;
        }
        double* c = coef + gw * SFAR_NCOEF;
#pragma unroll
        for (int k = 0; k <= K1; ++k) {
            c[2 * k] = sre[k];
            c[2 * k + 1] = sim[k];
        }
    }
}

// Entry k of the combined coefficients the Horner loops read (the three families summed over `nslice`
// rows each):  [3k] a1_k = X1_k / (2k) (k = 0: C0),  [3k+1] a2_k = X1_(k+1),  [3k+2] b_k = conj(X3_k)/2 + conj(X2_k)/(2k)
// ALL: the layers in this apply (MODE_SLP: X1..X3; MODE_DLP: D1, D2, D4 = the stresslet's Y1, Y2, Y4, see the head of
// the section; rinv = 1 / r of the block).
template <int ALL>
__device__ __forceinline__ void sfar_combine(double2* __restrict__ W, int k, const double* __restrict__ X1,
                                             const double* __restrict__ X2, const double* __restrict__ X3,
                                             const double* __restrict__ D1, const double* __restrict__ D2,
                                             const double* __restrict__ D4, double rinv, int nslice) {
    double a1r = 0.0, a1i = 0.0, a2r = 0.0, a2i = 0.0, br = 0.0, bi = 0.0;
    const double hk = k >= 1 ? 0.5 / (double)k : 1.0;
    const double q4 = -0.25 * rinv, q2 = -0.5 * rinv * (double)(k + 1), q4k = q4 * (double)(k + 1);
    for (int sl = 0; sl < nslice; ++sl) {
        if (ALL & MODE_SLP) {
            const double* Y1 = X1 + (size_t)sl * SFAR_NCOEF;
            const double* Y2 = X2 + (size_t)sl * SFAR_NCOEF;
            const double* Y3 = X3 + (size_t)sl * SFAR_NCOEF;
            a1r += hk * Y1[2 * k];
            a1i += hk * Y1[2 * k + 1];
            a2r += Y1[2 * (k + 1)];
            a2i += Y1[2 * (k + 1) + 1];
            br += 0.5 * Y3[2 * k] + (k >= 1 ? hk * Y2[2 * k] : 0.0);
            bi -= 0.5 * Y3[2 * k + 1] + (k >= 1 ? hk * Y2[2 * k + 1] : 0.0);
        }
        if (ALL & MODE_DLP) {
            const double* Z1 = D1 + (size_t)sl * SFAR_NCOEF;
            const double* Z2 = D2 + (size_t)sl * SFAR_NCOEF;
            const double* Z4 = D4 + (size_t)sl * SFAR_NCOEF;
            a1r += q4 * Z1[2 * k];
            a1i += q4 * Z1[2 * k + 1];
            a2r += q2 * Z1[2 * (k + 1)];
            a2i += q2 * Z1[2 * (k + 1) + 1];
            br += q4 * Z2[2 * k] + q4k * Z4[2 * k];
            bi -= q4 * Z2[2 * k + 1] + q4k * Z4[2 * k + 1];
        }
    }
    W[3 * k] = double2{a1r, a1i};
    W[3 * k + 1] = double2{a2r, a2i};
    W[3 * k + 2] = double2{br, bi};
}

// MODE: the layer of the near pairs this launch sums (MODE_SLP or MODE_DLP); ALL: the layers of the apply (what the
// coefficients hold, what a table miss recomputes); FAR: this launch starts from the block's expansion and STORES;
// !FAR: it ADDS its near pairs to what is stored.  Both layers: <SLP, 3, true> then <DLP, 3, false> (one launch with
// both near bodies would not fit the register file).
template <int NT, int MODE, int ALL, bool FAR>
__global__ __launch_bounds__(NT) void stokes_patch_far_kernel(
    const double* __restrict__ rec, int ns_pad, const double* __restrict__ pxy, int64_t np,
    const int* __restrict__ pout, double* __restrict__ ou, double* __restrict__ ov, double* __restrict__ op,
    const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab, unsigned key_lo, unsigned nkeys,
    const double* __restrict__ head, const double* __restrict__ c1, const double* __restrict__ c2,
    const double* __restrict__ c3, const double* __restrict__ c4, const double* __restrict__ c5,
    const double* __restrict__ c6, const unsigned* __restrict__ near, int nch,
    const unsigned* __restrict__ taken) {
    extern __shared__ double2 ltab[];
    for (unsigned i = threadIdx.x; i < nkeys; i += NT) ltab[i] = gtab[i];
    __syncthreads();
    TabAddr ta;
    const double s1 = ldexp(1.0, prm->sh);
    // wave w of workgroup b takes block w * gridDim.x + b (see laplace_patch_far_kernel)
    const int wv = threadIdx.x >> 6;
    const int64_t g = __builtin_amdgcn_readfirstlane((int)(wv * gridDim.x + blockIdx.x));
    if (g * 64 >= np) return;                          // (whole waves, after the only barrier)
    const int64_t lane = g * 64 + (threadIdx.x & 63);
    const int64_t t = min(lane, np - 1);
    // the block's combined coefficients, parked in LDS behind the table (lane k stages entry k)
    constexpr int WCL = 3 * (SFAR_P + 1);
    double2* wc = ltab + nkeys + wv * WCL;
    if (FAR && (threadIdx.x & 63) <= SFAR_P)
        sfar_combine<ALL>(wc, threadIdx.x & 63, c1 + g * SFAR_NCOEF, c2 + g * SFAR_NCOEF, c3 + g * SFAR_NCOEF,
                          c4 + g * SFAR_NCOEF, c5 + g * SFAR_NCOEF, c6 + g * SFAR_NCOEF, head[g * SFAR_HDR + 2], 1);
    __builtin_amdgcn_wave_barrier();
    double xs[4], ys[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        xs[a] = pxy[(int64_t)a * np + t] * s1;
        ys[a] = pxy[(int64_t)(4 + a) * np + t] * s1;
    }
    StokesAcc acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = StokesAcc{0.0, 0.0, 0.0, 0.0, 0.0};
    // far sources first: the block's expansion values are the accumulators' starting values (the parent's
    // expansion is added by stokes_far_parent_kernel afterwards: this kernel is at its register limit)
    if (FAR) {
        const double* h = head + g * SFAR_HDR;
        const double cx = h[0], cy = h[1], rinv = h[2];
        const double2* W = wc;
        const double2 c0 = W[0];
        double zy[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) zy[b] = (ys[b] - cy) * rinv;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double zx = (xs[a] - cx) * rinv;
            double s1r[4], s1i[4], s2r[4], s2i[4], s3r[4], s3i[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) s1r[b] = s1i[b] = s2r[b] = s2i[b] = s3r[b] = s3i[b] = 0.0;
#pragma unroll 1
            for (int k = SFAR_P; k >= 0; --k) {
                const double2 a1 = k >= 1 ? W[3 * k] : double2{0.0, 0.0};
                const double2 a2 = W[3 * k + 1];
                const double2 b3 = W[3 * k + 2];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    double nr = s1r[b] * zx - s1i[b] * zy[b] + a1.x;
                    s1i[b] = fma(s1r[b], zy[b], s1i[b] * zx) + a1.y;
                    s1r[b] = nr;
                    nr = s2r[b] * zx - s2i[b] * zy[b] + a2.x;
                    s2i[b] = fma(s2r[b], zy[b], s2i[b] * zx) + a2.y;
                    s2r[b] = nr;
                    // conj(zeta) = zx - i zy
                    nr = s3r[b] * zx + s3i[b] * zy[b] + b3.x;
                    s3i[b] = fma(-s3r[b], zy[b], s3i[b] * zx) + b3.y;
                    s3r[b] = nr;
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                // zeta conj(S2) = (zx + i zy)(s2r - i s2i)
                const double qr = zx * s2r[b] + zy[b] * s2i[b], qi = zy[b] * s2r[b] - zx * s2i[b];
                acc[4 * a + b] = StokesAcc{0.0, 0.0, c0.x + s1r[b] + s3r[b] - 0.5 * qr,
                                           c0.y + s1i[b] + s3i[b] - 0.5 * qi, -rinv * s2r[b]};
            }
        }
    }
    // near sources: batch by batch through the table
    const unsigned* nm = near + g * nch;
    for (int c = 0; c < nch; ++c) {
        unsigned m = nm[c];
        while (m) {
            const int bt = __builtin_ctz(m);
            m &= m - 1;
            const double* row = rec + ((size_t)(8 * c + bt) * IPDE_SRC_NCH) * IPDE_SRC_PAD;
#pragma unroll
            for (int u = 0; u < IPDE_SRC_PAD; ++u) {
                const double sx = row[u], sy = row[IPDE_SRC_PAD + u];
                if (MODE == MODE_SLP) {
                    const double fx = row[2 * IPDE_SRC_PAD + u], fy = row[3 * IPDE_SRC_PAD + u];
                    double dy[4], dy2[4], fydy[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        dy[b] = ys[b] - sy;
                        dy2[b] = dy[b] * dy[b];
                        fydy[b] = fy * dy[b];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const double dx = xs[a] - sx;
                        const double dx2 = dx * dx, fxdx = fx * dx;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const double d2 = dx2 + dy2[b];
                            const double2 e = ta.lookup(ltab, d2);
                            const double yy = tab_y(d2, e.x);
                            const double ri = rcp_from_y_fast(e.x, yy);
                            const double L = log_from_y(yy, e.y);
                            const double tt = (fxdx + fydy[b]) * ri;
                            StokesAcc& A = acc[4 * a + b];
                            A.uL = fma(fx, L, A.uL);
                            A.vL = fma(fy, L, A.vL);
                            A.u = fma(tt, dx, A.u);
                            A.v = fma(tt, dy[b], A.v);
                            A.p += tt;
                        }
                    }
                } else {
                    // stresslet: w = (d.n)(d.g') / d2^2;  u += w dx, v += w dy, p/2 += w - (n.g') / (2 d2)
                    const double gx = row[4 * IPDE_SRC_PAD + u], gy = row[5 * IPDE_SRC_PAD + u];
                    const double nx = row[6 * IPDE_SRC_PAD + u], ny = row[7 * IPDE_SRC_PAD + u];
                    const double hng = -0.5 * row[8 * IPDE_SRC_PAD + u];
                    double dy[4], dy2[4], nydy[4], gydy[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        dy[b] = ys[b] - sy;
                        dy2[b] = dy[b] * dy[b];
                        nydy[b] = ny * dy[b];
                        gydy[b] = gy * dy[b];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const double dx = xs[a] - sx;
                        const double dx2 = dx * dx, nxdx = nx * dx, gxdx = gx * dx;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const double d2 = dx2 + dy2[b];
                            const double2 e = ta.lookup(ltab, d2);
                            const double ri = rcp_from_y_fast(e.x, tab_y(d2, e.x));
                            const double w = (nxdx + nydy[b]) * (gxdx + gydy[b]) * (ri * ri);
                            StokesAcc& A = acc[4 * a + b];
                            A.u = fma(w, dx, A.u);
                            A.v = fma(w, dy[b], A.v);
                            A.p += fma(hng, ri, w);
                        }
                    }
                }
            }
        }
    }
    if (!ta.all_inside(key_lo) || prm->pad) {
        // a near pair of this patch left the table (or the scaling failed: then nothing is in an
        // expansion): all sources of its targets again with the generic math, every layer of the apply — all
        // but the batches the parent block took (stokes_far_parent_kernel adds those to whatever is stored
        // here).  The launch that only adds near pairs sees the same misses (same pairs, same table) and then
        // adds nothing: the storing launch has recomputed everything.
        const unsigned* tk = taken + (g >> 4) * nch;
        const int nbatch = ns_pad / IPDE_SRC_PAD;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double xa[4] = {xs[a], xs[a], xs[a], xs[a]};
            StokesAcc gs[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) gs[b] = StokesAcc{0, 0, 0, 0, 0};
            if (FAR) {
                for (int c = 0; c < nch; ++c) {
                    unsigned m = ~tk[c] & 0xFFu;
                    while (m) {
                        const int bt = __builtin_ctz(m);
                        m &= m - 1;
                        if (8 * c + bt < nbatch)
                            stokes_generic_loop<ALL, false, 4>(rec, 8 * (8 * c + bt), 8 * (8 * c + bt) + 8, xa, ys, gs);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[4 * a + b] = gs[b];
        }
    }
    if (lane >= np) return;
    const double cu = (ALL & MODE_SLP) ? prm->corr : 0.0, cv = (ALL & MODE_SLP) ? prm->corr2 : 0.0;
    const double ps = 2.0 * s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = pout[(int64_t)r * np + t];
        if (i >= 0) {
            if (FAR) {
                ou[i] = fma(-0.5, acc[r].uL + cu, acc[r].u);
                ov[i] = fma(-0.5, acc[r].vL + cv, acc[r].v);
                if (op) op[i] = ps * acc[r].p;
            } else {
                ou[i] += fma(-0.5, acc[r].uL, acc[r].u);
                ov[i] += fma(-0.5, acc[r].vL, acc[r].v);
                if (op) op[i] += ps * acc[r].p;
            }
        }
    }
}

// The parent blocks' expansions added to the stored values (one lane per patch, all of the parent's
// sixteen blocks read the same 3 x 27 combined coefficients: staged in LDS by the workgroup = one parent).
template <int ALL>
__global__ __launch_bounds__(1024) void stokes_far_parent_kernel(const double* __restrict__ pxy, int64_t np,
                                                                 const int* __restrict__ pout, double* __restrict__ ou,
                                                                 double* __restrict__ ov, double* __restrict__ op,
                                                                 const ApplyParams* __restrict__ prm,
                                                                 const double* __restrict__ head2,
                                                                 const double* __restrict__ p1,
                                                                 const double* __restrict__ p2,
                                                                 const double* __restrict__ p3,
                                                                 const double* __restrict__ p4,
                                                                 const double* __restrict__ p5,
                                                                 const double* __restrict__ p6, int nslice) {
    constexpr int WCL = 3 * (SFAR_P + 1);
    __shared__ double2 W[WCL];
    const int64_t par = blockIdx.x;                    // patches [1024 par, 1024 par + 1024)
    if (threadIdx.x <= SFAR_P) {
        const size_t o = (size_t)par * nslice * SFAR_NCOEF;
        sfar_combine<ALL>(W, threadIdx.x, p1 + o, p2 + o, p3 + o, p4 + o, p5 + o, p6 + o, head2[par * SFAR_HDR + 2],
                          nslice);
    }
    __syncthreads();
    const int64_t t = par * 1024 + threadIdx.x;
    if (t >= np || prm->pad) return;
    const double s1 = ldexp(1.0, prm->sh);
    const double cx = head2[par * SFAR_HDR], cy = head2[par * SFAR_HDR + 1], rinv = head2[par * SFAR_HDR + 2];
    const double2 c0 = W[0];
    const double ps = 2.0 * s1;
    double zy[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) zy[b] = (pxy[(int64_t)(4 + b) * np + t] * s1 - cy) * rinv;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double zx = (pxy[(int64_t)a * np + t] * s1 - cx) * rinv;
        double s1r[4], s1i[4], s2r[4], s2i[4], s3r[4], s3i[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) s1r[b] = s1i[b] = s2r[b] = s2i[b] = s3r[b] = s3i[b] = 0.0;
#pragma unroll 2
        for (int k = SFAR_P; k >= 0; --k) {
            const double2 a1 = k >= 1 ? W[3 * k] : double2{0.0, 0.0};
            const double2 a2 = W[3 * k + 1];
            const double2 b3 = W[3 * k + 2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double nr = s1r[b] * zx - s1i[b] * zy[b] + a1.x;
                s1i[b] = fma(s1r[b], zy[b], s1i[b] * zx) + a1.y;
                s1r[b] = nr;
                nr = s2r[b] * zx - s2i[b] * zy[b] + a2.x;
                s2i[b] = fma(s2r[b], zy[b], s2i[b] * zx) + a2.y;
                s2r[b] = nr;
                nr = s3r[b] * zx + s3i[b] * zy[b] + b3.x;
                s3i[b] = fma(-s3r[b], zy[b], s3i[b] * zx) + b3.y;
                s3r[b] = nr;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = pout[(int64_t)(4 * a + b) * np + t];
            if (i < 0) continue;
            const double qr = zx * s2r[b] + zy[b] * s2i[b], qi = zy[b] * s2r[b] - zx * s2i[b];
            ou[i] += c0.x + s1r[b] + s3r[b] - 0.5 * qr;
            ov[i] += c0.y + s1i[b] + s3i[b] - 0.5 * qi;
            if (op) op[i] += ps * (-rinv * s2r[b]);
        }
    }
}

// ---------------------------------------------------------------------------
// The far-field form for the radial grids of the annuli (ipde_stokes_apply_columns_far; the scheme and the
// stand-in patch list are described at modhelm_cols_far_kernel in layer_modhelm.hip): targets (M, N)
// row-major, column j = one radial line; blocks of 64 columns, one level, eight source slices per block; a
// wave takes four rows of a block.
__global__ __launch_bounds__(256) void stokes_columns_as_patches_kernel(const double* __restrict__ tx,
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

// ALL: the layers of the apply (both near bodies in one launch here: four targets per lane leave the room)
template <int NT, int ALL>
__global__ __launch_bounds__(NT) void stokes_cols_far_kernel(
    const double* __restrict__ rec, int ns_pad, const double* __restrict__ tx, const double* __restrict__ ty, int M,
    int64_t N, double* __restrict__ ou, double* __restrict__ ov, double* __restrict__ op,
    const ApplyParams* __restrict__ prm, const double2* __restrict__ gtab, unsigned key_lo, unsigned nkeys,
    const double* __restrict__ head, const double* __restrict__ c1, const double* __restrict__ c2,
    const double* __restrict__ c3, const double* __restrict__ c4, const double* __restrict__ c5,
    const double* __restrict__ c6, int nslice, const unsigned* __restrict__ near, int nch) {
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
    constexpr int WCL = 3 * (SFAR_P + 1);
    double2* W = ltab + nkeys + wv * WCL;
    if (ln <= SFAR_P) {
        const size_t o = (size_t)g * nslice * SFAR_NCOEF;
        sfar_combine<ALL>(W, ln, c1 + o, c2 + o, c3 + o, c4 + o, c5 + o, c6 + o, head[g * SFAR_HDR + 2], nslice);
    }
    __builtin_amdgcn_wave_barrier();
    const double* h = head + g * SFAR_HDR;
    const double cx = h[0], cy = h[1], rinv = h[2];
    const int64_t j = g * 64 + ln, jj = min(j, N - 1);
    double x[4], y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = (int64_t)min(r0 + i, M - 1) * N + jj;
        x[i] = tx[t] * s1;
        y[i] = ty[t] * s1;
    }
    // far sources first
    double fu[4], fv[4], fq[4];
    {
        const double2 c0 = W[0];
        double zx[4], zy[4], s1r[4], s1i[4], s2r[4], s2i[4], s3r[4], s3i[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            zx[i] = (x[i] - cx) * rinv;
            zy[i] = (y[i] - cy) * rinv;
            s1r[i] = s1i[i] = s2r[i] = s2i[i] = s3r[i] = s3i[i] = 0.0;
        }
#pragma unroll 2
        for (int k = SFAR_P; k >= 0; --k) {
            const double2 a1 = k >= 1 ? W[3 * k] : double2{0.0, 0.0};
            const double2 a2 = W[3 * k + 1];
            const double2 b3 = W[3 * k + 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double nr = s1r[i] * zx[i] - s1i[i] * zy[i] + a1.x;
                s1i[i] = fma(s1r[i], zy[i], s1i[i] * zx[i]) + a1.y;
                s1r[i] = nr;
                nr = s2r[i] * zx[i] - s2i[i] * zy[i] + a2.x;
                s2i[i] = fma(s2r[i], zy[i], s2i[i] * zx[i]) + a2.y;
                s2r[i] = nr;
                nr = s3r[i] * zx[i] + s3i[i] * zy[i] + b3.x;
                s3i[i] = fma(-s3r[i], zy[i], s3i[i] * zx[i]) + b3.y;
                s3r[i] = nr;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double qr = zx[i] * s2r[i] + zy[i] * s2i[i], qi = zy[i] * s2r[i] - zx[i] * s2i[i];
            const bool on = !prm->pad;
            fu[i] = on ? c0.x + s1r[i] + s3r[i] - 0.5 * qr : 0.0;
            fv[i] = on ? c0.y + s1i[i] + s3i[i] - 0.5 * qi : 0.0;
            fq[i] = on ? -rinv * s2r[i] : 0.0;
        }
    }
    StokesAcc acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = StokesAcc{0, 0, 0, 0, 0};
    TabAddr ta;
    const unsigned* nm = near + g * nch;
    for (int c = 0; c < nch; ++c) {
        unsigned m = nm[c];
        while (m) {
            const int bt = __builtin_ctz(m);
            m &= m - 1;
            const double* row = rec + ((size_t)(8 * c + bt) * IPDE_SRC_NCH) * IPDE_SRC_PAD;
#pragma unroll
            for (int u = 0; u < IPDE_SRC_PAD; ++u) {
                StokesSrc sr{};
                if (ALL & MODE_SLP) {
                    sr.fx = row[2 * IPDE_SRC_PAD + u];
                    sr.fy = row[3 * IPDE_SRC_PAD + u];
                }
                if (ALL & MODE_DLP) {
                    sr.gx = row[4 * IPDE_SRC_PAD + u];
                    sr.gy = row[5 * IPDE_SRC_PAD + u];
                    sr.nx = row[6 * IPDE_SRC_PAD + u];
                    sr.ny = row[7 * IPDE_SRC_PAD + u];
                    sr.ng = row[8 * IPDE_SRC_PAD + u];
                }
                const double sxu = row[u], syu = row[IPDE_SRC_PAD + u];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double dx = x[i] - sxu, dy = y[i] - syu;
                    const double d2 = fma(dy, dy, dx * dx);
                    const double2 e = ta.lookup(ltab, d2);
                    const double yy = tab_y(d2, e.x);
                    stokes_pair<ALL>(dx, dy, (ALL & MODE_SLP) ? log_from_y(yy, e.y) : 0.0, rcp_from_y_fast(e.x, yy), sr,
                                     acc[i]);
                }
            }
        }
    }
    if (!ta.all_inside(key_lo) || prm->pad) {
        StokesAcc gs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gs[i] = StokesAcc{0, 0, 0, 0, 0};
        if (prm->pad) {
            stokes_generic_loop<ALL, false, 4>(rec, 0, ns_pad, x, y, gs);
        } else {
            for (int c = 0; c < nch; ++c) {
                unsigned m = nm[c];
                while (m) {
                    const int bt = __builtin_ctz(m);
                    m &= m - 1;
                    stokes_generic_loop<ALL, false, 4>(rec, 8 * (8 * c + bt), 8 * (8 * c + bt) + 8, x, y, gs);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = gs[i];
    }
    if (j < N) {
        const double cu = (ALL & MODE_SLP) ? prm->corr : 0.0, cv = (ALL & MODE_SLP) ? prm->corr2 : 0.0;
        const double ps = 2.0 * s1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + i < M) {
                const int64_t t = (int64_t)(r0 + i) * N + j;
                ou[t] = fma(-0.5, acc[i].uL + cu, acc[i].u + fu[i]);
                ov[t] = fma(-0.5, acc[i].vL + cv, acc[i].v + fv[i]);
                if (op) op[t] = ps * (acc[i].p + fq[i]);
            }
    }
}

// ALL: MODE_SLP, MODE_DLP or both (the densities present in the records)
template <int ALL>
int launch_stokes_patches_far(ipde_ctx* ctx, const double* rec, int64_t ns, const double* pxy, int64_t np,
                              const int* pout, double* ou, double* ov, double* op, const ApplyParams* prm) {
    constexpr int NT = 512;
    constexpr int NSL = 8;                             // waves (slices of the sources) per parent block
    const LogTable& lt = ctx->logtab;
    const int ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    const int64_t ng = ceil_div64(np, 64);
    const int64_t ng2 = ceil_div64(ng, 16);
    const int nch = (int)ceil_div64(ns_pad, 64);
    // block families 1..6 (stokeslet X1, X2, X3; stresslet Y1, Y2, Y4), the same per parent and source slice
    const size_t nd = (size_t)ng * (SFAR_HDR + 6 * SFAR_NCOEF) + (size_t)ng2 * (SFAR_HDR + 6 * NSL * SFAR_NCOEF);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial,
                                 nd * sizeof(double) + (size_t)(ng + ng2) * nch * sizeof(unsigned)));
    double* head = (double*)ctx->partial.p;
    double* c[6];
    c[0] = head + (size_t)ng * SFAR_HDR;
    for (int f = 1; f < 6; ++f) c[f] = c[f - 1] + (size_t)ng * SFAR_NCOEF;
    double* head2 = c[5] + (size_t)ng * SFAR_NCOEF;
    double* q[6];
    q[0] = head2 + (size_t)ng2 * SFAR_HDR;
    for (int f = 1; f < 6; ++f) q[f] = q[f - 1] + (size_t)ng2 * NSL * SFAR_NCOEF;
    unsigned* near = (unsigned*)(q[5] + (size_t)ng2 * NSL * SFAR_NCOEF);
    unsigned* taken = near + (size_t)ng * nch;
    const unsigned gb = (unsigned)ceil_div64(ng, 4), gb2 = (unsigned)ceil_div64(ng2 * NSL, 4);
    const unsigned* none = nullptr;
    ipde_time_begin(ctx);
    // parents first (their bits steer the blocks' pass); the first family of the apply writes headers and bits
    constexpr int L1 = (ALL & MODE_SLP) ? 1 : 0;
#define SFAR_PARENT(WHICH, F, LEAD)                                                                                    \
    hipLaunchKernelGGL((stokes_far_coeff_kernel<WHICH, 16>), dim3(gb2), dim3(256), 0, ctx->stream, rec, ns_pad, pxy, np, \
                       prm, head2, q[F], taken, nch, none, NSL, LEAD)
#define SFAR_BLOCK(WHICH, F, LEAD)                                                                                     \
    hipLaunchKernelGGL((stokes_far_coeff_kernel<WHICH, 1>), dim3(gb), dim3(256), 0, ctx->stream, rec, ns_pad, pxy, np,   \
                       prm, head, c[F], near, nch, (const unsigned*)taken, 1, LEAD)
    if (ALL & MODE_SLP) {
        SFAR_PARENT(1, 0, 1);
        SFAR_PARENT(2, 1, 0);
        SFAR_PARENT(3, 2, 0);
    }
    if (ALL & MODE_DLP) {
        SFAR_PARENT(4, 3, 1 - L1);
        SFAR_PARENT(5, 4, 0);
        SFAR_PARENT(6, 5, 0);
    }
    if (ALL & MODE_SLP) {
        SFAR_BLOCK(1, 0, 1);
        SFAR_BLOCK(2, 1, 0);
        SFAR_BLOCK(3, 2, 0);
    }
    if (ALL & MODE_DLP) {
        SFAR_BLOCK(4, 3, 1 - L1);
        SFAR_BLOCK(5, 4, 0);
        SFAR_BLOCK(6, 5, 0);
    }
#undef SFAR_PARENT
#undef SFAR_BLOCK
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    // the table, then 27 x 3 combined coefficients for each of the workgroup's waves
    const size_t lds = ((size_t)lt.nkeys + (size_t)(NT / 64) * 3 * (SFAR_P + 1)) * sizeof(double2);
    auto patches = [&](auto kern) -> int {
        IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div64(64 * ng, NT)), dim3(NT), lds, ctx->stream, rec, ns_pad, pxy, np,
                           pout, ou, ov, op, prm, (const double2*)lt.d_tab, (unsigned)lt.key_lo, (unsigned)lt.nkeys,
                           (const double*)head, (const double*)c[0], (const double*)c[1], (const double*)c[2],
                           (const double*)c[3], (const double*)c[4], (const double*)c[5], (const unsigned*)near, nch,
                           (const unsigned*)taken);
        return IPDE_OK;
    };
    if (ALL & MODE_SLP) {
        IPDE_TRY(patches(stokes_patch_far_kernel<NT, MODE_SLP, ALL, true>));
        if (ALL & MODE_DLP) IPDE_TRY(patches(stokes_patch_far_kernel<NT, MODE_DLP, ALL, false>));
    } else {
        IPDE_TRY(patches(stokes_patch_far_kernel<NT, MODE_DLP, ALL, true>));
    }
    hipLaunchKernelGGL(stokes_far_parent_kernel<ALL>, dim3((unsigned)ng2), dim3(1024), 0, ctx->stream, pxy, np, pout, ou,
                       ov, op, prm, (const double*)head2, (const double*)q[0], (const double*)q[1], (const double*)q[2],
                       (const double*)q[3], (const double*)q[4], (const double*)q[5], NSL);
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
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

// the records of an apply (see the head of the file)
static void stokes_pack_args(ipde_ctx* ctx, PackArgs& pa, const double* sx, const double* sy, const double* wfx,
                             const double* wfy, const double* nx, const double* ny, const double* wdx,
                             const double* wdy) {
    const bool slp = wfx && wfy, dlp = wdx && wdy;
    pa.sx = sx;
    pa.sy = sy;
    pa.ch[0] = slp ? wfx : nullptr;
    pa.mul[0] = 0.25 / M_PI;
    pa.ch[1] = slp ? wfy : nullptr;
    pa.mul[1] = 0.25 / M_PI;
    pa.ch[2] = dlp ? wdx : nullptr;
    pa.mul[2] = 1.0 / M_PI;
    pa.pw[2] = 1;
    pa.ch[3] = dlp ? wdy : nullptr;
    pa.mul[3] = 1.0 / M_PI;
    pa.pw[3] = 1;
    pa.ch[4] = dlp ? nx : nullptr;
    pa.mul[4] = 1.0;
    pa.ch[5] = dlp ? ny : nullptr;
    pa.mul[5] = 1.0;
    pa.stokes_ng = dlp ? 1 : 0;
    pa.corr_ch = slp ? 0 : -1;
    pa.corr2_ch = slp ? 1 : -1;
    pa.use_scale = 1;
    pa.exp_hi = ctx->logtab.exp_hi;
}

// Stokeslet and / or stresslet sums (with pressure) onto a patch list whose 64-patch groups are 8 x 8 blocks of tiles
// (ipde_target_plan_build_blocks, pad_blocks = 1): far sources block by block in local expansions.
extern "C" int ipde_stokes_apply_patches_far(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                             const double* wfx, const double* wfy, const double* nx,
                                             const double* ny, const double* wdx, const double* wdy, int64_t np,
                                             const double* pxy, const int32_t* pout, double* out_u,
                                             double* out_v, double* out_p) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ns >= 0 && np >= 0 && ns < (1LL << 30) && np < (1LL << 27));
    const bool slp = wfx != nullptr && wfy != nullptr, dlp = wdx != nullptr && wdy != nullptr;
    IPDE_CHECK_ARG(ctx, (wfx == nullptr) == (wfy == nullptr));
    IPDE_CHECK_ARG(ctx, (wdx == nullptr) == (wdy == nullptr));
    IPDE_CHECK_ARG(ctx, slp || dlp);
    IPDE_CHECK_ARG(ctx, !dlp || (nx != nullptr && ny != nullptr));
    if (np == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, pxy && pout && out_u && out_v);
    IPDE_CHECK_ARG(ctx, ns > 0 && sx && sy);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    PackArgs pa{};
    stokes_pack_args(ctx, pa, sx, sy, wfx, wfy, nx, ny, wdx, wdy);
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, pxy, pxy + 4 * np, 4 * np, &rec, &prm));
    if (slp && dlp) return launch_stokes_patches_far<3>(ctx, rec, ns, pxy, np, pout, out_u, out_v, out_p, prm);
    if (slp) return launch_stokes_patches_far<MODE_SLP>(ctx, rec, ns, pxy, np, pout, out_u, out_v, out_p, prm);
    return launch_stokes_patches_far<MODE_DLP>(ctx, rec, ns, pxy, np, pout, out_u, out_v, out_p, prm);
}

template <int ALL>
static int launch_stokes_columns_far(ipde_ctx* ctx, const double* rec, int64_t ns, int M, int64_t N, const double* tx,
                                     const double* ty, double* out_u, double* out_v, double* out_p,
                                     const ApplyParams* prm) {
    constexpr int NT = 256, NSL = 8;
    const LogTable& lt = ctx->logtab;
    const int ns_pad = (int)(ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD);
    const int64_t ng = ceil_div64(N, 64);
    const int nch = (int)ceil_div64(ns_pad, 64);
    const size_t nd = (size_t)8 * N + (size_t)ng * (SFAR_HDR + 6 * NSL * SFAR_NCOEF);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, nd * sizeof(double) + (size_t)ng * nch * sizeof(unsigned)));
    double* pxy = (double*)ctx->partial.p;
    double* head = pxy + (size_t)8 * N;
    double* c[6];
    c[0] = head + (size_t)ng * SFAR_HDR;
    for (int f = 1; f < 6; ++f) c[f] = c[f - 1] + (size_t)ng * NSL * SFAR_NCOEF;
    unsigned* near = (unsigned*)(c[5] + (size_t)ng * NSL * SFAR_NCOEF);
    const unsigned gb = (unsigned)ceil_div64(ng * NSL, 4);
    const unsigned* none = nullptr;
    ipde_time_begin(ctx);
    hipLaunchKernelGGL(stokes_columns_as_patches_kernel, dim3((unsigned)ceil_div64(N, 256)), dim3(256), 0, ctx->stream,
                       tx, ty, M, N, pxy);
    constexpr int L1 = (ALL & MODE_SLP) ? 1 : 0;
#define SFAR_COLS(WHICH, F, LEAD)                                                                                  \
    hipLaunchKernelGGL((stokes_far_coeff_kernel<WHICH, 1>), dim3(gb), dim3(256), 0, ctx->stream, rec, ns_pad,          \
                       (const double*)pxy, N, prm, head, c[F], near, nch, none, NSL, LEAD)
    if (ALL & MODE_SLP) {
        SFAR_COLS(1, 0, 1);
        SFAR_COLS(2, 1, 0);
        SFAR_COLS(3, 2, 0);
    }
    if (ALL & MODE_DLP) {
        SFAR_COLS(4, 3, 1 - L1);
        SFAR_COLS(5, 4, 0);
        SFAR_COLS(6, 5, 0);
    }
#undef SFAR_COLS
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    const size_t lds = ((size_t)lt.nkeys + (size_t)(NT / 64) * 3 * (SFAR_P + 1)) * sizeof(double2);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stokes_cols_far_kernel<NT, ALL>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((stokes_cols_far_kernel<NT, ALL>), dim3((unsigned)ceil_div64(64 * ng * ((M + 3) / 4), NT)), dim3(NT),
                       lds, ctx->stream, rec, ns_pad, tx, ty, M, N, out_u, out_v, out_p, prm, (const double2*)lt.d_tab,
                       (unsigned)lt.key_lo, (unsigned)lt.nkeys, (const double*)head, (const double*)c[0],
                       (const double*)c[1], (const double*)c[2], (const double*)c[3], (const double*)c[4],
                       (const double*)c[5], NSL, (const unsigned*)near, nch);
    ipde_time_end(ctx);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// Stokeslet and / or stresslet sums (with pressure when out_p != NULL) onto an (M, N) radial grid (row-major DEVICE
// arrays; column j = one radial line): the radial sums of the Stokes helpers' correct()
// (reference ipde/solvers/internals/vector.py:140-162) with the far sources of every block of 64 lines in
// local expansions.
extern "C" int ipde_stokes_apply_columns_far(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                             const double* wfx, const double* wfy, const double* nx,
                                             const double* ny, const double* wdx, const double* wdy, int M,
                                             int64_t N, const double* tx, const double* ty, double* out_u,
                                             double* out_v, double* out_p) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, ns >= 0 && N >= 0 && M >= 1 && ns < (1LL << 30) && N < (1LL << 30) && (int64_t)M * N < (1LL << 40));
    const bool slp = wfx != nullptr && wfy != nullptr, dlp = wdx != nullptr && wdy != nullptr;
    IPDE_CHECK_ARG(ctx, (wfx == nullptr) == (wfy == nullptr));
    IPDE_CHECK_ARG(ctx, (wdx == nullptr) == (wdy == nullptr));
    IPDE_CHECK_ARG(ctx, slp || dlp);
    IPDE_CHECK_ARG(ctx, !dlp || (nx != nullptr && ny != nullptr));
    if (N == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, tx && ty && out_u && out_v);
    IPDE_CHECK_ARG(ctx, ns > 0 && sx && sy);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    PackArgs pa{};
    stokes_pack_args(ctx, pa, sx, sy, wfx, wfy, nx, ny, wdx, wdy);
    const double* rec;
    const ApplyParams* prm;
    IPDE_TRY(ipde_layer_prepare(ctx, pa, ns, tx, ty, (int64_t)M * N, &rec, &prm));
    if (slp && dlp) return launch_stokes_columns_far<3>(ctx, rec, ns, M, N, tx, ty, out_u, out_v, out_p, prm);
    if (slp) return launch_stokes_columns_far<MODE_SLP>(ctx, rec, ns, M, N, tx, ty, out_u, out_v, out_p, prm);
    return launch_stokes_columns_far<MODE_DLP>(ctx, rec, ns, M, N, tx, ty, out_u, out_v, out_p, prm);
}
