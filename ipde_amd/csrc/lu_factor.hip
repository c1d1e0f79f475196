// Blocked LU factorisation with partial pivoting of the QFS collocation matrices (SURVEY §8f
// rank 2; the reference's third-party `qfs` package factors them with host LAPACK `getrf`,
// round 1 of this library with rocSOLVER: ~37 000 column-level launches at n = 4096, 65 ms,
// host-launch bound).  Right-looking over 64-column panels, on the TILED storage the
// substitution kernels of dense.hip read (64x64 tiles contiguous, column-major inside a tile,
// identity padding to a multiple of 128) — the factors never change layout.  Per panel K:
//
//   lu_panel_kernel    ONE workgroup of 1024 threads factors the (n - 64K) x 64 panel.  The panel
//                      (2 MB at n = 4096) does not fit a CU, so it is LEFT-looking inside: W
//                      columns at a time live in registers (thread t owns the row pairs (2t, 2t+1), (2t+2048, 2t+2049), ...),
//                      are first brought up to date against the finished columns of the panel
//                      (one pass over those columns, U block through LDS), then factored
//                      right-looking in registers: pivot search = DPP wave reduction + one
//                      LDS round that also carries the candidate rows — one barrier per
//                      column, no cross-workgroup traffic.  Pivot rule, scaling by the
//                      reciprocal and update order are LAPACK dgetf2's (first maximal |a|).
//   lu_rowswap_kernel  the panel's 64 interchanges, composed into <= 128 row moves by the panel
//                      kernel, applied to every other tile column (read all, barrier, write all).
//   lu_trsm_kernel     U(K, J) = L(K,K)^-1 A(K, J): a thread per column, L(K,K) broadcast from
//                      LDS; also leaves a row-major copy for the update's operand loads.
//   lu_panel_multi_kernel  panels of more than 4096 rows in systems beyond 8192 padded rows: the panel dealt out to
//                      ceil(rows / 512) workgroups, a row per thread with its 64 columns in registers, right-looking,
//                      one exchange per column through global slots (candidates, winner's row, diagonal row; the data
//                      is the flag: NaN-sentinel slots, agent-scope relaxed atomics, bounded spins, sticky abort word).
//   lu_update_kernel   A(I, J) -= L(I,K) U(K,J), one workgroup per tile, v_mfma_f64_16x16x4_f64:
//                      the product is formed transposed (D = U^T L^T) so that both operands and
//                      the read-modify-write of A(I, J) are contiguous per 16-lane group.  Since round 4 only for
//                      the border of a panel PAIR (tile column and tile row K0 + 1);
//   lu_update_pair_kernel  the trailing update of both panels of a pair in one pass (one read-modify-write of the
//                      trailing matrix per 128 columns), a tile per wave, 2 x 2 tiles per workgroup, XCD-aware.
//
// Measured (MI355X): see DESIGN.md §3.  Up to 32 768 padded rows (64 workgroups of 512 panel rows).
#include "ipde_common.h"

namespace {

constexpr int TB = 64;                 // tile edge
constexpr size_t TT = (size_t)TB * TB; // doubles per tile

__global__ void lu_iota_kernel(int* __restrict__ perm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = i;
}

__device__ __forceinline__ double rl_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// wave-wide maximum with DPP moves inside the 16-lane rows (quad swaps, half mirror, mirror) and
// four v_readlane across them: ~25 instructions, against six ds_bpermute round trips per operand
// for the shuffle form — the pivot search is on every column's critical path
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));       // quad_perm [1,0,3,2]
    v = fmax(v, dpp_mov_f64<0x4E>(v));       // quad_perm [2,3,0,1]
    v = fmax(v, dpp_mov_f64<0x141>(v));      // row_half_mirror
    v = fmax(v, dpp_mov_f64<0x140>(v));      // row_mirror
    return fmax(fmax(rl_f64(v, 15), rl_f64(v, 31)), fmax(rl_f64(v, 47), rl_f64(v, 63)));
}

// moves: [0, 128) destination rows, [128, 256) source rows (global row numbers) of this panel's
// composed interchanges, for lu_rowswap_kernel.
//
// One barrier per column: before it every wave publishes its best candidate TOGETHER WITH that
// row's W values, and the owner of the diagonal row publishes that row; behind it every thread
// picks the winner from the 16 candidates and already has the pivot row and the displaced row in
// LDS (ping-pong buffers: a fast thread's publication for column j+1 must not overwrite what a
// slow thread still reads for column j).  The interchanges of the panel's OTHER columns (finished
// ones and those still raw) are not done one by one on the critical path: the W interchanges of a
// sub-panel are composed into <= 2W row moves and applied at its end by all threads (one gather,
// one barrier, one scatter); `perm` gets the whole panel's composed moves at the end.
#ifdef IPDE_LU_STAMPS
__device__ unsigned long long* g_lu_stamps;      // tools/lu_panel_probe.hip only
#define PSTAMP(k)                                                  \
    do {                                                           \
        if (tid == 0 && g_lu_stamps) g_lu_stamps[(k)] = wall_clock64(); \
    } while (0)
#else
#define PSTAMP(k) \
    do {          \
    } while (0)
#endif

template <int NT, int RPT, int W>
__global__ __launch_bounds__(NT) void lu_panel_kernel(double* __restrict__ T, int nb, int K,
                                                      int* __restrict__ perm, int* __restrict__ moves) {
    static_assert(RPT % 2 == 0, "rows come in adjacent pairs");
    constexpr int NW = NT / 64, HP = RPT / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = (nb - K) * TB;                       // rows of the panel
    double* __restrict__ P = T + ((size_t)K * nb + K) * TT;
    const size_t tile_step = (size_t)nb * TT;          // next tile row, same tile column
    auto at = [&](int q, int c) -> double* { return P + (size_t)(q >> 6) * tile_step + (size_t)c * TB + (q & 63); };
    // Thread t owns the row PAIRS (2t, 2t+1), (2t + 2NT, 2t + 2NT + 1), ...: two adjacent rows are
    // 16 contiguous bytes of a tile column, so every access of the panel is a 16-byte one — the
    // left-looking pass is bound by the bytes one CU keeps in flight, not by its arithmetic.
    auto rowq = [&](int i) -> int { return 2 * tid + (i >> 1) * 2 * NT + (i & 1); };
    auto owner_tid = [&](int q) -> int { return (q % (2 * NT)) >> 1; };
    auto owner_i = [&](int q) -> int { return 2 * (q / (2 * NT)) + (q & 1); };

    __shared__ double sh_u[TB * W];          // U block of the left-looking update: [k][c]
    __shared__ double2 sh_l2[TB * TB / 2];   // finished columns of the diagonal tile (column-major)
    double* sh_l = (double*)sh_l2;
    __shared__ double sh_cand[2][NW][W];     // each wave's candidate row
    __shared__ double sh_diag[2][W];         // the diagonal row as it is before the interchange
    __shared__ double sh_cv[2][NW];
    __shared__ int sh_cq[2][NW];
    __shared__ int sh_piv[TB];
    __shared__ int sh_mdst[2 * W], sh_msrc[2 * W], sh_mn;   // a sub-panel's composed moves

    double a[RPT][W];
    for (int c0 = 0; c0 < TB; c0 += W) {
        PSTAMP((c0 / W) * 16 + 0);
        // --- the W columns' raw values
#pragma unroll
        for (int h = 0; h < HP; ++h) {
            const int q = rowq(2 * h);
#pragma unroll
            for (int c = 0; c < W; ++c) {
                double2 v = make_double2(0.0, 0.0);
                if (q < R) v = *(const double2*)at(q, c0 + c);
                a[2 * h][c] = v.x;
                a[2 * h + 1][c] = v.y;
            }
        }
        PSTAMP((c0 / W) * 16 + 1);
        if (c0 > 0) {
            // --- U block: rows k < c0 of these columns, forward substitution with the unit lower
            // triangle of the finished columns.  Those rows sit in wave 0 (lane l: rows 2l, 2l+1);
            // the W chains run interleaved, x_m broadcast with v_readlane.  The triangle comes from
            // LDS, fetched by all threads in one batch: read from memory inside the chain it was a
            // load latency per step (44 us of a 256-row panel).
            for (int e = tid; e < c0 * (TB / 2); e += NT) {
                const int m = e / (TB / 2), r2 = e % (TB / 2);
                *(double2*)&sh_l[m * TB + 2 * r2] = *(const double2*)(P + (size_t)m * TB + 2 * r2);
            }
            __syncthreads();
            if (wave == 0) {
                for (int m2 = 0; m2 < c0 / 2; ++m2) {
                    // row 2 m2 (lane m2, first of its pair), then row 2 m2 + 1
                    double2 l = make_double2(0.0, 0.0);
                    if (2 * lane + 1 > 2 * m2 && 2 * lane < c0) l = *(const double2*)&sh_l[(2 * m2) * TB + 2 * lane];
                    const double l0 = (2 * lane > 2 * m2) ? l.x : 0.0;          // row 2 lane
                    const double l1 = (2 * lane + 1 > 2 * m2 && 2 * lane + 1 < c0) ? l.y : 0.0;   // row 2 lane + 1
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        const double xm = rl_f64(a[0][c], m2);
                        a[0][c] = fma(-l0, xm, a[0][c]);
                        a[1][c] = fma(-l1, xm, a[1][c]);
                    }
                    double2 g = make_double2(0.0, 0.0);
                    if (2 * lane + 1 > 2 * m2 + 1 && 2 * lane < c0) g = *(const double2*)&sh_l[(2 * m2 + 1) * TB + 2 * lane];
                    const double g0 = (2 * lane > 2 * m2 + 1) ? g.x : 0.0;
                    const double g1 = (2 * lane + 1 > 2 * m2 + 1 && 2 * lane + 1 < c0) ? g.y : 0.0;
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        const double xm = rl_f64(a[1][c], m2);
                        a[0][c] = fma(-g0, xm, a[0][c]);
                        a[1][c] = fma(-g1, xm, a[1][c]);
                    }
                }
                if (2 * lane < c0) {
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        sh_u[(2 * lane) * W + c] = a[0][c];
                        sh_u[(2 * lane + 1) * W + c] = a[1][c];
                        *(double2*)at(2 * lane, c0 + c) = make_double2(a[0][c], a[1][c]);
                    }
                }
            }
            __syncthreads();
            PSTAMP((c0 / W) * 16 + 2);
            // --- rows below: a -= L[q][0:c0] U   (eight k per batch: HP x 8 16-byte loads in flight)
            constexpr int KB = HP <= 2 ? 4 : 2;      // c0 is a multiple of W, hence of KB; KB x HP 16-byte loads in flight
            for (int k0 = 0; k0 < c0; k0 += KB) {
                double2 l[KB][HP];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                    for (int h = 0; h < HP; ++h) {
                        const int q = rowq(2 * h);
                        l[kk][h] = (q >= c0 && q < R) ? *(const double2*)at(q, k0 + kk) : make_double2(0.0, 0.0);
                    }
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    double u[W];
#pragma unroll
                    for (int c = 0; c < W; ++c) u[c] = sh_u[(k0 + kk) * W + c];
#pragma unroll
                    for (int h = 0; h < HP; ++h) {
                        const int q = rowq(2 * h);
                        if (q >= c0 && q < R) {      // (c0 is even: a pair is above or below together)
#pragma unroll
                            for (int c = 0; c < W; ++c) {
                                a[2 * h][c] = fma(-l[kk][h].x, u[c], a[2 * h][c]);
                                a[2 * h + 1][c] = fma(-l[kk][h].y, u[c], a[2 * h + 1][c]);
                            }
                        }
                    }
                }
            }
        }
        // --- right-looking factorisation of the W columns in registers
        int pvt[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int d = c0 + j;                      // diagonal position (panel row and column)
            const int buf = j & 1;
            PSTAMP((c0 / W) * 16 + 3 + j);
            // pivot search: first row of maximal |a| among rows >= d
            double bv = -1.0;
            int bq = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int q = rowq(i);
                const double v = fabs(a[i][j]);
                if (q >= d && q < R && v > bv) {       // rows of a thread ascend: strict > keeps the first
                    bv = v;
                    bq = q;
                }
            }
            {
                const double wmax = wave_max_f64(bv);
                unsigned long long tie = __ballot(bv == wmax && bq != 0x7fffffff);
                int wq = 0x7fffffff;
                while (tie) {                          // (one lane, except on exact ties)
                    const int l = __ffsll((long long)tie) - 1;
                    const int cq = __builtin_amdgcn_readlane(bq, l);
                    wq = cq < wq ? cq : wq;
                    tie &= tie - 1;
                }
                bv = wmax;
                bq = wq;
            }
            if (lane == 0) {
                sh_cv[buf][wave] = bv;
                sh_cq[buf][wave] = bq;
            }
            if (bq != 0x7fffffff && tid == owner_tid(bq)) {      // the wave's candidate row, by its owner
                const int ib = owner_i(bq);
#pragma unroll
                for (int i = 0; i < RPT; ++i)
                    if (i == ib) {
#pragma unroll
                        for (int c = 0; c < W; ++c) sh_cand[buf][wave][c] = a[i][c];
                    }
            }
            if (tid == owner_tid(d)) {
#pragma unroll
                for (int c = 0; c < W; ++c) sh_diag[buf][c] = (d & 1) ? a[1][c] : a[0][c];
            }
            __syncthreads();
            // the winner among the waves' candidates: lane w of every wave takes candidate w, then
            // the same DPP maximum (a serial scan of the 16 LDS entries was 1 us of every column)
            int p = 0x7fffffff, pw = 0;
            {
                double cvv = -2.0;
                int cqq = 0x7fffffff;
                if (lane < NW) {
                    cvv = sh_cv[buf][lane];
                    cqq = sh_cq[buf][lane];
                }
                const double gmax = wave_max_f64(cvv);
                unsigned long long tie = __ballot(cvv == gmax && cqq != 0x7fffffff);
                while (tie) {
                    const int l = __ffsll((long long)tie) - 1;
                    const int cq = __builtin_amdgcn_readlane(cqq, l);
                    if (cq < p) {
                        p = cq;
                        pw = l;
                    }
                    tie &= tie - 1;
                }
            }
            double prow[W];
            if (p == 0x7fffffff) {                     // (an all-NaN column: keep the diagonal)
                p = d;
#pragma unroll
                for (int c = 0; c < W; ++c) prow[c] = sh_diag[buf][c];
            } else {
#pragma unroll
                for (int c = 0; c < W; ++c) prow[c] = sh_cand[buf][pw][c];
            }
            if (p != d) {
                if (tid == owner_tid(d)) {
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        if (d & 1)
                            a[1][c] = prow[c];
                        else
                            a[0][c] = prow[c];
                    }
                }
                if (tid == owner_tid(p)) {
                    const int ip = owner_i(p);
#pragma unroll
                    for (int i = 0; i < RPT; ++i)
                        if (i == ip) {
#pragma unroll
                            for (int c = 0; c < W; ++c) a[i][c] = sh_diag[buf][c];
                        }
                }
            }
            pvt[j] = p;
            if (tid == NT - 1) sh_piv[d] = p;        // (for the panel's composed moves, read behind later barriers)
            const double pv = prow[j];
            const double rec = pv != 0.0 ? 1.0 / pv : 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int q = rowq(i);
                if (q > d && q < R) {
                    const double l = a[i][j] * rec;
                    a[i][j] = l;
#pragma unroll
                    for (int c = j + 1; c < W; ++c) a[i][c] = fma(-l, prow[c], a[i][c]);
                }
            }
        }
        PSTAMP((c0 / W) * 16 + 3 + W);
        // --- the finished columns go back (rows < c0 were written with the U block)
#pragma unroll
        for (int h = 0; h < HP; ++h) {
            const int q = rowq(2 * h);
            if (q >= c0 && q < R) {
#pragma unroll
                for (int c = 0; c < W; ++c) *(double2*)at(q, c0 + c) = make_double2(a[2 * h][c], a[2 * h + 1][c]);
            }
        }
        PSTAMP((c0 / W) * 16 + 4 + W);
        // The sub-panel's W interchanges composed into 2W row moves, in parallel: thread e takes a
        // touched position (a diagonal position or a pivot's row) and walks the interchanges
        // BACKWARDS to the position whose row ends up there.  (A position met twice gives the same
        // move twice.)  Every thread knows the sub-panel's pivots (pvt): no LDS, no ordering question.
        if (tid < 2 * W) {
            int fin = c0 + tid;
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (tid == W + j) fin = pvt[j];
            int pos = fin;
#pragma unroll
            for (int j = W - 1; j >= 0; --j) {
                const int d = c0 + j;
                pos = pos == d ? pvt[j] : (pos == pvt[j] ? d : pos);
            }
            sh_mdst[tid] = fin;
            sh_msrc[tid] = pos;
        }
        if (tid == NT - 1) sh_mn = 2 * W;
        __syncthreads();
        PSTAMP((c0 / W) * 16 + 5 + W);
        // --- this sub-panel's interchanges on the panel's other columns: thread = (move, column)
        {
            constexpr int PASSES = (2 * W * TB + NT - 1) / NT;     // (move, column) pairs per thread
            double v[PASSES];
            bool act[PASSES];
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int e = (tid + ps * NT) >> 6, c = tid & 63;
                act[ps] = e < sh_mn && (c < c0 || c >= c0 + W) && sh_mdst[e] != sh_msrc[e];
                v[ps] = act[ps] ? *at(sh_msrc[e], c) : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int e = (tid + ps * NT) >> 6, c = tid & 63;
                if (act[ps]) *at(sh_mdst[e], c) = v[ps];
            }
        }
        __syncthreads();
    }
    // composed interchanges of this panel as row moves dst <- src, and the same moves on perm
    if (tid < 2 * TB) {
        const int q = tid < TB ? tid : sh_piv[tid - TB];
        int pos = q;                               // backwards through the panel's 64 interchanges
        for (int j = TB - 1; j >= 0; --j) {
            const int pj = sh_piv[j];
            pos = pos == j ? pj : (pos == pj ? j : pos);
        }
        const int dst = K * TB + q, src = K * TB + pos;
        moves[tid] = dst;
        moves[2 * TB + tid] = src;
        const int pv = perm[src];
        __syncthreads();
        perm[dst] = pv;
    } else {
        __syncthreads();
    }
}

// ---- the panel across several CUs (systems beyond 8192 padded rows) --------------------------------------------
// The single-workgroup panel above keeps W columns in registers and is left-looking over the rest: at 19 200 rows
// (BASELINE configs[4]'s Stokes QFS systems) one CU would re-read 157 MB per panel, 0.85 s for the 300 panels.
// Here the panel's rows are dealt out to G workgroups of 512 threads, a ROW PER THREAD with all 64 columns in
// registers (128 VGPRs): plain right-looking elimination, nothing is re-read.  What crosses workgroups is the
// pivot search — once per column every workgroup publishes its best candidate TOGETHER WITH that row's 64 values,
// workgroup 0 the diagonal row; wave 0 of every workgroup reads the G candidates, picks the winner by LAPACK's rule
// (first row of maximal |a|) and fetches its row: one exchange per column, ~3 us.  Hand-off as in the substitution
// kernels of dense.hip (cdna guide, Guideline 16: the data is the flag): every published value is one 8-byte
// agent-scope relaxed atomic store into a slot of its own, every read an agent-scope relaxed atomic load polled until
// the slot no longer holds the sentinel (a NaN pattern no arithmetic produces, filled by one memset per panel); every
// spin is bounded, a time-out sets the context's sticky abort word and every waiter leaves.  All G <= 64 workgroups
// are resident together (one per CU, 38 at 19 200 rows).
constexpr unsigned long long LUX_SENTINEL = 0xFFF7A5A5FFF7A5A5ull;
constexpr unsigned LUX_SENTINEL32 = 0xFFF7A5A5u;
constexpr unsigned LUX_SPIN_LIMIT = 1u << 24;
constexpr int LUX_REC = TB + 2;                 // a record: value, row, the row's 64 entries
typedef __attribute__((address_space(1))) unsigned long long lux_u64;
typedef __attribute__((address_space(1))) unsigned int lux_u32;

__device__ __forceinline__ double lux_wait(const double* p, unsigned* abort_word) {
    unsigned long long bits;
    for (unsigned spins = 0;;) {
        bits = __hip_atomic_load((lux_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits != LUX_SENTINEL) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
            if (__hip_atomic_load((lux_u32*)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (spins >= LUX_SPIN_LIMIT) {
                __hip_atomic_store((lux_u32*)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    return __longlong_as_double((long long)bits);
}
__device__ __forceinline__ void lux_publish(double* p, double v) {
    __hip_atomic_store((lux_u64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// (the 64 column steps are a compile-time recursion: as a loop the optimizer would not unroll a body of this size,
// and a[j] with a run-time j put the row into scratch memory)
template <int NT>
struct LuMultiPanel {
    static constexpr int NW = NT / 64;
    double a[TB];
    int tid, lane, wave, g, G, q;
    bool live;
    double* xch;
    unsigned* abort_word;
    double* sh_cv;
    int* sh_cq;
    double* sh_prow;
    double* sh_drow;
    int* sh_p;
    int* sh_piv;

    template <int J>
    __device__ __forceinline__ void step() {
        constexpr int d = J;                           // diagonal position (panel row and column)
        // this workgroup's candidate: first row of maximal |a| among its rows >= d
        double bv = (live && q >= d) ? fabs(a[J]) : -1.0;
        int bq = (live && q >= d && bv >= 0.0) ? q : 0x7fffffff;
        if (bq == 0x7fffffff) bv = -1.0;               // (a NaN entry: never a candidate)
        {
            const double wmax = wave_max_f64(bv);
            unsigned long long tie = __ballot(bv == wmax && bq != 0x7fffffff);
            int wq = 0x7fffffff;
            while (tie) {
                const int l = __ffsll((long long)tie) - 1;
                const int cq = __builtin_amdgcn_readlane(bq, l);
                wq = cq < wq ? cq : wq;
                tie &= tie - 1;
            }
            if (lane == 0) {
                sh_cv[wave] = wq == 0x7fffffff ? -1.0 : wmax;
                sh_cq[wave] = wq;
            }
        }
        __syncthreads();
        double gv = -1.0;
        int gq = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const double cv = sh_cv[w];
            const int cq = sh_cq[w];
            if (cq != 0x7fffffff && (cv > gv || (cv == gv && cq < gq))) {
                gv = cv;
                gq = cq;
            }
        }
        double* rec = xch + ((size_t)J * (G + 1) + g) * LUX_REC;
        if (live && q == gq) {                         // the candidate row, by its owner, then its value (the flag last)
#pragma unroll
            for (int c = 0; c < TB; ++c) lux_publish(rec + 2 + c, a[c]);
            lux_publish(rec + 1, (double)gq);
            lux_publish(rec, gv);
        }
        if (gq == 0x7fffffff && tid == 0) {            // nothing to offer (all its rows above the diagonal, or NaN)
            lux_publish(rec + 1, 2147483647.0);
            lux_publish(rec, -1.0);
        }
        if (g == 0 && tid == d) {                      // the diagonal row as it is before the interchange
            double* rd = xch + ((size_t)J * (G + 1) + G) * LUX_REC;
#pragma unroll
            for (int c = 0; c < TB; ++c) lux_publish(rd + 2 + c, a[c]);
        }
        if (wave == 0) {
            double cv = -2.0;
            int cq = 0x7fffffff;
            if (lane < G) {
                const double* r = xch + ((size_t)J * (G + 1) + lane) * LUX_REC;
                cv = lux_wait(r, abort_word);
                cq = (int)lux_wait(r + 1, abort_word);
            }
            const double gmax = wave_max_f64(cv);
            unsigned long long tie = __ballot(cv == gmax && cq != 0x7fffffff && cv >= 0.0);
            int p = 0x7fffffff, pw = 0;
            while (tie) {
                const int l = __ffsll((long long)tie) - 1;
                const int c2 = __builtin_amdgcn_readlane(cq, l);
                if (c2 < p) {
                    p = c2;
                    pw = l;
                }
                tie &= tie - 1;
            }
            const double dr = lux_wait(xch + ((size_t)J * (G + 1) + G) * LUX_REC + 2 + lane, abort_word);
            double pr = dr;
            if (p == 0x7fffffff)
                p = d;                                 // (an all-NaN column: keep the diagonal)
            else if (p != d)
                pr = lux_wait(xch + ((size_t)J * (G + 1) + pw) * LUX_REC + 2 + lane, abort_word);
            sh_prow[lane] = pr;
            sh_drow[lane] = dr;
            if (lane == 0) {
                *sh_p = p;
                if (g == 0) sh_piv[d] = p;
            }
        }
        __syncthreads();
        const int p = *sh_p;
        if (p != d && live) {
            if (q == d) {
#pragma unroll
                for (int c = 0; c < TB; ++c) a[c] = sh_prow[c];
            } else if (q == p) {
#pragma unroll
                for (int c = 0; c < TB; ++c) a[c] = sh_drow[c];
            }
        }
        const double pv = sh_prow[J];
        const double rcp = pv != 0.0 ? 1.0 / pv : 0.0;
        if (live && q > d) {
            const double l = a[J] * rcp;
            a[J] = l;
#pragma unroll
            for (int c = J + 1; c < TB; ++c) a[c] = fma(-l, sh_prow[c], a[c]);
        }
        if constexpr (J + 1 < TB) step<J + 1>();
    }
};

template <int NT>
__global__ __launch_bounds__(NT) void lu_panel_multi_kernel(double* __restrict__ T, int nb, int K,
                                                            int* __restrict__ perm, int* __restrict__ moves,
                                                            double* __restrict__ xch, unsigned* abort_word) {
    constexpr int NW = NT / 64;
    __shared__ double sh_cv[NW];
    __shared__ int sh_cq[NW];
    __shared__ double sh_prow[TB], sh_drow[TB];
    __shared__ int sh_p;
    __shared__ int sh_piv[TB];
    LuMultiPanel<NT> S;
    S.tid = threadIdx.x;
    S.lane = S.tid & 63;
    S.wave = S.tid >> 6;
    S.g = blockIdx.x;
    S.G = gridDim.x;
    const int R = (nb - K) * TB;
    S.q = S.g * NT + S.tid;                            // this thread's row of the panel
    S.live = S.q < R;
    S.xch = xch;
    S.abort_word = abort_word;
    S.sh_cv = sh_cv;
    S.sh_cq = sh_cq;
    S.sh_prow = sh_prow;
    S.sh_drow = sh_drow;
    S.sh_p = &sh_p;
    S.sh_piv = sh_piv;
    double* __restrict__ P = T + ((size_t)K * nb + K) * TT;
    const size_t tile_step = (size_t)nb * TT;
    double* rowp = P + (size_t)(S.q >> 6) * tile_step + (S.q & 63);      // + c * TB: column c
#pragma unroll
    for (int c = 0; c < TB; ++c) S.a[c] = S.live ? rowp[(size_t)c * TB] : 0.0;
    S.template step<0>();
    if (S.live) {
#pragma unroll
        for (int c = 0; c < TB; ++c) rowp[(size_t)c * TB] = S.a[c];
    }
    // composed interchanges of this panel as row moves dst <- src, and the same moves on perm (workgroup 0)
    if (S.g == 0) {
        const int tid = S.tid;
        __syncthreads();
        if (tid < 2 * TB) {
            const int qq = tid < TB ? tid : sh_piv[tid - TB];
            int pos = qq;                              // backwards through the panel's 64 interchanges
            for (int j = TB - 1; j >= 0; --j) {
                const int pj = sh_piv[j];
                pos = pos == j ? pj : (pos == pj ? j : pos);
            }
            const int dst = K * TB + qq, src = K * TB + pos;
            moves[tid] = dst;
            moves[2 * TB + tid] = src;
            const int pv = perm[src];
            __syncthreads();
            perm[dst] = pv;
        } else {
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(128) void lu_rowswap_kernel(double* __restrict__ T, int nb, int K,
                                                         const int* __restrict__ moves) {
    const int J = (int)blockIdx.x < K ? (int)blockIdx.x : (int)blockIdx.x + 1;     // every tile column but K
    const int m = threadIdx.x;
    const int dst = moves[m], src = moves[2 * TB + m];
    const bool act = dst != src;
    double v[TB];
    if (act) {
        const double* s = T + ((size_t)(src >> 6) * nb + J) * TT + (src & 63);
#pragma unroll
        for (int c = 0; c < TB; ++c) v[c] = s[(size_t)c * TB];
    }
    __syncthreads();
    if (act) {
        double* d = T + ((size_t)(dst >> 6) * nb + J) * TT + (dst & 63);
#pragma unroll
        for (int c = 0; c < TB; ++c) d[(size_t)c * TB] = v[c];
    }
}

// Ur: row-major copies of the U tiles of this block row, tile J at Ur + J * TT
__global__ __launch_bounds__(64) void lu_trsm_kernel(double* __restrict__ T, int nb, int K,
                                                     double* __restrict__ Ur) {
    const int J = K + 1 + (int)blockIdx.x;
    const int c = threadIdx.x;
    __shared__ double sL[TT];
    const double* L = T + ((size_t)K * nb + K) * TT;
    for (int k = 0; k < TB; ++k) sL[k * TB + c] = L[(size_t)k * TB + c];      // (column k, row c)
    double* A = T + ((size_t)K * nb + J) * TT + (size_t)c * TB;
    double x[TB];
#pragma unroll
    for (int r = 0; r < TB; ++r) x[r] = A[r];
    __syncthreads();
    // column by column (x[m] final -> all rows below): the 63 - m updates of a step are independent
    // of each other, the sums of a row still run over m ascending
#pragma unroll
    for (int m = 0; m < TB - 1; ++m) {
        const double xm = x[m];
#pragma unroll
        for (int r = m + 1; r < TB; ++r) x[r] = fma(-sL[m * TB + r], xm, x[r]);
    }
    double* U = Ur + (size_t)J * TT;
#pragma unroll
    for (int r = 0; r < TB; ++r) {
        A[r] = x[r];
        U[(size_t)r * TB + c] = x[r];
    }
}

typedef double d4 __attribute__((ext_vector_type(4)));

// (I0, J0: the first tile row / column of the launch's tiles; the border launches of a panel pair run one tile
// column and one tile row)
__global__ __launch_bounds__(256) void lu_update_kernel(double* __restrict__ T, int nb, int K,
                                                        const double* __restrict__ Ur, int I0, int J0) {
    const int I = I0 + (int)blockIdx.y, J = J0 + (int)blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    const double* __restrict__ L = T + ((size_t)I * nb + K) * TT;      // column-major: (i, k) at k*64 + i
    const double* __restrict__ U = Ur + (size_t)J * TT;                // row-major: (k, j) at k*64 + j
    double* __restrict__ C = T + ((size_t)I * nb + J) * TT;            // (i, j) at j*64 + i
    // D = U^T L^T blockwise: lane holds D[row j' = hi + 4 reg][col i' = lo] of the 16 x 16 block
    // (j-block jb, i-block w), i.e. C(16 w + lo, 16 jb + hi + 4 reg)
    d4 acc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][r] = C[(size_t)(16 * jb + hi + 4 * r) * TB + 16 * w + lo];
#pragma unroll 4
    for (int s = 0; s < TB / 4; ++s) {
        const int k = 4 * s + hi;
        const double b = -L[(size_t)k * TB + 16 * w + lo];            // B operand: L^T[k][i'], negated
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const double a = U[(size_t)k * TB + 16 * jb + lo];        // A operand: U^T[j'][k]
            acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[jb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(size_t)(16 * jb + hi + 4 * r) * TB + 16 * w + lo] = acc[jb][r];
}

// The trailing update of a PAIR of panels K0, K0 + 1 in one pass:  A(I, J) -= L(I,K0) U(K0,J) + L(I,K0+1) U(K0+1,J)
// for I, J > K0 + 1 — one read-modify-write of the trailing matrix per 128 columns instead of per 64 (at 19 200 rows
// the single-panel form moved 590 GB of A per factorisation: the HBM floor of the whole factorisation), the same
// MFMA chain per element (k ascending through both panels), hence the same bits as two single-panel updates.
// A wave owns a whole 64 x 64 tile (16 accumulator blocks: eight operand loads feed sixteen MFMAs; the one-tile-per-
// workgroup form above loads five for four), a workgroup 2 x 2 tiles.  Operands straight from L2, one k-step ahead
// of the MFMAs.  The hand-out is XCD-aware: workgroup b runs on XCD b mod 8, which is given a contiguous band of the
// tile-row pairs — its share of the two L panels (2.4 MB at 19 200 rows) stays in that XCD's L2 — and walks a column
// pair's tiles band-first, so a U tile pair is fetched once per XCD.
template <int OCC>
__global__ __launch_bounds__(256, OCC) void lu_update_pair_kernel(double* __restrict__ T, int nb, int K0,
                                                             const double* __restrict__ Ur, int gx, int gy) {
    const int wg = (int)blockIdx.x, xcd = wg & 7, slot = wg >> 3;
    const int iy0 = (xcd * gy) >> 3, nI = (((xcd + 1) * gy) >> 3) - iy0;
    if (nI <= 0) return;
    const int jx = slot / nI, iy = iy0 + slot % nI;
    if (jx >= gx) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    const int I = K0 + 2 + 2 * iy + (w >> 1), J = K0 + 2 + 2 * jx + (w & 1);
    if (I >= nb || J >= nb) return;                                     // (wave-uniform; no barriers below)
    double* __restrict__ C = T + ((size_t)I * nb + J) * TT;
    const double* __restrict__ L = T + ((size_t)I * nb + K0) * TT;     // panel p: + p TT (the next tile of the row)
    const double* __restrict__ U = Ur + (size_t)J * TT;                // panel p: + p nb TT
    const size_t ustep = (size_t)nb * TT;
    // acc[ib][jb], reg r: C(16 ib + lo, 16 jb + hi + 4 r)
    d4 acc[4][4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ib][jb][r] = C[(size_t)(16 * jb + hi + 4 * r) * TB + 16 * ib + lo];
    auto fetch = [&](int q, double (&a)[4], double (&b)[4]) {          // k-step q of 32: panel q / 16, k = 4 (q % 16) + hi
        const int p = q >> 4, k = 4 * (q & 15) + hi;
        const double* Lp = L + (size_t)p * TT + (size_t)k * TB + lo;
        const double* Up = U + (size_t)p * ustep + (size_t)k * TB + lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b[i] = Lp[16 * i];
            a[i] = Up[16 * i];
        }
    };
    auto product = [&](const double (&a)[4], const double (&b)[4]) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const double nb_ = -b[ib];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jb], nb_, acc[ib][jb], 0, 0, 0);
        }
    };
    double a0[4], b0[4], a1[4], b1[4];
    fetch(0, a0, b0);
#pragma unroll 1
    for (int q = 0; q < 32; q += 2) {
        fetch(q + 1, a1, b1);
        product(a0, b0);
        if (q + 2 < 32) fetch(q + 2, a0, b0);
        product(a1, b1);
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(16 * jb + hi + 4 * r) * TB + 16 * ib + lo] = acc[ib][jb][r];
}

}  // namespace

extern "C" int ipde_dense_lu_factor(ipde_ctx* ctx, int64_t n_pad, double* tiles, int* perm) {
    if (!ctx) return IPDE_ERR_INVALID;
    constexpr int MNT = 512;                            // threads (= rows) per workgroup of the multi-CU panel
    IPDE_CHECK_ARG(ctx, tiles && perm && n_pad >= 128 && n_pad % 128 == 0 && n_pad <= 64 * MNT);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int nb = (int)(n_pad / TB);
    // (IPDE_LU_FORCE_MULTI: the multi-CU panel for every panel of more than 512 rows — the tests' way to run it
    // against host LAPACK at small sizes)
    static const bool force_multi = getenv("IPDE_LU_FORCE_MULTI") != nullptr;
    const bool multi = n_pad > 8192 || force_multi;
    const int multi_min_rows = force_multi ? 512 : 4096;
    const int Gmax = (int)((n_pad + MNT - 1) / MNT);
    // scratch: row-major U tiles of the current block row, the panel's row moves, and (multi-CU panel) the exchange
    // records of a panel: 64 columns x (G + 1) records
    const size_t xch_doubles = multi ? (size_t)TB * (Gmax + 1) * LUX_REC : 0;
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->lu_work,
                                 (2 * (size_t)nb * TT + xch_doubles) * sizeof(double) + 4 * TB * sizeof(int)));
    double* Ur = (double*)ctx->lu_work.p;               // two block rows: the panels of a pair
    double* xch = Ur + 2 * (size_t)nb * TT;
    int* moves = (int*)(xch + xch_doubles);
    if (multi && !ctx->d_lu_abort) {
        IPDE_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_lu_abort, 16));
        IPDE_HIP_CHECK(ctx, hipMemset(ctx->d_lu_abort, 0, 16));
        *(volatile unsigned*)(ctx->h_pinned + ctx->h_pinned_bytes / sizeof(double) - 1) = 0;
    }
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(lu_iota_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, perm, (int)n_pad);
    auto panel = [&](int K) -> int {
        const int R = (nb - K) * TB;
        if (multi && R > multi_min_rows) {
            const int G = (R + MNT - 1) / MNT;
            IPDE_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)xch, (int)LUX_SENTINEL32,
                                                  (size_t)TB * (G + 1) * LUX_REC * 2, st));
            hipLaunchKernelGGL((lu_panel_multi_kernel<MNT>), dim3(G), dim3(MNT), 0, st, tiles, nb, K, perm, moves, xch,
                               ctx->d_lu_abort);
        } else if (n_pad <= 4096 || R <= 4096) {
            // 1024 threads of 128 VGPRs: the sub-panel (rows per thread x W doubles) is half of that
            hipLaunchKernelGGL((lu_panel_kernel<1024, 4, 8>), dim3(1), dim3(1024), 0, st, tiles, nb, K, perm, moves);
        } else {
            hipLaunchKernelGGL((lu_panel_kernel<1024, 8, 4>), dim3(1), dim3(1024), 0, st, tiles, nb, K, perm, moves);
        }
        if (nb > 1) hipLaunchKernelGGL(lu_rowswap_kernel, dim3(nb - 1), dim3(128), 0, st, tiles, nb, K, moves);
        return IPDE_OK;
    };
    // Panels in pairs (nb is even): panel K0, its U block row, its update of tile column K0 + 1 only; panel K0 + 1 on
    // the updated column; panel K0's update of tile row K0 + 1, that row's U blocks; then one update of the trailing
    // matrix with both panels (lu_update_pair_kernel).  The interchanges of panel K0 + 1 permute rows of a trailing
    // matrix that has not seen panel K0's update yet: they permute L(., K0) with it, so the update commutes — as long
    // as every row gets it AFTER the interchanges, tile row K0 + 1 included.  Per element the same operations in the
    // same order as panel-by-panel updates.
    static const bool single_updates = getenv("IPDE_LU_SINGLE_UPDATES") != nullptr;      // (A/B: the round-3 schedule)
    static const bool pair_occ2 = getenv("IPDE_LU_PAIR_OCC1") == nullptr;                // (two workgroups per CU: 243 against 262 ms at 19 200 rows, spills and all)
    for (int K0 = 0; K0 < nb; K0 += 2) {
        const int K1 = K0 + 1, rest0 = nb - K0 - 1, rest1 = nb - K1 - 1;
        double* Ur0 = Ur;
        double* Ur1 = Ur + (size_t)nb * TT;
        IPDE_TRY(panel(K0));
        hipLaunchKernelGGL(lu_trsm_kernel, dim3(rest0), dim3(64), 0, st, tiles, nb, K0, Ur0);
        if (single_updates) {
            hipLaunchKernelGGL(lu_update_kernel, dim3(rest0, rest0), dim3(256), 0, st, tiles, nb, K0, Ur0, K0 + 1, K0 + 1);
            IPDE_TRY(panel(K1));
            if (rest1 > 0) {
                hipLaunchKernelGGL(lu_trsm_kernel, dim3(rest1), dim3(64), 0, st, tiles, nb, K1, Ur0);
                hipLaunchKernelGGL(lu_update_kernel, dim3(rest1, rest1), dim3(256), 0, st, tiles, nb, K1, Ur0, K1 + 1, K1 + 1);
            }
            continue;
        }
        hipLaunchKernelGGL(lu_update_kernel, dim3(1, rest0), dim3(256), 0, st, tiles, nb, K0, Ur0, K0 + 1, K1);
        IPDE_TRY(panel(K1));
        if (rest1 > 0) {
            // (tile row K1 takes panel K0's update only now: panel K1's interchanges bring rows up from below, and
            // those have not seen it)
            hipLaunchKernelGGL(lu_update_kernel, dim3(rest1, 1), dim3(256), 0, st, tiles, nb, K0, Ur0, K1, K1 + 1);
            hipLaunchKernelGGL(lu_trsm_kernel, dim3(rest1), dim3(64), 0, st, tiles, nb, K1, Ur1);
            const int g2 = (rest1 + 1) / 2, maxI = (g2 + 7) / 8;
            if (pair_occ2)
                hipLaunchKernelGGL(lu_update_pair_kernel<2>, dim3((unsigned)(8 * maxI * g2)), dim3(256), 0, st, tiles, nb,
                                   K0, Ur, g2, g2);
            else
                hipLaunchKernelGGL(lu_update_pair_kernel<1>, dim3((unsigned)(8 * maxI * g2)), dim3(256), 0, st, tiles, nb,
                                   K0, Ur, g2, g2);
        }
    }
    if (multi)      // a waiter that gave up leaves the abort word set: the next library call reports it (ipde_ctx_sync)
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_pinned + ctx->h_pinned_bytes / sizeof(double) - 1, ctx->d_lu_abort, 4,
                                           hipMemcpyDeviceToHost, st));
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}
