// Dense LU substitution on the GPU for the QFS collocation systems (SURVEY §8f rank 2;
// the reference's third-party `qfs` package solves them with host LAPACK).
//
// The factors come from rocSOLVER (torch.linalg.lu_factor) and are fine; the library
// triangular solve that follows (rocBLAS TRSM with inverted diagonal blocks) is not
// backward stable on these matrices (cond ~1e12: residual 1.3e-9, against 4e-14 for
// plain substitution with the SAME factors — measured), and a single right-hand side
// makes it latency bound (3.2 ms at n = 4096).  Here: blocked plain substitution,
// 64-row blocks.  Step k is one launch: every workgroup re-solves the 64x64 diagonal
// block k (LDS, one wave, 64 shuffle steps — redundant but free) and subtracts its own
// off-diagonal block times x_k from the running right-hand side; workgroup 0 stores
// x_k.  No inter-workgroup waiting inside a kernel; 2 n/64 launches per solve.
//
// Layout: the factors are handed over TILED — 64x64 tiles stored contiguously (32 KB),
// tile (I, K) at ((I nb + K) 4096) doubles, padded with identity to a multiple of 64 —
// because a step reads one tile per block row: in the row-major matrix that is a 512-byte
// piece every n*8 bytes (one DRAM page / TLB entry per row; 13.5 us per step measured at
// n = 4096), tiled it is one contiguous 32 KB read per workgroup.  Inside a tile the
// storage is COLUMN-major (element (r, c) at c*64 + r): lane = row, so every load
// instruction of a wave reads one 512-byte column.
//
// Replaying the launches of a solve from a hipGraph changed nothing (0.589 vs 0.581 ms at
// n = 4096 — measured: the host is not the limit), so they are simply launched one by one.
#include "ipde_common.h"

namespace {

constexpr int DB = 64;        // block size
constexpr int DT = 256;       // threads per workgroup

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// LOWER: unit lower triangle (forward), else upper triangle with diagonal (backward).
// grid.x = number of block rows touched in this step (first = the diagonal block itself)
template <bool LOWER>
__global__ __launch_bounds__(DT) void lu_subst_step(const double* __restrict__ LU, int n, int nb,
                                                    int k, double* __restrict__ v,
                                                    double* __restrict__ x) {
    __shared__ double xs[DB];
    __shared__ double red[DT / DB][DB];
    const int tid = threadIdx.x;
    const int r0 = k * DB;
    // this workgroup's off-diagonal tile goes into registers first: the loads are in flight
    // while wave 0 runs the 64-step dependency chain of the diagonal solve
    const int i = LOWER ? k + (int)blockIdx.x : k - (int)blockIdx.x;   // block row to update
    const int row = tid & (DB - 1), part = tid >> 6;       // part = wave: 16 columns each
    double a[16];
    if (blockIdx.x != 0) {
        const double* ap = LU + ((size_t)i * nb + k) * (DB * DB) + (size_t)part * 16 * DB + row;
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = ap[c * DB];
    }
    if (tid < DB) {
        const int lane = tid;
        // row `lane` of the diagonal tile straight into registers (one coalesced column
        // per load instruction)
        const double* dt = LU + ((size_t)k * nb + k) * (DB * DB) + lane;
        double lrow[DB];
#pragma unroll
        for (int c = 0; c < DB; ++c) lrow[c] = dt[c * DB];
        double val = (r0 + lane < n) ? v[r0 + lane] : 0.0;
        // x_j is broadcast with v_readlane (compile-time lane index under full unrolling):
        // a few cycles per step instead of a ds_bpermute round trip
        if (LOWER) {
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                double xj = readlane_f64(val, j);
                if (lane > j) val = fma(-lrow[j], xj, val);
            }
        } else {
            double dinv = 1.0;
#pragma unroll
            for (int c = 0; c < DB; ++c) dinv = (c == lane) ? lrow[c] : dinv;
            dinv = 1.0 / dinv;
#pragma unroll
            for (int j = DB - 1; j >= 0; --j) {
                if (lane == j) val *= dinv;
                double xj = readlane_f64(val, j);
                if (lane < j) val = fma(-lrow[j], xj, val);
            }
        }
        xs[lane] = val;
        if (blockIdx.x == 0 && r0 + lane < n) x[r0 + lane] = val;
    }
    if (blockIdx.x == 0) return;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s = fma(a[c], xs[part * 16 + c], s);      // xs: LDS broadcast
    red[part][row] = s;
    __syncthreads();
    const int gr = i * DB + row;
    if (part == 0 && gr < n) v[gr] -= (red[0][row] + red[1][row]) + (red[2][row] + red[3][row]);
}

// Two block rows per launch (the default).  Step K owns the 128 rows of block pair
// (2K, 2K+1): wave 0 solves the first diagonal block, all waves subtract the coupling tile
// times that solution from the second block's right-hand side, wave 1 (whose registers have
// held the second diagonal tile since the start of the kernel) solves it, then workgroup j
// updates its own pair of block rows with its four tiles.  The arithmetic — products,
// partial sums and their order — is that of two consecutive lu_subst_step launches, so the
// results are bitwise the same.  Measured: 0.534 vs 0.581 ms at n = 4096 (8.3 us per pair
// against 2 x 4.5 us): most of a step is not launch latency but the two things that cannot
// overlap — the tile loads (~2 us from HBM / Infinity Cache) and the 64-step dependency chain
// (~1 us per diagonal block) — so the pairing only saves the launch gap and the second
// tile's load latency.
template <bool LOWER>
__device__ __forceinline__ double diag_chain(const double (&lrow)[DB], int lane, double val) {
    if (LOWER) {
#pragma unroll
        for (int j = 0; j < DB; ++j) {
            double xj = readlane_f64(val, j);
            if (lane > j) val = fma(-lrow[j], xj, val);
        }
    } else {
        double dinv = 1.0;
#pragma unroll
        for (int c = 0; c < DB; ++c) dinv = (c == lane) ? lrow[c] : dinv;
        dinv = 1.0 / dinv;
#pragma unroll
        for (int j = DB - 1; j >= 0; --j) {
            if (lane == j) val *= dinv;
            double xj = readlane_f64(val, j);
            if (lane < j) val = fma(-lrow[j], xj, val);
        }
    }
    return val;
}

// Up to LU_MAXB independent systems of the same size advance in lock-step, blockIdx.y = system:
// a step is latency bound, so the second system rides along for free (the two QFS solves of
// an interface — grid side and annulus side — are issued this way).
// The systems may differ in size: step K of a pass is step K of every system that still has
// one (forward: K < its number of block pairs; backward: the pass starts at the largest
// system's last pair and the smaller ones join when K reaches theirs), so a batch costs the
// steps of its largest member.
constexpr int LU_MAXB = 8;
struct LuBatch {
    const double* lu[LU_MAXB];
    double* v[LU_MAXB];   // running right-hand side of the pass
    double* x[LU_MAXB];   // result of the pass
    int n[LU_MAXB];
};

template <bool LOWER>
__global__ __launch_bounds__(DT) void lu_subst_step2(LuBatch B, int K) {
    const int n = B.n[blockIdx.y];
    const int nbp = (n + 2 * DB - 1) / (2 * DB);
    const int nb = 2 * nbp;
    // this system has no step K, or fewer block pairs left to update than the grid is wide
    if (K >= nbp || (LOWER && (int)blockIdx.x >= nbp - K)) return;
    const double* __restrict__ LU = B.lu[blockIdx.y];
    double* __restrict__ v = B.v[blockIdx.y];
    double* __restrict__ x = B.x[blockIdx.y];
    __shared__ double xs[2 * DB];
    __shared__ double redc[DT / DB][DB];
    __shared__ double red[2][2][DT / DB][DB];
    const int tid = threadIdx.x;
    const int row = tid & (DB - 1), part = tid >> 6;
    // the two diagonal blocks in the order they are solved
    const int first = LOWER ? 2 * K : 2 * K + 1, second = LOWER ? 2 * K + 1 : 2 * K;
    const int ip = LOWER ? K + (int)blockIdx.x : K - (int)blockIdx.x;   // block pair to update
    const size_t T = (size_t)DB * DB;
    const size_t off = (size_t)part * 16 * DB + row;
    // everything this thread will need is requested before the first dependency chain starts
    double lrow[DB];
    if (part < 2) {
        const int d = part == 0 ? first : second;
        const double* dt = LU + ((size_t)d * nb + d) * T + row;
#pragma unroll
        for (int c = 0; c < DB; ++c) lrow[c] = dt[c * DB];
    }
    double cpl[16];
    {
        const double* ap = LU + ((size_t)second * nb + first) * T + off;
#pragma unroll
        for (int c = 0; c < 16; ++c) cpl[c] = ap[c * DB];
    }
    double u[2][2][16];
    if (blockIdx.x != 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const double* ap = LU + ((size_t)(2 * ip + rb) * nb + (cb == 0 ? first : second)) * T + off;
#pragma unroll
                for (int c = 0; c < 16; ++c) u[rb][cb][c] = ap[c * DB];
            }
    }
    // first diagonal block: wave 0
    if (part == 0) {
        const int g = first * DB + row;
        double val = (g < n) ? v[g] : 0.0;
        val = diag_chain<LOWER>(lrow, row, val);
        xs[row] = val;
        if (blockIdx.x == 0 && g < n) x[g] = val;
    }
    __syncthreads();
    // second block's right-hand side -= coupling tile * x_first
    {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) s = fma(cpl[c], xs[part * 16 + c], s);
        redc[part][row] = s;
    }
    __syncthreads();
    if (part == 1) {
        const int g = second * DB + row;
        double val = (g < n) ? v[g] : 0.0;
        val -= (redc[0][row] + redc[1][row]) + (redc[2][row] + redc[3][row]);
        val = diag_chain<LOWER>(lrow, row, val);
        xs[DB + row] = val;
        if (blockIdx.x == 0 && g < n) x[g] = val;
    }
    if (blockIdx.x == 0) return;
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) s = fma(u[rb][cb][c], xs[cb * DB + part * 16 + c], s);
            red[rb][cb][part][row] = s;
        }
    __syncthreads();
    if (part < 2) {
        const int rb = part;
        const int g = (2 * ip + rb) * DB + row;
        if (g < n) {
            double vv = v[g];
            vv -= (red[rb][0][0][row] + red[rb][0][1][row]) + (red[rb][0][2][row] + red[rb][0][3][row]);
            vv -= (red[rb][1][0][row] + red[rb][1][1][row]) + (red[rb][1][2][row] + red[rb][1][3][row]);
            v[g] = vv;
        }
    }
}

__global__ void permute_kernel(const double* __restrict__ b, const int* __restrict__ perm, int n,
                               double* __restrict__ v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = b[perm[i]];
}

}  // namespace

extern "C" int ipde_dense_lu_solve_batch(ipde_ctx* ctx, int nsys, const int64_t* n, const double* const* lu,
                                         const int* const* perm, const double* const* b, double* const* x) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, nsys >= 1 && nsys <= LU_MAXB && n && lu && perm && b && x);
    size_t total = 0;
    int nbp_max = 0;
    for (int s = 0; s < nsys; ++s) {
        IPDE_CHECK_ARG(ctx, n[s] > 0 && n[s] < (1 << 24) && lu[s] && perm[s] && b[s] && x[s]);
        total += (size_t)n[s];
        nbp_max = std::max(nbp_max, (int)((n[s] + 2 * DB - 1) / (2 * DB)));
    }
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, 2 * total * sizeof(double)));
    LuBatch fw{}, bw{};
    double* base = (double*)ctx->partial.p;
    for (int s = 0; s < nsys; ++s) {
        double* v = base;
        double* y = v + n[s];
        base = y + n[s];
        fw.lu[s] = bw.lu[s] = lu[s];
        fw.n[s] = bw.n[s] = (int)n[s];
        fw.v[s] = v;
        fw.x[s] = y;
        bw.v[s] = y;
        bw.x[s] = x[s];
        hipLaunchKernelGGL(permute_kernel, dim3((unsigned)((n[s] + 255) / 256)), dim3(256), 0, ctx->stream, b[s],
                           perm[s], (int)n[s], v);
    }
    if (ctx->opt_dense_pairs) {
        for (int K = 0; K < nbp_max; ++K)
            hipLaunchKernelGGL(lu_subst_step2<true>, dim3(nbp_max - K, nsys), dim3(DT), 0, ctx->stream, fw, K);
        for (int K = nbp_max - 1; K >= 0; --K)
            hipLaunchKernelGGL(lu_subst_step2<false>, dim3(K + 1, nsys), dim3(DT), 0, ctx->stream, bw, K);
    } else {
        for (int s = 0; s < nsys; ++s) {
            const int nb = 2 * (int)((n[s] + 2 * DB - 1) / (2 * DB));
            for (int k = 0; k < nb; ++k)
                hipLaunchKernelGGL(lu_subst_step<true>, dim3(nb - k), dim3(DT), 0, ctx->stream, lu[s], (int)n[s], nb,
                                   k, fw.v[s], fw.x[s]);
            for (int k = nb - 1; k >= 0; --k)
                hipLaunchKernelGGL(lu_subst_step<false>, dim3(k + 1), dim3(DT), 0, ctx->stream, lu[s], (int)n[s],
                                   nb, k, bw.v[s], bw.x[s]);
        }
    }
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_dense_lu_solve(ipde_ctx* ctx, int64_t n, const double* lu, const int* perm,
                                   const double* b, double* x) {
    return ipde_dense_lu_solve_batch(ctx, 1, &n, &lu, &perm, &b, &x);
}
