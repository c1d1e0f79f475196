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

// The 64-step dependency chain of one diagonal block, one row per lane.  What sits between
// two steps is what bounds a substitution, so the chain carries nothing it does not need: the
// row is masked ONCE, ahead of the chain (coefficients a lane must not apply become 0, and
// fma(-0, x_j, val) == val), so a step is v_readlane x2 -> v_fma_f64 with no compare / select in
// the dependency (launch-per-step substitution, batch of six up to n = 6400: 1.11 -> 1.01 ms); in the upper
// solve the division by the diagonal is a multiplication by its reciprocal done by every lane
// every step (lane j's product is the one broadcast), and the lanes' own results are formed
// after the chain from the same operands — the same values, bit for bit.
// (A non-finite x_j would turn 0 * x_j into NaN in rows that are already final: only for a
// system whose solution is already non-finite.)
template <bool LOWER>
__device__ __forceinline__ void diag_mask(double (&lrow)[DB], int lane, double& dinv) {
    dinv = 1.0;
    if (!LOWER) {
#pragma unroll
        for (int c = 0; c < DB; ++c) dinv = (c == lane) ? lrow[c] : dinv;
        dinv = 1.0 / dinv;
    }
#pragma unroll
    for (int c = 0; c < DB; ++c) lrow[c] = (LOWER ? lane > c : lane < c) ? lrow[c] : 0.0;
}

template <bool LOWER>
__device__ __forceinline__ double diag_chain(const double (&lrow)[DB], double dinv, double val) {
    if (LOWER) {
#pragma unroll
        for (int j = 0; j < DB; ++j) {
            const double xj = readlane_f64(val, j);
            val = fma(-lrow[j], xj, val);
        }
        return val;
    }
#pragma unroll
    for (int j = DB - 1; j >= 0; --j) {
        const double xj = readlane_f64(val * dinv, j);
        val = fma(-lrow[j], xj, val);
    }
    return val * dinv;
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
        double dinv;
        diag_mask<LOWER>(lrow, lane, dinv);
        val = diag_chain<LOWER>(lrow, dinv, val);
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
    double dinv = 1.0;
    if (part < 2) {
        const int d = part == 0 ? first : second;
        const double* dt = LU + ((size_t)d * nb + d) * T + row;
#pragma unroll
        for (int c = 0; c < DB; ++c) lrow[c] = dt[c * DB];
        diag_mask<LOWER>(lrow, row, dinv);
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
        val = diag_chain<LOWER>(lrow, dinv, val);
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
        val = diag_chain<LOWER>(lrow, dinv, val);
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

// ---------------------------------------------------------------------------------------------
// One launch per triangular pass (the default, option "dense_persistent").
//
// The step-per-launch kernels above cost ~10 us per 128 rows: a dependent kernel boundary, the
// tile loads of every workgroup behind it (224 KB each, issued only once the launch starts) and
// two 64-step dependency chains.  Here workgroup p OWNS block pair p (128 rows) for the whole
// pass: it keeps the running right-hand side of its rows in registers, consumes the solutions
// x_q of the pairs before it in dependency order as they appear, then solves its own diagonal
// pair and publishes x_p.  What is left on the critical path per pair is the hand-off (~1 us),
// one 128x128 update, and the two chains; the tile loads of step q are in flight while the
// workgroup waits for x_q.
//
// Hand-off (cdna guide, Guideline 16, form R2 — the data is the flag): every x value is ONE
// 8-byte agent-scope relaxed atomic store (write-through), every read of it an agent-scope
// relaxed atomic load straight into a register; a slot holds SENTINEL (a NaN bit pattern no
// arithmetic produces; set by one hipMemsetD32Async per call) until its value exists.  No fences,
// no separate flags, and no plain load ever touches a handed-off word.
// Order: a workgroup takes its place in the dependency order from an atomic ticket, so every
// workgroup it waits for has started before it — correct for any dispatch order and any
// residency.  Every spin is bounded: on a time-out the abort word is set, all waiters leave, the
// kernel completes, and the next library call reports the failure.
// The arithmetic — products, partial sums and their order — is that of lu_subst_step2, so the
// results are bitwise the same (tests/test_dense_gpu.py).
constexpr unsigned long long LU_SENTINEL = 0xFFF7A5A5FFF7A5A5ull;
constexpr unsigned LU_SENTINEL32 = 0xFFF7A5A5u;
constexpr unsigned LU_SPIN_LIMIT = 1u << 22;    // polls of ~1 us: seconds

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct LuPersist {
    const double* lu[LU_MAXB];
    const int* perm[LU_MAXB];     // forward pass: rhs = b[perm]
    const double* b[LU_MAXB];     // forward: the caller's right-hand side; backward: the forward pass's slots
    double* slots[LU_MAXB];       // this pass's hand-off slots, 128 per block pair
    double* x[LU_MAXB];           // backward pass: the caller's result
    int n[LU_MAXB];
    int nsys;
    int nbp_max;
    unsigned* ticket;             // starts at LU_SENTINEL32
    unsigned* abort_word;         // 0 = running; set (and left set) by a waiter that gave up
#ifdef IPDE_LU_STAMPS
    unsigned long long* stamps;   // tools/lu_persist_probe.hip only: 8 clock readings per workgroup
#endif
};

#ifdef IPDE_LU_STAMPS
#define LU_STAMP(k)                                                                 \
    do {                                                                            \
        if ((tid & 63) == 0) B.stamps[((size_t)ticket * 4 + (tid >> 6)) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define LU_STAMP(k) \
    do {            \
    } while (0)
#endif

// `near`: this workgroup is next in the dependency order (its wait is the critical path): poll
// back to back.  Otherwise ~50 workgroups would hammer the same eight lines of the frontier and
// delay the one poll that matters: the others look once per microsecond — they have a whole
// step of slack.
__device__ __forceinline__ double lu_wait_slot(const double* p, unsigned* abort_word, bool near) {
    unsigned long long bits;
    for (unsigned spins = 0;;) {
        bits = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits != LU_SENTINEL) break;
        if (near)
            __builtin_amdgcn_s_sleep(1);
        else
            __builtin_amdgcn_s_sleep(32);
        if ((++spins & 255u) == 0) {
            if (__hip_atomic_load((gu32*)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (spins >= LU_SPIN_LIMIT) {
                __hip_atomic_store((gu32*)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    return __longlong_as_double((long long)bits);
}

__device__ __forceinline__ void lu_publish_slot(double* p, double v) {
    __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

template <bool LOWER>
__global__ __launch_bounds__(DT) void lu_subst_persistent(LuPersist B) {
    __shared__ unsigned sh_ticket;
    __shared__ double xs[2 * DB];
    __shared__ double redc[DT / DB][DB];
    __shared__ double red[2][2][DT / DB][DB];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned total = (unsigned)(B.nsys * B.nbp_max);
        const unsigned t = __hip_atomic_fetch_add((gu32*)B.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                           LU_SENTINEL32;
        sh_ticket = t;
        (void)total;
    }
    __syncthreads();
    const int ticket = (int)sh_ticket;
    LU_STAMP(0);
    const int sys = ticket % B.nsys, pos = ticket / B.nsys;      // position in the dependency order
    const int n = B.n[sys];
    const int nbp = (n + 2 * DB - 1) / (2 * DB);
    if (pos >= nbp) return;
    const int nb = 2 * nbp;
    const int p = LOWER ? pos : nbp - 1 - pos;                   // this workgroup's block pair
    const double* __restrict__ LU = B.lu[sys];
    double* __restrict__ slots = B.slots[sys];
    const int row = tid & (DB - 1), part = tid >> 6;
    const size_t T = (size_t)DB * DB;
    const size_t off = (size_t)part * 16 * DB + row;
    // the two diagonal blocks of a pair in the order they are solved; waves 0 and 1 own the
    // running right-hand sides of `first` and `second` of THIS pair
    const int first = LOWER ? 2 * p : 2 * p + 1, second = LOWER ? 2 * p + 1 : 2 * p;
    double vv = 0.0;
    if (part < 2) {
        const int g = (part == 0 ? first : second) * DB + row;
        if (LOWER) {
            if (g < n) vv = B.b[sys][B.perm[sys][g]];
        } else {
            vv = B.b[sys][g];        // forward slots: final (previous launch), zero in the identity padding
        }
    }
    // this pair's own diagonal blocks and coupling tile: requested now, needed when the last of
    // the earlier solutions has arrived
    double lrow[DB];
    double dinv = 1.0;
    if (part < 2) {
        const int d = part == 0 ? first : second;
        const double* dt = LU + ((size_t)d * nb + d) * T + row;
#pragma unroll
        for (int c = 0; c < DB; ++c) lrow[c] = dt[c * DB];
        diag_mask<LOWER>(lrow, row, dinv);
    }
    double cpl[16];
    {
        const double* ap = LU + ((size_t)second * nb + first) * T + off;
#pragma unroll
        for (int c = 0; c < 16; ++c) cpl[c] = ap[c * DB];
    }
    // the pairs before this one, in dependency order
    for (int step = 0; step < pos; ++step) {
        const int q = LOWER ? step : nbp - 1 - step;
        const int qf = LOWER ? 2 * q : 2 * q + 1, qs = LOWER ? 2 * q + 1 : 2 * q;
        double u[2][2][16];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                // rb = 0 / 1: the rows of `first` / `second` (held by waves 0 / 1)
                const double* ap = LU + ((size_t)(rb == 0 ? first : second) * nb + (cb == 0 ? qf : qs)) * T + off;
#pragma unroll
                for (int c = 0; c < 16; ++c) u[rb][cb][c] = ap[c * DB];
            }
        if (tid < 2 * DB)
            xs[tid] = lu_wait_slot(slots + (size_t)(tid < DB ? qf : qs) * DB + row, B.abort_word, step == pos - 1);
        if (step == pos - 1) LU_STAMP(1);
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
            vv -= (red[rb][0][0][row] + red[rb][0][1][row]) + (red[rb][0][2][row] + red[rb][0][3][row]);
            vv -= (red[rb][1][0][row] + red[rb][1][1][row]) + (red[rb][1][2][row] + red[rb][1][3][row]);
        }
    }
    LU_STAMP(2);
    if (part == 0) {
        const int g = first * DB + row;
        const double val = diag_chain<LOWER>(lrow, dinv, vv);
        xs[row] = val;
        lu_publish_slot(slots + g, val);
        if (!LOWER && g < n) B.x[sys][g] = val;
        LU_STAMP(3);
    }
    __syncthreads();
    {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) s = fma(cpl[c], xs[part * 16 + c], s);
        redc[part][row] = s;
    }
    __syncthreads();
    if (part == 1) {
        const int g = second * DB + row;
        double val = vv - ((redc[0][row] + redc[1][row]) + (redc[2][row] + redc[3][row]));
        LU_STAMP(4);
        val = diag_chain<LOWER>(lrow, dinv, val);
        lu_publish_slot(slots + g, val);
        if (!LOWER && g < n) B.x[sys][g] = val;
        LU_STAMP(5);
    }
}

__global__ void permute_kernel(const double* __restrict__ b, const int* __restrict__ perm, int n,
                               double* __restrict__ v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = b[perm[i]];
}

}  // namespace

namespace {

// y = A x (+ y), A row-major (m, n): the QFS boundary limits S sigma + D tau and the refinement
// residual b - A x (reference: numpy products in the third-party qfs package).  HBM bound — the
// matrix once, 134 MB at n = 4096 — one wave per row, 16-byte loads, eight in flight per lane;
// the sum of a row: lane-strided partial sums, then a fixed shuffle tree (deterministic).
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double* __restrict__ A, int64_t m, int64_t n,
                                                        const double* __restrict__ x, double* __restrict__ y,
                                                        int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const double* a = A + row * n;
    double s0 = 0.0, s1 = 0.0;
    const int64_t n2 = n >> 1;
    if ((n & 1) == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)x & 15) == 0) {
        const double2* a2 = (const double2*)a;
        const double2* x2 = (const double2*)x;
#pragma unroll 8
        for (int64_t k = lane; k < n2; k += 64) {
            const double2 av = a2[k], xv = x2[k];
            s0 = fma(av.x, xv.x, s0);
            s1 = fma(av.y, xv.y, s1);
        }
    } else {
#pragma unroll 4
        for (int64_t k = lane; k < n; k += 64) s0 = fma(a[k], x[k], s0);
    }
    double s = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = accumulate ? y[row] + s : s;
}

// r = b - A x with the products and the sum carried in double-double (TwoProd by fma, TwoSum): the residual of an
// iterative-refinement step on the QFS systems of condition ~1e15, where the plain sum's own rounding
// (eps |A| |x|, densities of 10^3 .. 10^4) is the size of the residual it is asked for.  One wave per row as
// gemv_rows_kernel; HBM bound all the same (the matrix once), ~10 flops per element instead of 2.
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
__global__ __launch_bounds__(256) void residual_dd_kernel(const double* __restrict__ A, int64_t m, int64_t n,
                                                          const double* __restrict__ x,
                                                          const double* __restrict__ b, double* __restrict__ r) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const double* a = A + row * n;
    double hi = 0.0, lo = 0.0;
#pragma unroll 4
    for (int64_t k = lane; k < n; k += 64) {
        const double av = a[k], xv = x[k];
        const double p = av * xv;
        const double pe = fma(av, xv, -p);
        double s, e;
        two_sum(hi, p, s, e);
        hi = s;
        lo += e + pe;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double oh = __shfl_xor(hi, o), ol = __shfl_xor(lo, o);
        double s, e;
        two_sum(hi, oh, s, e);
        hi = s;
        lo += ol + e;
    }
    if (lane == 0) {
        double s, e;
        two_sum(b[row], -hi, s, e);
        r[row] = s + (e - lo);
    }
}

}  // namespace

extern "C" int ipde_dense_residual(ipde_ctx* ctx, int64_t m, int64_t n, const double* A, const double* x,
                                   const double* b, double* r) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, m >= 0 && n >= 0 && m < (1ll << 31));
    if (m == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, A && x && b && r);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(residual_dd_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, A, m, n, x, b, r);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_dense_gemv(ipde_ctx* ctx, int64_t m, int64_t n, const double* A, const double* x, double* y,
                               int accumulate) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, m >= 0 && n >= 0 && m < (1ll << 31) && (accumulate == 0 || accumulate == 1));
    if (m == 0) return IPDE_OK;
    IPDE_CHECK_ARG(ctx, A && x && y);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, A, m, n, x, y,
                       accumulate);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

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
    if (ctx->opt_dense_persistent) {
        // the abort word lives in device memory of its own (never refilled) and is copied to pinned
        // host memory behind every call's kernels: a waiter of an earlier call gave up (its results
        // are garbage) -> say so now
        volatile unsigned* seen = (volatile unsigned*)(ctx->h_pinned + ctx->h_pinned_bytes / sizeof(double) - 1);
        if (!ctx->d_lu_abort) {
            IPDE_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_lu_abort, 16));
            IPDE_HIP_CHECK(ctx, hipMemset(ctx->d_lu_abort, 0, 16));
            *seen = 0;
        }
        if (*seen != 0) {
            IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            IPDE_HIP_CHECK(ctx, hipMemset(ctx->d_lu_abort, 0, 16));
            *seen = 0;
            IPDE_SET_ERR(ctx, "ipde_dense_lu_solve_batch: a substitution workgroup of an earlier call timed out "
                              "waiting for its predecessors");
            return IPDE_ERR_HIP;
        }
        // [ticket fw, ticket bw, pad | forward slots | backward slots], one fill for all
        size_t slots = 0;
        for (int s = 0; s < nsys; ++s) slots += (size_t)((n[s] + 2 * DB - 1) / (2 * DB)) * 2 * DB;
        const size_t bytes = 16 + 2 * slots * sizeof(double);
        IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->partial, bytes));
        IPDE_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->partial.p, (int)LU_SENTINEL32, bytes / 4, ctx->stream));
        unsigned* head = (unsigned*)ctx->partial.p;
        double* base = (double*)((char*)ctx->partial.p + 16);
        LuPersist fw{}, bw{};
        fw.nsys = bw.nsys = nsys;
        fw.nbp_max = bw.nbp_max = nbp_max;
        fw.ticket = head;
        bw.ticket = head + 1;
        fw.abort_word = bw.abort_word = ctx->d_lu_abort;
        for (int s = 0; s < nsys; ++s) {
            const size_t ns = (size_t)((n[s] + 2 * DB - 1) / (2 * DB)) * 2 * DB;
            fw.lu[s] = bw.lu[s] = lu[s];
            fw.n[s] = bw.n[s] = (int)n[s];
            fw.perm[s] = perm[s];
            fw.b[s] = b[s];
            fw.slots[s] = base;
            bw.b[s] = base;
            bw.slots[s] = base + slots;
            bw.x[s] = x[s];
            base += ns;
        }
        const unsigned grid = (unsigned)(nsys * nbp_max);
        hipLaunchKernelGGL(lu_subst_persistent<true>, dim3(grid), dim3(DT), 0, ctx->stream, fw);
        hipLaunchKernelGGL(lu_subst_persistent<false>, dim3(grid), dim3(DT), 0, ctx->stream, bw);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync((void*)seen, ctx->d_lu_abort, sizeof(unsigned), hipMemcpyDeviceToHost,
                                           ctx->stream));
        return IPDE_OK;
    }
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
