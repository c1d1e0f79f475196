// Internal shared declarations for libipde_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include "../../include/ipde_hip.h"

// A growable device buffer owned by the context (never freed between calls so
// that the hot path does no hipMalloc).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// LDS log/reciprocal table, see mathtab.h.
struct LogTable {
    double* d_tab = nullptr;  // nkeys * 2 doubles: {R, -log(R)}
    int mant_bits = 0;
    int key_lo = 0;           // first key = (hi32(x) >> (20-mant_bits)) covered
    int nkeys = 0;
    int exp_lo = 0, exp_hi = 0;  // covers x in [2^exp_lo, 2^exp_hi)
};

struct ipde_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // workspaces
    DevBuf src_pack;     // packed source records
    DevBuf stage[16];    // host-staging buffers for IPDE_HOST calls
    DevBuf partial;      // split-source partial sums
    DevBuf scratch;      // reductions (bbox etc.)
    DevBuf fftwork[8];
    DevBuf r2g[4];       // ipde_radial_to_grid: coefficient rows, spectra, oversampled rows
    DevBuf lu_work;      // ipde_dense_lu_factor: row-major U tiles of a block row, row moves of a panel
    DevBuf cut_work;     // ipde_density_noise_cut: the density as a complex signal and its spectrum
    std::map<int, double*> cheb_tab;   // M -> device M x M analysis matrix of the Chebyshev-Gauss nodes
    double* h_pinned = nullptr;  // small pinned host buffer for scalars
    size_t h_pinned_bytes = 0;
    LogTable logtab;
    // modified-Helmholtz K0/K1 piecewise table
    double* d_ktab = nullptr;
    // timing
    int timing = 0;
    // ring of event pairs around the dominant kernel of every layer-potential apply since timing
    // was switched on: a timed loop records without a host sync and resolves the ring afterwards
    // (ipde_ctx_kernel_ms_history); pairs are created at first use
    static constexpr int TIMING_RING = 256;
    hipEvent_t ring_ev0[TIMING_RING] = {}, ring_ev1[TIMING_RING] = {};
    int64_t ring_n = 0;          // pairs recorded since ipde_ctx_enable_timing(1)
    int ring_open = -1;          // slot whose start event is recorded and whose stop is not
    double last_kernel_ms = 0.0;
    // 1-D batched fft plan cache: key (batch, n)
    std::map<std::pair<int64_t, int64_t>, void*> fft1_plans;
    std::mutex fft1_mutex;   // plan creation may run in a host warm-up thread (ipde_fft1_prepare)
    int num_cu = 256;
    // tuning knobs (ipde_ctx_set_option)
    int opt_laplace_variant = 9;   // row-run single layer (variant 1 for the other modes)
    int opt_annular_grouped = 2;  // Stokes annular operator: 0 = one launch per term, 1 = grouped launches, 2 = grouped, copies and closing launches merged
    int opt_annular_fused_fft = 1; // scalar annular operator: transform pairs as one kernel for power-of-two n <= 4096 (0: rocFFT + pointwise)
    int opt_gmres_lookahead = 1;  // annular GMRES: inner iteration j + 1 enters the stream before the host has read column j (0: enqueue, wait, enqueue)
    int opt_gmres_fused_scale = 1; // annular GMRES: v_j = w / ||w|| formed inside the preconditioner's kernel (0: a launch of its own)
    int opt_gmres_persistent = 0; // scalar annular GMRES: 1 = the first cycle in ONE launch, Arnoldi bookkeeping on the device (annular_gmres_persist.h). Measured 8-14 % SLOWER than the launch-per-stage cycle with its look-ahead (profiles/r03_gmres_persistent_ab.txt): off
    int opt_gmres_graphs = 0;     // annular GMRES: 1 = inner iterations replayed from hipGraphs (no gain measured: profiles/r02_gmres_graph_ab.txt)
    int opt_dense_pairs = 1;      // substitution: two 64-row blocks per launch (0: one)
    int opt_dense_persistent = 1; // substitution: one launch per triangular pass, in-launch hand-off (0: a launch per step)
    unsigned* d_lu_abort = nullptr;   // sticky time-out word of the persistent substitution (last 8 bytes of h_pinned mirror it)
    int opt_modhelm_variant = 0;  // table kernel geometry: 0: 2 targets per lane, 1: 4, 2: 2 x two sources in flight, 3: 3
    int opt_stokes_variant = 1;   // 1: row-run stokeslet kernel (single layer), 0: strided table kernel
    int opt_interp_shifted = 0;   // 1: ipde_grid_interp always through four shifted coarse transforms (testing)
    int opt_timing_split = 0;     // 1: ipde_laplace_apply_patches_far records an event pair per stage (parents, blocks, patches)
    int opt_interp_band = 1;      // 1: ipde_grid_interp in its band form (FFT along x only, exact sums along y: nufft.hip), 0: full fine grids
    int opt_fft2d = 1;            // 1: hand-written 2-D FFT pipeline on power-of-two grids (fft2d.hip), 0: rocFFT
};

#define IPDE_SET_ERR(ctx, ...)                                      \
    do {                                                            \
        char _b[512];                                               \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                      \
        if (ctx) (ctx)->err = _b;                                   \
    } while (0)

#define IPDE_HIP_CHECK(ctx, call)                                                      \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            IPDE_SET_ERR(ctx, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,            \
                         hipGetErrorString(_e));                                       \
            return IPDE_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)

#define IPDE_CHECK_ARG(ctx, cond)                                                      \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            IPDE_SET_ERR(ctx, "%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return IPDE_ERR_INVALID;                                                   \
        }                                                                              \
    } while (0)

#define IPDE_TRY(expr)                  \
    do {                                \
        int _s = (expr);                \
        if (_s != IPDE_OK) return _s;   \
    } while (0)

// ensure capacity (grow-only). Returns IPDE_OK / IPDE_ERR_ALLOC.
int ipde_devbuf_reserve(ipde_ctx* ctx, DevBuf& b, size_t bytes);

// Resolve an input array to a device pointer: for IPDE_DEVICE returns the
// pointer itself, for IPDE_HOST copies into ctx->stage[slot].
int ipde_stage_in(ipde_ctx* ctx, int loc, int slot, const double* p, size_t n_doubles,
                  const double** dptr);
// Resolve an output array: device pointer to write into.
int ipde_stage_out(ipde_ctx* ctx, int loc, int slot, double* p, size_t n_doubles,
                   double** dptr);
// After the kernels: copy staged outputs back (no-op for IPDE_DEVICE).
int ipde_stage_finish(ipde_ctx* ctx, int loc, int slot, double* p, size_t n_doubles);

int ipde_build_log_table(ipde_ctx* ctx);
int ipde_build_k_table(ipde_ctx* ctx);

// Bracket the dominant kernel of an apply (no-ops when timing is off).
static inline void ipde_time_begin(ipde_ctx* ctx) {
    if (!ctx->timing) return;
    const int slot = (int)(ctx->ring_n % ipde_ctx::TIMING_RING);
    if (!ctx->ring_ev0[slot] &&
        (hipEventCreate(&ctx->ring_ev0[slot]) != hipSuccess || hipEventCreate(&ctx->ring_ev1[slot]) != hipSuccess))
        return;
    if (hipEventRecord(ctx->ring_ev0[slot], ctx->stream) == hipSuccess) ctx->ring_open = slot;
}
// The next kernel belongs to the pair recorded last (an accumulating second launch of one apply):
// its stop event is recorded again behind it.
static inline void ipde_time_continue(ipde_ctx* ctx) {
    if (!ctx->timing || ctx->ring_n == 0) return;
    ctx->ring_n -= 1;
    ctx->ring_open = (int)(ctx->ring_n % ipde_ctx::TIMING_RING);
}
static inline void ipde_time_end(ipde_ctx* ctx) {
    if (!ctx->timing || ctx->ring_open < 0) return;
    hipEventRecord(ctx->ring_ev1[ctx->ring_open], ctx->stream);
    ctx->ring_open = -1;
    ctx->ring_n += 1;
    ctx->last_kernel_ms = -1.0;  // resolved lazily in ipde_ctx_last_kernel_ms
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
