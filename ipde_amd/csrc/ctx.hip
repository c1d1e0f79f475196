// Context, workspaces, host staging and math-table construction.
#include "ipde_common.h"
#include <cmath>

extern "C" const char* ipde_version(void) { return "ipde_hip 0.1 (gfx950)"; }

int ipde_devbuf_reserve(ipde_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return IPDE_OK;
    if (b.p) {
        // the buffer may still be in use by queued work
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        IPDE_HIP_CHECK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&b.p, cap);
    if (e != hipSuccess) {
        IPDE_SET_ERR(ctx, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        b.p = nullptr;
        return IPDE_ERR_ALLOC;
    }
    b.cap = cap;
    return IPDE_OK;
}

int ipde_stage_in(ipde_ctx* ctx, int loc, int slot, const double* p, size_t n,
                  const double** dptr) {
    if (p == nullptr) {
        *dptr = nullptr;
        return IPDE_OK;
    }
    if (loc == IPDE_DEVICE) {
        *dptr = p;
        return IPDE_OK;
    }
    DevBuf& b = ctx->stage[slot];
    IPDE_TRY(ipde_devbuf_reserve(ctx, b, n * sizeof(double)));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(b.p, p, n * sizeof(double), hipMemcpyHostToDevice,
                                       ctx->stream));
    *dptr = (const double*)b.p;
    return IPDE_OK;
}

int ipde_stage_out(ipde_ctx* ctx, int loc, int slot, double* p, size_t n, double** dptr) {
    if (p == nullptr) {
        *dptr = nullptr;
        return IPDE_OK;
    }
    if (loc == IPDE_DEVICE) {
        *dptr = p;
        return IPDE_OK;
    }
    DevBuf& b = ctx->stage[slot];
    IPDE_TRY(ipde_devbuf_reserve(ctx, b, n * sizeof(double)));
    *dptr = (double*)b.p;
    return IPDE_OK;
}

int ipde_stage_finish(ipde_ctx* ctx, int loc, int slot, double* p, size_t n) {
    if (p == nullptr || loc == IPDE_DEVICE) return IPDE_OK;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(p, ctx->stage[slot].p, n * sizeof(double),
                                       hipMemcpyDeviceToHost, ctx->stream));
    IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return IPDE_OK;
}

// ---------------------------------------------------------------------------
// log table: for every key kappa = hi32(x) >> (20 - B) in the covered range the
// entry holds R ~ 1/center(interval) and T = -log(R), so that
//   log(x) = T + log1p(x*R - 1),  |x*R - 1| <= 2^-(B+1) (+ rounding).
// Built on the host in long double and uploaded once.
// One table per device and process, shared by every context of the device (read-only; never freed):
// a context of its own per solver thread / factorisation stream would otherwise each pay the
// synchronous upload — a blocking copy that waits for whatever the GPU is doing (30 ms apiece
// inside a set-up).
static std::mutex g_logtab_mutex;
static std::map<int, LogTable> g_logtab;

int ipde_build_log_table(ipde_ctx* ctx) {
    std::lock_guard<std::mutex> guard(g_logtab_mutex);
    auto it = g_logtab.find(ctx->device);
    if (it != g_logtab.end()) {
        ctx->logtab = it->second;
        return IPDE_OK;
    }
    LogTable& t = ctx->logtab;
    // 32 binades x 256 mantissa intervals = 8192 entries (128 KiB of LDS).  An entry
    // sits at position (key mod 8192): the kernels index with a shift and a mask
    // only, and validate the covered range once per lane from min/max of hi32(x).
    t.mant_bits = 8;
    t.exp_hi = 4;
    t.exp_lo = t.exp_hi - 32;
    const int B = t.mant_bits;
    t.key_lo = (1023 + t.exp_lo) << B;
    t.nkeys = (t.exp_hi - t.exp_lo) << B;
    std::vector<double> h((size_t)t.nkeys * 2);
    for (int n = 0; n < t.nkeys; ++n) {
        uint64_t key = (uint64_t)(t.key_lo + n);
        const int i = (int)(key & (uint64_t)(t.nkeys - 1));
        uint64_t lo_bits = key << (52 - B);
        uint64_t hi_bits = (key + 1) << (52 - B);
        double xlo, xhi;
        memcpy(&xlo, &lo_bits, 8);
        memcpy(&xhi, &hi_bits, 8);
        long double c = 0.5L * ((long double)xlo + (long double)xhi);
        double R = (double)(1.0L / c);
        double T = (double)(-logl((long double)R));
        h[2 * i] = 0.5 * R;  // Rh = R/2 (exact), see layer_common.h
        h[2 * i + 1] = T;
    }
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&t.d_tab, h.size() * sizeof(double)));
    IPDE_HIP_CHECK(ctx, hipMemcpy(t.d_tab, h.data(), h.size() * sizeof(double),
                                  hipMemcpyHostToDevice));
    g_logtab[ctx->device] = t;
    return IPDE_OK;
}

extern "C" int ipde_ctx_create(int device_id, ipde_ctx** out) {
    if (!out) return IPDE_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return IPDE_ERR_NOGPU;
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess) return IPDE_ERR_NOGPU;
    }
    if (device_id >= ndev) return IPDE_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return IPDE_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return IPDE_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "ipde_hip: device %d is %s, this library is built for gfx950 only\n",
                device_id, prop.gcnArchName);
        return IPDE_ERR_NOGPU;
    }
    ipde_ctx* ctx = new ipde_ctx();
    ctx->device = device_id;
    ctx->num_cu = prop.multiProcessorCount;
    // every failure below goes through ipde_ctx_destroy, which frees whatever exists
    // a BLOCKING stream: it orders itself against the legacy default stream, which is
    // where a host framework (torch) allocates and fills the buffers it hands us
    int s = IPDE_OK;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamDefault) != hipSuccess) s = IPDE_ERR_HIP;
    ctx->stream = ctx->own_stream;
    ctx->h_pinned_bytes = 1 << 16;
    if (s == IPDE_OK && hipHostMalloc((void**)&ctx->h_pinned, ctx->h_pinned_bytes) != hipSuccess) {
        ctx->h_pinned = nullptr;
        s = IPDE_ERR_ALLOC;
    }
    if (s == IPDE_OK) {
        s = ipde_build_log_table(ctx);
        // (the modified-Helmholtz K0/K1 table — 125 000 long-double Bessel evaluations, 0.1 s —
        // is built by the first ipde_modhelm_apply)
        if (s != IPDE_OK) fprintf(stderr, "ipde_hip: table construction failed: %s\n", ctx->err.c_str());
    }
    if (s != IPDE_OK) {
        ipde_ctx_destroy(ctx);
        return s;
    }
    *out = ctx;
    return IPDE_OK;
}

void ipde_fft1_plans_destroy(ipde_ctx* ctx);  // spectral.hip

extern "C" int ipde_ctx_destroy(ipde_ctx* ctx) {
    if (!ctx) return IPDE_ERR_INVALID;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    ipde_fft1_plans_destroy(ctx);
    auto freebuf = [](DevBuf& b) {
        if (b.p) hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    };
    freebuf(ctx->src_pack);
    for (auto& b : ctx->stage) freebuf(b);
    freebuf(ctx->partial);
    freebuf(ctx->scratch);
    for (auto& b : ctx->fftwork) freebuf(b);
    for (auto& b : ctx->r2g) freebuf(b);
    freebuf(ctx->lu_work);
    freebuf(ctx->cut_work);
    for (auto& kv : ctx->cheb_tab)
        if (kv.second) hipFree(kv.second);
    ctx->cheb_tab.clear();
    // (the log table belongs to the device, see ipde_build_log_table)
    // (d_ktab belongs to the device, not to the context: layer_modhelm.hip)
    if (ctx->d_lu_abort) hipFree(ctx->d_lu_abort);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    for (int i = 0; i < ipde_ctx::TIMING_RING; ++i) {
        if (ctx->ring_ev0[i]) hipEventDestroy(ctx->ring_ev0[i]);
        if (ctx->ring_ev1[i]) hipEventDestroy(ctx->ring_ev1[i]);
    }
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return IPDE_OK;
}

extern "C" int ipde_ctx_sync(ipde_ctx* ctx) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // the persistent substitution's sticky time-out word (dense.hip) has arrived by now
    if (ctx->d_lu_abort && ctx->h_pinned) {
        volatile unsigned* seen = (volatile unsigned*)(ctx->h_pinned + ctx->h_pinned_bytes / sizeof(double) - 1);
        if (*seen != 0) {
            IPDE_SET_ERR(ctx, "a substitution workgroup of ipde_dense_lu_solve_batch timed out waiting for its "
                              "predecessors: the results of that call are invalid");
            return IPDE_ERR_HIP;
        }
    }
    return IPDE_OK;
}

extern "C" int ipde_ctx_set_stream(ipde_ctx* ctx, void* s) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return IPDE_OK;
}

extern "C" int ipde_ctx_use_legacy_stream(ipde_ctx* ctx) { return ipde_ctx_set_stream(ctx, (void*)hipStreamLegacy); }

// The context's own stream replaced by a NON-blocking stream of the LOWEST priority the device offers: background
// work (the QFS factorisations of a set-up) then yields the CUs to whatever the other streams submit, instead of
// making a 50 us kernel and the host thread waiting for its result queue behind 0.5 s of trailing updates.
extern "C" int ipde_ctx_use_background_stream(ipde_ctx* ctx) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    int least = 0, greatest = 0;
    IPDE_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    IPDE_HIP_CHECK(ctx, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least));
    if (ctx->own_stream) {
        hipStreamSynchronize(ctx->own_stream);
        hipStreamDestroy(ctx->own_stream);
    }
    ctx->own_stream = s;
    ctx->stream = s;
    return IPDE_OK;
}

extern "C" void* ipde_ctx_get_stream(ipde_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" const char* ipde_last_error(ipde_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int* option_slot(ipde_ctx* ctx, const char* name) {
    if (!strcmp(name, "laplace_variant")) return &ctx->opt_laplace_variant;
    if (!strcmp(name, "stokes_variant")) return &ctx->opt_stokes_variant;
    if (!strcmp(name, "dense_pairs")) return &ctx->opt_dense_pairs;
    if (!strcmp(name, "dense_persistent")) return &ctx->opt_dense_persistent;
    if (!strcmp(name, "gmres_graphs")) return &ctx->opt_gmres_graphs;
    if (!strcmp(name, "gmres_persistent")) return &ctx->opt_gmres_persistent;
    if (!strcmp(name, "modhelm_variant")) return &ctx->opt_modhelm_variant;
    if (!strcmp(name, "gmres_lookahead")) return &ctx->opt_gmres_lookahead;
    if (!strcmp(name, "gmres_fused_scale")) return &ctx->opt_gmres_fused_scale;
    if (!strcmp(name, "annular_fused_fft")) return &ctx->opt_annular_fused_fft;
    if (!strcmp(name, "annular_grouped")) return &ctx->opt_annular_grouped;
    if (!strcmp(name, "fft2d")) return &ctx->opt_fft2d;
    if (!strcmp(name, "interp_shifted")) return &ctx->opt_interp_shifted;
    if (!strcmp(name, "interp_band")) return &ctx->opt_interp_band;
    if (!strcmp(name, "timing_split")) return &ctx->opt_timing_split;
    return nullptr;
}

extern "C" int ipde_ctx_set_option(ipde_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return IPDE_ERR_INVALID;
    int* slot = option_slot(ctx, name);
    if (!slot) {
        IPDE_SET_ERR(ctx, "unknown option '%s'", name);
        return IPDE_ERR_INVALID;
    }
    *slot = value;
    return IPDE_OK;
}

extern "C" int ipde_ctx_get_option(ipde_ctx* ctx, const char* name, int* value) {
    if (!ctx || !name || !value) return IPDE_ERR_INVALID;
    int* slot = option_slot(ctx, name);
    if (!slot) {
        IPDE_SET_ERR(ctx, "unknown option '%s'", name);
        return IPDE_ERR_INVALID;
    }
    *value = *slot;
    return IPDE_OK;
}

extern "C" int ipde_ctx_enable_timing(ipde_ctx* ctx, int on) {
    if (!ctx) return IPDE_ERR_INVALID;
    if (on && !ctx->timing) {   // a new measurement: forget the pairs of the previous one
        ctx->ring_n = 0;
        ctx->ring_open = -1;
        ctx->last_kernel_ms = 0.0;
    }
    ctx->timing = on;
    return IPDE_OK;
}

static int ring_slot_ms(ipde_ctx* ctx, int slot, double* ms) {
    IPDE_HIP_CHECK(ctx, hipEventSynchronize(ctx->ring_ev1[slot]));
    float f = 0.f;
    IPDE_HIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->ring_ev0[slot], ctx->ring_ev1[slot]));
    *ms = (double)f;
    return IPDE_OK;
}

extern "C" int ipde_ctx_last_kernel_ms(ipde_ctx* ctx, double* ms) {
    if (!ctx || !ms) return IPDE_ERR_INVALID;
    if (ctx->timing && ctx->last_kernel_ms < 0.0 && ctx->ring_n > 0)
        IPDE_TRY(ring_slot_ms(ctx, (int)((ctx->ring_n - 1) % ipde_ctx::TIMING_RING), &ctx->last_kernel_ms));
    *ms = ctx->last_kernel_ms;
    return IPDE_OK;
}

extern "C" int ipde_ctx_kernel_ms_history(ipde_ctx* ctx, double* ms, int cap, int* n) {
    if (!ctx || !ms || !n || cap < 0) return IPDE_ERR_INVALID;
    int64_t have = ctx->ring_n < ipde_ctx::TIMING_RING ? ctx->ring_n : ipde_ctx::TIMING_RING;
    if (have > cap) have = cap;
    for (int64_t i = 0; i < have; ++i)   // oldest of the kept ones first
        IPDE_TRY(ring_slot_ms(ctx, (int)((ctx->ring_n - have + i) % ipde_ctx::TIMING_RING), &ms[i]));
    *n = (int)have;
    return IPDE_OK;
}
