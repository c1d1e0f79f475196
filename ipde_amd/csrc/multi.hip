// One host process driving several GPUs: the multi-device context of SURVEY §8(b)
// ("ipde_ctx_create(ndev, dev_ids, &ctx): owns streams, rocFFT plans, RCCL comm").
//
// A resident target set is split into ndev contiguous slices (targets are independent units,
// §8e), one per device.  An apply uploads the source arrays once to the first device, broadcasts
// them to the others with ncclBroadcast over xGMI (one grouped call), launches the per-device
// sums on the devices' own streams — they run side by side — and copies each slice of the result
// back into the caller's array.  No collective touches the targets or the results.
//
// RCCL is loaded at run time (dlopen) the first time a communicator is asked for: a one-device
// context, and every single-GPU user of the library, never loads it.  IPDE_MULTI_FORCE_COMM in
// `flags` builds the communicator (and routes the sources through ncclBroadcast) also for one
// device — the path a one-GPU box can exercise.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "ipde_common.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

const char* load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.lib) return nullptr;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return "librccl.so.1 cannot be loaded";
    Rccl r;
    r.lib = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.Broadcast || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
        return "librccl.so.1 lacks an expected symbol";
    g_rccl = r;
    return nullptr;
}

}  // namespace

struct ipde_multi {
    int ndev = 0;
    std::vector<int> dev;
    std::vector<ipde_ctx*> ctx;
    std::vector<ncclComm_t> comm;       // empty: no communicator (one device, not forced)
    std::vector<double*> d_src;         // per device: the broadcast source rows, NROW x ns_cap doubles
    int64_t ns_cap = 0;
    // resident targets
    int64_t nt = 0;
    std::vector<int64_t> t0;            // slice starts (ndev + 1)
    std::vector<double*> d_tx, d_ty, d_out;   // per device; d_out: 3 result rows of the slice
    double* h_src = nullptr;            // pinned staging of the source rows
    std::string err;
};

#define M_SET_ERR(m, ...)                          \
    do {                                           \
        char _b[512];                              \
        snprintf(_b, sizeof(_b), __VA_ARGS__);     \
        (m)->err = _b;                             \
    } while (0)
#define M_HIP(m, call)                                                                         \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) {                                                                \
            M_SET_ERR(m, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
            return IPDE_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)
#define M_NCCL(m, call)                                                                                 \
    do {                                                                                                \
        ncclResult_t _e = (call);                                                                       \
        if (_e != ncclSuccess) {                                                                        \
            M_SET_ERR(m, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(_e));      \
            return IPDE_ERR_HIP;                                                                        \
        }                                                                                               \
    } while (0)

// Every entry point below walks the devices with hipSetDevice: the caller's current device is put back on
// the way out (a Python caller's torch.cuda.current_device() / get_context() must not move to another GPU).
struct DeviceGuard {
    int d = -1;
    DeviceGuard() {
        if (hipGetDevice(&d) != hipSuccess) d = -1;
    }
    ~DeviceGuard() {
        if (d >= 0) hipSetDevice(d);
    }
};

// After a failure part-way through an apply: kernels and copies already queued on other devices still read the
// staging buffers and write the caller's arrays — wait for all of them before the error is handed back.
static void drain(ipde_multi* m) {
    for (int i = 0; i < m->ndev; ++i) {
        if (hipSetDevice(m->dev[i]) != hipSuccess || !m->ctx[i]) continue;
        hipStreamSynchronize(m->ctx[i]->stream);
    }
}

template <typename F>
static int guarded_apply(ipde_multi* m, F body) {
    DeviceGuard g;
    const int st = body();
    if (st != IPDE_OK) drain(m);
    return st;
}

static constexpr int NROW = 8;   // source rows an apply can carry: x, y and up to six density / normal rows

static void free_targets(ipde_multi* m) {
    for (int i = 0; i < m->ndev; ++i) {
        hipSetDevice(m->dev[i]);
        for (auto* v : {&m->d_tx, &m->d_ty, &m->d_out})
            if ((int)v->size() > i && (*v)[i]) {
                hipFree((*v)[i]);
                (*v)[i] = nullptr;
            }
    }
    m->nt = 0;
}

extern "C" int ipde_multi_destroy(ipde_multi* m) {
    if (!m) return IPDE_ERR_INVALID;
    DeviceGuard g;
    for (ipde_ctx* c : m->ctx)
        if (c) ipde_ctx_sync(c);
    free_targets(m);
    for (int i = 0; i < m->ndev; ++i) {
        hipSetDevice(m->dev[i]);
        if ((int)m->d_src.size() > i && m->d_src[i]) hipFree(m->d_src[i]);
        if ((int)m->comm.size() > i && m->comm[i]) g_rccl.CommDestroy(m->comm[i]);
        if ((int)m->ctx.size() > i && m->ctx[i]) ipde_ctx_destroy(m->ctx[i]);
    }
    if (m->h_src) hipHostFree(m->h_src);
    delete m;
    return IPDE_OK;
}

extern "C" int ipde_multi_create(int ndev, const int* dev_ids, int flags, ipde_multi** out) {
    if (!out || ndev < 1 || ndev > 64 || !dev_ids) return IPDE_ERR_INVALID;
    *out = nullptr;
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (dev_ids[i] == dev_ids[j]) return IPDE_ERR_INVALID;   // one context per physical device
    DeviceGuard g;
    ipde_multi* m = new ipde_multi();
    m->ndev = ndev;
    m->dev.assign(dev_ids, dev_ids + ndev);
    m->ctx.assign(ndev, nullptr);
    m->d_src.assign(ndev, nullptr);
    m->d_tx.assign(ndev, nullptr);
    m->d_ty.assign(ndev, nullptr);
    m->d_out.assign(ndev, nullptr);
    int st = IPDE_OK;
    for (int i = 0; i < ndev && st == IPDE_OK; ++i) st = ipde_ctx_create(dev_ids[i], &m->ctx[i]);
    if (st == IPDE_OK && (ndev > 1 || (flags & IPDE_MULTI_FORCE_COMM))) {
        const char* e = load_rccl();
        if (e) {
            fprintf(stderr, "ipde_hip: %s\n", e);
            st = IPDE_ERR_HIP;
        } else {
            m->comm.assign(ndev, nullptr);
            if (g_rccl.CommInitAll(m->comm.data(), ndev, dev_ids) != ncclSuccess) {
                fprintf(stderr, "ipde_hip: ncclCommInitAll over %d device(s) failed\n", ndev);
                m->comm.clear();
                st = IPDE_ERR_HIP;
            }
        }
    }
    if (st != IPDE_OK) {
        ipde_multi_destroy(m);
        return st;
    }
    *out = m;
    return IPDE_OK;
}

extern "C" int ipde_multi_ndev(ipde_multi* m, int* ndev) {
    if (!m || !ndev) return IPDE_ERR_INVALID;
    *ndev = m->ndev;
    return IPDE_OK;
}

extern "C" int ipde_multi_ctx(ipde_multi* m, int i, ipde_ctx** ctx) {
    if (!m || !ctx || i < 0 || i >= m->ndev) return IPDE_ERR_INVALID;
    *ctx = m->ctx[i];
    return IPDE_OK;
}

extern "C" int ipde_multi_has_comm(ipde_multi* m, int* yes) {
    if (!m || !yes) return IPDE_ERR_INVALID;
    *yes = m->comm.empty() ? 0 : 1;
    return IPDE_OK;
}

extern "C" const char* ipde_multi_last_error(ipde_multi* m) { return m ? m->err.c_str() : "null multi-device context"; }

extern "C" int ipde_multi_target_slice(ipde_multi* m, int i, int64_t* start, int64_t* stop) {
    if (!m || i < 0 || i >= m->ndev || !start || !stop || m->t0.empty()) return IPDE_ERR_INVALID;
    *start = m->t0[i];
    *stop = m->t0[i + 1];
    return IPDE_OK;
}

// contiguous, balanced slices (sizes differ by at most one): the partition of ipde_amd/sharding.py
extern "C" int ipde_multi_set_targets(ipde_multi* m, int64_t nt, const double* tx, const double* ty) {
    if (!m || nt < 0 || (nt > 0 && (!tx || !ty))) return IPDE_ERR_INVALID;
    DeviceGuard g;
    free_targets(m);
    m->t0.assign(m->ndev + 1, 0);
    const int64_t base = nt / m->ndev, rem = nt % m->ndev;
    for (int i = 0; i < m->ndev; ++i) m->t0[i + 1] = m->t0[i] + base + (i < rem ? 1 : 0);
    for (int i = 0; i < m->ndev; ++i) {
        const int64_t n = m->t0[i + 1] - m->t0[i];
        if (n == 0) continue;
        M_HIP(m, hipSetDevice(m->dev[i]));
        M_HIP(m, hipMalloc((void**)&m->d_tx[i], n * sizeof(double)));
        M_HIP(m, hipMalloc((void**)&m->d_ty[i], n * sizeof(double)));
        M_HIP(m, hipMalloc((void**)&m->d_out[i], 3 * n * sizeof(double)));
        M_HIP(m, hipMemcpy(m->d_tx[i], tx + m->t0[i], n * sizeof(double), hipMemcpyHostToDevice));
        M_HIP(m, hipMemcpy(m->d_ty[i], ty + m->t0[i], n * sizeof(double), hipMemcpyHostToDevice));
    }
    m->nt = nt;
    return IPDE_OK;
}

// rows[r] (host, ns doubles, may be NULL) -> row r of every device's source buffer
static int distribute_sources(ipde_multi* m, int64_t ns, const double* const* rows, int nrow) {
    if (ns > m->ns_cap) {
        const int64_t cap = ns + ns / 4 + 64;
        for (int i = 0; i < m->ndev; ++i) {
            M_HIP(m, hipSetDevice(m->dev[i]));
            if (m->d_src[i]) M_HIP(m, hipFree(m->d_src[i]));
            m->d_src[i] = nullptr;
            M_HIP(m, hipMalloc((void**)&m->d_src[i], (size_t)NROW * cap * sizeof(double)));
        }
        if (m->h_src) M_HIP(m, hipHostFree(m->h_src));
        m->h_src = nullptr;
        M_HIP(m, hipHostMalloc((void**)&m->h_src, (size_t)NROW * cap * sizeof(double)));
        m->ns_cap = cap;
    }
    // (the previous apply's copies out of the pinned staging area have completed: every apply ends
    // with a synchronisation of all devices)
    for (int r = 0; r < nrow; ++r)
        if (rows[r]) memcpy(m->h_src + (size_t)r * m->ns_cap, rows[r], ns * sizeof(double));
    const size_t count = (size_t)nrow * m->ns_cap;
    if (m->comm.empty()) {
        for (int i = 0; i < m->ndev; ++i) {
            M_HIP(m, hipSetDevice(m->dev[i]));
            M_HIP(m, hipMemcpyAsync(m->d_src[i], m->h_src, count * sizeof(double), hipMemcpyHostToDevice,
                                    m->ctx[i]->stream));
        }
        return IPDE_OK;
    }
    // up once to the first device, then to the others over xGMI
    M_HIP(m, hipSetDevice(m->dev[0]));
    M_HIP(m, hipMemcpyAsync(m->d_src[0], m->h_src, count * sizeof(double), hipMemcpyHostToDevice,
                            m->ctx[0]->stream));
    M_NCCL(m, g_rccl.GroupStart());
    for (int i = 0; i < m->ndev; ++i) {
        ncclResult_t e = g_rccl.Broadcast(m->d_src[i], m->d_src[i], count, ncclDouble, 0, m->comm[i],
                                          m->ctx[i]->stream);
        if (e != ncclSuccess) {
            g_rccl.GroupEnd();
            M_SET_ERR(m, "ncclBroadcast (device %d): %s", m->dev[i], g_rccl.GetErrorString(e));
            return IPDE_ERR_HIP;
        }
    }
    M_NCCL(m, g_rccl.GroupEnd());
    return IPDE_OK;
}

// per-device sums of one apply: fn(i, ctx, source rows on the device, nt_i, tx, ty, out rows)
template <typename F>
static int run_on_slices(ipde_multi* m, int nout, double* const* outs, F fn) {
    for (int i = 0; i < m->ndev; ++i) {
        const int64_t n = m->t0[i + 1] - m->t0[i];
        if (n == 0) continue;
        M_HIP(m, hipSetDevice(m->dev[i]));
        const int st = fn(i, m->ctx[i], m->d_src[i], n, m->d_tx[i], m->d_ty[i], m->d_out[i]);
        if (st != IPDE_OK) {
            M_SET_ERR(m, "device %d: %s", m->dev[i], ipde_last_error(m->ctx[i]));
            return st;
        }
    }
    // (a copy into pageable host memory blocks the host until it is done: all sums are in their
    // streams before the first one is waited for)
    for (int i = 0; i < m->ndev; ++i) {
        const int64_t n = m->t0[i + 1] - m->t0[i];
        if (n == 0) continue;
        M_HIP(m, hipSetDevice(m->dev[i]));
        for (int c = 0; c < nout; ++c)
            if (outs[c])
                M_HIP(m, hipMemcpyAsync(outs[c] + m->t0[i], m->d_out[i] + (size_t)c * n, n * sizeof(double),
                                        hipMemcpyDeviceToHost, m->ctx[i]->stream));
    }
    for (int i = 0; i < m->ndev; ++i) {
        M_HIP(m, hipSetDevice(m->dev[i]));
        M_HIP(m, hipStreamSynchronize(m->ctx[i]->stream));
    }
    return IPDE_OK;
}

static int check_apply(ipde_multi* m, int64_t ns) {
    if (!m) return IPDE_ERR_INVALID;
    if (m->t0.empty()) {
        M_SET_ERR(m, "ipde_multi_set_targets has not been called");
        return IPDE_ERR_INVALID;
    }
    if (ns < 1 || ns >= (1LL << 30)) {
        M_SET_ERR(m, "ns out of range");
        return IPDE_ERR_INVALID;
    }
    return IPDE_OK;
}

extern "C" int ipde_multi_laplace_apply(ipde_multi* m, int64_t ns, const double* sx, const double* sy,
                                        const double* w_sigma, const double* nx, const double* ny,
                                        const double* w_tau, double* out, int flags) {
    IPDE_TRY(check_apply(m, ns));
    if (!sx || !sy || !out || (!w_sigma && !w_tau) || (w_tau && (!nx || !ny))) return IPDE_ERR_INVALID;
    if (m->nt == 0) return IPDE_OK;
    const double* rows[6] = {sx, sy, w_sigma, nx, ny, w_tau};
    return guarded_apply(m, [&]() {
    IPDE_TRY(distribute_sources(m, ns, rows, 6));
    const int64_t cap = m->ns_cap;
    double* outs[1] = {out};
    return run_on_slices(m, 1, outs, [&](int, ipde_ctx* c, const double* s, int64_t n, const double* tx,
                                         const double* ty, double* o) {
        return ipde_laplace_apply(c, IPDE_DEVICE, ns, s, s + cap, w_sigma ? s + 2 * cap : nullptr,
                                  w_tau ? s + 3 * cap : nullptr, w_tau ? s + 4 * cap : nullptr,
                                  w_tau ? s + 5 * cap : nullptr, n, tx, ty, o, flags);
    });
    });
}

extern "C" int ipde_multi_modhelm_apply(ipde_multi* m, double k, int64_t ns, const double* sx,
                                        const double* sy, const double* w_sigma, const double* nx,
                                        const double* ny, const double* w_tau, double* out, int flags) {
    IPDE_TRY(check_apply(m, ns));
    if (!sx || !sy || !out || (!w_sigma && !w_tau) || (w_tau && (!nx || !ny))) return IPDE_ERR_INVALID;
    if (m->nt == 0) return IPDE_OK;
    const double* rows[6] = {sx, sy, w_sigma, nx, ny, w_tau};
    return guarded_apply(m, [&]() {
    IPDE_TRY(distribute_sources(m, ns, rows, 6));
    const int64_t cap = m->ns_cap;
    double* outs[1] = {out};
    return run_on_slices(m, 1, outs, [&](int, ipde_ctx* c, const double* s, int64_t n, const double* tx,
                                         const double* ty, double* o) {
        return ipde_modhelm_apply(c, IPDE_DEVICE, k, ns, s, s + cap, w_sigma ? s + 2 * cap : nullptr,
                                  w_tau ? s + 3 * cap : nullptr, w_tau ? s + 4 * cap : nullptr,
                                  w_tau ? s + 5 * cap : nullptr, n, tx, ty, o, flags);
    });
    });
}

extern "C" int ipde_multi_stokes_apply(ipde_multi* m, int64_t ns, const double* sx, const double* sy,
                                       const double* wfx, const double* wfy, const double* nx,
                                       const double* ny, const double* wdx, const double* wdy,
                                       double* out_u, double* out_v, double* out_p, int flags) {
    IPDE_TRY(check_apply(m, ns));
    const bool slp = wfx && wfy, dlp = wdx && wdy;
    if (!sx || !sy || !out_u || !out_v || (!slp && !dlp) || (dlp && (!nx || !ny))) return IPDE_ERR_INVALID;
    if (m->nt == 0) return IPDE_OK;
    const double* rows[8] = {sx, sy, slp ? wfx : nullptr, slp ? wfy : nullptr, dlp ? nx : nullptr,
                             dlp ? ny : nullptr, dlp ? wdx : nullptr, dlp ? wdy : nullptr};
    return guarded_apply(m, [&]() {
    IPDE_TRY(distribute_sources(m, ns, rows, 8));
    const int64_t cap = m->ns_cap;
    double* outs[3] = {out_u, out_v, out_p};
    return run_on_slices(m, 3, outs, [&](int, ipde_ctx* c, const double* s, int64_t n, const double* tx,
                                         const double* ty, double* o) {
        return ipde_stokes_apply(c, IPDE_DEVICE, ns, s, s + cap, slp ? s + 2 * cap : nullptr,
                                 slp ? s + 3 * cap : nullptr, dlp ? s + 4 * cap : nullptr,
                                 dlp ? s + 5 * cap : nullptr, dlp ? s + 6 * cap : nullptr,
                                 dlp ? s + 7 * cap : nullptr, n, tx, ty, o, o + n, out_p ? o + 2 * n : nullptr,
                                 flags);
    });
    });
}
