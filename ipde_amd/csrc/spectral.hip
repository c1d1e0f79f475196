// Periodic spectral grid operators: rocFFT real 2-D
// transforms + fused symbol kernels, 4th-order stencils, batched 1-D FFTs.
//
// Roofline: HBM bound.  Algorithmic bytes of one grid solve = read f + write u
// (16 B per grid point); the practical r2c -> symbol -> c2r pipeline moves about
// 4x that.  The reference does full complex fft2/ifft2(...).real; we use D2Z/Z2D
// and reproduce the `.real` projection exactly by Hermitian-symmetrising the
// effective symbol (matters only on the Nyquist row/column of odd symbols).
#include "ipde_common.h"
#include "fft2d.h"
#include <rocfft/rocfft.h>
#include <mutex>

#define IPDE_FFT_CHECK(ctx, call)                                                         \
    do {                                                                                  \
        rocfft_status _s = (call);                                                        \
        if (_s != rocfft_status_success) {                                                \
            IPDE_SET_ERR(ctx, "%s:%d: %s -> rocfft status %d", __FILE__, __LINE__, #call, \
                         (int)_s);                                                        \
            return IPDE_ERR_FFT;                                                          \
        }                                                                                 \
    } while (0)

// rocFFT plan creation (and its run-time kernel compilation) is entered by one thread at a
// time, whatever context or plan kind it is for: contexts of different boundaries are driven
// from different host threads (concurrent annular solves), a warm-up thread creates plans
// during set-up, and a two-rank rehearsal with two threads inside plan creation ended in a
// SIGSEGV (no backtrace of it was kept; serialising ALL plan creation is the mitigation).
static std::mutex g_rocfft_plan_mutex;
static std::once_flag g_rocfft_once;
static void rocfft_setup_once() {
    std::call_once(g_rocfft_once, []() { rocfft_setup(); });
}

struct ipde_fft_plan {
    ipde_ctx* ctx = nullptr;
    int64_t nx = 0, ny = 0, nyh = 0;
    double hx = 0, hy = 0;
    rocfft_plan r2c = nullptr, c2r = nullptr, c2c_f = nullptr, c2c_b = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;
    double2* spec[3] = {nullptr, nullptr, nullptr};
    Fft2dPlan fast;   // hand-written pipeline (power-of-two grids), fft2d.hip
    bool keep_spec = false, have_spec = false;   // fast.W[1] holds the last solve's spectrum
    GridInterp* interp = nullptr;                // created by the first ipde_grid_interp
    // the same for grids outside fft2d's sizes (rocFFT path): the half spectrum is copied aside
    bool have_spec_general = false;
    double2* kept_half = nullptr;                // (nx, nyh): fft2(f) * symbol / (nx ny)
    double2* gspec[3] = {nullptr, nullptr, nullptr};   // half spectra of ipde_grid_interp_fields' inputs
    GridInterp* interp_general = nullptr;
    double* rbuf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // host staging
    double2* cbuf[2] = {nullptr, nullptr};                            // host staging (complex)
};

namespace {

__device__ __forceinline__ double wavenumber(int64_t i, int64_t n, double dk) {
    // np.fft.fftfreq ordering: 0..(n-1)/2, -n/2..-1 ; times 2 pi / L
    int64_t s = (i < (n + 1) / 2) ? i : i - n;
    return (double)s * dk;
}
// wavenumber of the mirrored mode -k as the complex pipeline sees it: the Nyquist
// index (even n) maps onto itself
__device__ __forceinline__ double wavenumber_neg(int64_t i, int64_t n, double dk) {
    if ((n & 1) == 0 && i == n / 2) return wavenumber(i, n, dk);
    return -wavenumber(i, n, dk);
}

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cplx cconj(cplx a) { return cplx{a.re, -a.im}; }

enum { SYM_POISSON = 0, SYM_MODHELM = 1, SYM_DX = 2, SYM_DY = 3 };

// complex symbol S(kx, ky) of the scalar operators as the reference evaluates it
template <int SYM>
__device__ __forceinline__ cplx scalar_symbol(double kx, double ky, double k2h, bool is00) {
    if (SYM == SYM_POISSON) {
        if (is00) return cplx{0.0, 0.0};
        return cplx{1.0 / (-kx * kx - ky * ky), 0.0};
    } else if (SYM == SYM_MODHELM) {
        return cplx{1.0 / (k2h - (-kx * kx - ky * ky)), 0.0};
    } else if (SYM == SYM_DX) {
        return cplx{0.0, kx};
    } else {
        return cplx{0.0, ky};
    }
}

// W (nx, nyh) half spectrum, in place: W *= S_eff * scale; optionally writes the
// full (nx, ny) spectrum uhat = fft2(f) * S (unsymmetrised, unscaled) as well.
template <int SYM>
__global__ __launch_bounds__(256) void scalar_symbol_kernel(double2* __restrict__ W, int64_t nx,
                                                            int64_t ny, int64_t nyh, double dkx,
                                                            double dky, double k2h, double scale,
                                                            double2* __restrict__ uhat) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * nyh) return;
    int64_t i = idx / nyh, j = idx - i * nyh;
    double kx = wavenumber(i, nx, dkx), ky = wavenumber(j, ny, dky);
    double kxn = wavenumber_neg(i, nx, dkx), kyn = wavenumber_neg(j, ny, dky);
    bool is00 = (i == 0 && j == 0);
    cplx S = scalar_symbol<SYM>(kx, ky, k2h, is00);
    cplx Sn = cconj(scalar_symbol<SYM>(kxn, kyn, k2h, is00));
    cplx Se{0.5 * (S.re + Sn.re), 0.5 * (S.im + Sn.im)};
    double2 w = W[idx];
    cplx wc{w.x, w.y};
    if (uhat) {
        cplx u = cmul(wc, S);
        uhat[i * ny + j] = double2{u.re, u.im};
        // mirrored entry: fft2(f)(-k) = conj(fft2(f)(k)) for real f
        if (j > 0 && j < ny - j) {
            int64_t im = (nx - i) % nx, jm = ny - j;
            double kxm = wavenumber(im, nx, dkx), kym = wavenumber(jm, ny, dky);
            cplx Sm = scalar_symbol<SYM>(kxm, kym, k2h, false);
            cplx um = cmul(cconj(wc), Sm);
            uhat[im * ny + jm] = double2{um.re, um.im};
        }
    }
    cplx o = cmul(wc, Se);
    W[idx] = double2{o.re * scale, o.im * scale};
}

// Stokes: Fu, Fv half spectra in; U -> Fu, V -> Fv, P -> Pbuf (all scaled).
//   ph = ilap*(ikx fu + iky fv); uh = ilap*(ikx ph - fu); vh = ilap*(iky ph - fv)
// evaluated in complex arithmetic at k and at -k, then symmetrised (== .real of
// the complex ifft2 in multi_boundary/stokes.py:34-45).
struct Sym2 {
    cplx a, b;  // out = a*fu + b*fv
};
__device__ __forceinline__ void stokes_symbols(double kx, double ky, bool is00, Sym2& P, Sym2& U,
                                               Sym2& V) {
    double il = is00 ? 0.0 : 1.0 / (-kx * kx - ky * ky);
    cplx ikx{0.0, kx}, iky{0.0, ky};
    P.a = cplx{0.0, il * kx};
    P.b = cplx{0.0, il * ky};
    cplx t = cmul(ikx, P.a);
    U.a = cplx{il * (t.re - 1.0), il * t.im};
    t = cmul(ikx, P.b);
    U.b = cplx{il * t.re, il * t.im};
    t = cmul(iky, P.a);
    V.a = cplx{il * t.re, il * t.im};
    t = cmul(iky, P.b);
    V.b = cplx{il * (t.re - 1.0), il * t.im};
}
__device__ __forceinline__ cplx sym_eff(cplx s, cplx sn) {
    return cplx{0.5 * (s.re + sn.re), 0.5 * (s.im - sn.im)};
}
__global__ __launch_bounds__(256) void stokes_symbol_kernel(double2* __restrict__ Fu,
                                                            double2* __restrict__ Fv,
                                                            double2* __restrict__ Pb, int64_t nx,
                                                            int64_t ny, int64_t nyh, double dkx,
                                                            double dky, double scale) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * nyh) return;
    int64_t i = idx / nyh, j = idx - i * nyh;
    double kx = wavenumber(i, nx, dkx), ky = wavenumber(j, ny, dky);
    double kxn = wavenumber_neg(i, nx, dkx), kyn = wavenumber_neg(j, ny, dky);
    bool is00 = (i == 0 && j == 0);
    Sym2 P, U, V, Pn, Un, Vn;
    stokes_symbols(kx, ky, is00, P, U, V);
    stokes_symbols(kxn, kyn, is00, Pn, Un, Vn);
    cplx fu{Fu[idx].x, Fu[idx].y}, fv{Fv[idx].x, Fv[idx].y};
    auto apply = [&](const Sym2& s, const Sym2& sn) {
        cplx a = sym_eff(s.a, sn.a), b = sym_eff(s.b, sn.b);
        cplx x = cmul(a, fu), y = cmul(b, fv);
        return double2{(x.re + y.re) * scale, (x.im + y.im) * scale};
    };
    double2 p = apply(P, Pn), u = apply(U, Un), v = apply(V, Vn);
    Pb[idx] = p;
    Fu[idx] = u;
    Fv[idx] = v;
}

// general symbol given as a full (nx, ny) complex array
__global__ __launch_bounds__(256) void general_symbol_kernel(double2* __restrict__ W,
                                                             const double2* __restrict__ sym,
                                                             int64_t nx, int64_t ny, int64_t nyh,
                                                             double scale) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * nyh) return;
    int64_t i = idx / nyh, j = idx - i * nyh;
    int64_t im = (nx - i) % nx, jm = (ny - j) % ny;
    double2 s = sym[i * ny + j], sn = sym[im * ny + jm];
    cplx Se{0.5 * (s.x + sn.x), 0.5 * (s.y - sn.y)};
    double2 w = W[idx];
    cplx o = cmul(cplx{w.x, w.y}, Se);
    W[idx] = double2{o.re * scale, o.im * scale};
}

// expand a half spectrum of a real field to the full (nx, ny) complex spectrum
__global__ __launch_bounds__(256) void expand_hermitian_kernel(const double2* __restrict__ W,
                                                               int64_t nx, int64_t ny, int64_t nyh,
                                                               double2* __restrict__ full) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * ny) return;
    int64_t i = idx / ny, j = idx - i * ny;
    if (j < nyh) {
        full[idx] = W[i * nyh + j];
    } else {
        int64_t im = (nx - i) % nx, jm = ny - j;
        double2 w = W[im * nyh + jm];
        full[idx] = double2{w.x, -w.y};
    }
}

// 4th-order centred differences (ipde/derivatives.py:3-23)
template <int AXIS>
__global__ __launch_bounds__(256) void fd4_kernel(const double* __restrict__ f, int64_t nx,
                                                  int64_t ny, double iah, int periodic,
                                                  double* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * ny) return;
    int64_t i = idx / ny, j = idx - i * ny;
    const int64_t n = AXIS == 0 ? nx : ny;
    const int64_t c = AXIS == 0 ? i : j;
    const int64_t stride = AXIS == 0 ? ny : 1;
    double r = 0.0;
    if (c >= 2 && c < n - 2) {
        r = -(f[idx + 2 * stride] - 8.0 * f[idx + stride] + 8.0 * f[idx - stride] -
              f[idx - 2 * stride]) * iah;
    } else if (periodic) {
        auto at = [&](int64_t cc) {
            cc = ((cc % n) + n) % n;
            return AXIS == 0 ? f[cc * ny + j] : f[i * ny + cc];
        };
        r = -(at(c + 2) - 8.0 * at(c + 1) + 8.0 * at(c - 1) - at(c - 2)) * iah;
    }
    out[idx] = r;
}

__global__ __launch_bounds__(256) void scale_kernel(double* __restrict__ a, int64_t n, double s) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) a[idx] *= s;
}

int make_plan(ipde_ctx* ctx, rocfft_plan* plan, rocfft_transform_type type, size_t dims,
              const size_t* lengths, size_t batch) {
    std::lock_guard<std::mutex> global(g_rocfft_plan_mutex);
    IPDE_FFT_CHECK(ctx, rocfft_plan_create(plan, rocfft_placement_notinplace, type,
                                           rocfft_precision_double, dims, lengths, batch, nullptr));
    return IPDE_OK;
}

int ensure_work(ipde_fft_plan* p, rocfft_plan plan) {
    size_t sz = 0;
    IPDE_FFT_CHECK(p->ctx, rocfft_plan_get_work_buffer_size(plan, &sz));
    if (sz > p->work_bytes) {
        if (p->work) {
            hipStreamSynchronize(p->ctx->stream);
            hipFree(p->work);
        }
        IPDE_HIP_CHECK(p->ctx, hipMalloc(&p->work, sz));
        p->work_bytes = sz;
    }
    if (p->work_bytes)
        IPDE_FFT_CHECK(p->ctx, rocfft_execution_info_set_work_buffer(p->info, p->work, p->work_bytes));
    return IPDE_OK;
}

int exec(ipde_fft_plan* p, rocfft_plan plan, void* in, void* out) {
    IPDE_FFT_CHECK(p->ctx, rocfft_execution_info_set_stream(p->info, p->ctx->stream));
    void* ib[1] = {in};
    void* ob[1] = {out};
    IPDE_FFT_CHECK(p->ctx, rocfft_execute(plan, ib, ob, p->info));
    return IPDE_OK;
}

inline unsigned nblk(int64_t n) { return (unsigned)ceil_div64(n, 256); }

// real-array staging helpers for IPDE_HOST calls
int stage_real_in(ipde_fft_plan* p, int loc, int slot, const double* h, const double** d) {
    if (loc == IPDE_DEVICE) {
        *d = h;
        return IPDE_OK;
    }
    size_t bytes = (size_t)p->nx * p->ny * sizeof(double);
    if (!p->rbuf[slot]) IPDE_HIP_CHECK(p->ctx, hipMalloc((void**)&p->rbuf[slot], bytes));
    IPDE_HIP_CHECK(p->ctx, hipMemcpyAsync(p->rbuf[slot], h, bytes, hipMemcpyHostToDevice,
                                          p->ctx->stream));
    *d = p->rbuf[slot];
    return IPDE_OK;
}
int stage_real_out(ipde_fft_plan* p, int loc, int slot, double* h, double** d) {
    if (loc == IPDE_DEVICE || h == nullptr) {
        *d = h;
        return IPDE_OK;
    }
    size_t bytes = (size_t)p->nx * p->ny * sizeof(double);
    if (!p->rbuf[slot]) IPDE_HIP_CHECK(p->ctx, hipMalloc((void**)&p->rbuf[slot], bytes));
    *d = p->rbuf[slot];
    return IPDE_OK;
}
int finish_real_out(ipde_fft_plan* p, int loc, int slot, double* h) {
    if (loc == IPDE_DEVICE || h == nullptr) return IPDE_OK;
    size_t bytes = (size_t)p->nx * p->ny * sizeof(double);
    IPDE_HIP_CHECK(p->ctx, hipMemcpyAsync(h, p->rbuf[slot], bytes, hipMemcpyDeviceToHost,
                                          p->ctx->stream));
    return IPDE_OK;
}
int stage_cplx_in(ipde_fft_plan* p, int loc, int slot, const double* h, const double2** d) {
    if (loc == IPDE_DEVICE) {
        *d = (const double2*)h;
        return IPDE_OK;
    }
    size_t bytes = (size_t)p->nx * p->ny * sizeof(double2);
    if (!p->cbuf[slot]) IPDE_HIP_CHECK(p->ctx, hipMalloc((void**)&p->cbuf[slot], bytes));
    IPDE_HIP_CHECK(p->ctx, hipMemcpyAsync(p->cbuf[slot], h, bytes, hipMemcpyHostToDevice,
                                          p->ctx->stream));
    *d = p->cbuf[slot];
    return IPDE_OK;
}
int stage_cplx_out(ipde_fft_plan* p, int loc, int slot, double* h, double2** d) {
    if (loc == IPDE_DEVICE || h == nullptr) {
        *d = (double2*)h;
        return IPDE_OK;
    }
    size_t bytes = (size_t)p->nx * p->ny * sizeof(double2);
    if (!p->cbuf[slot]) IPDE_HIP_CHECK(p->ctx, hipMalloc((void**)&p->cbuf[slot], bytes));
    *d = p->cbuf[slot];
    return IPDE_OK;
}
int finish_cplx_out(ipde_fft_plan* p, int loc, int slot, double* h) {
    if (loc == IPDE_DEVICE || h == nullptr) return IPDE_OK;
    size_t bytes = (size_t)p->nx * p->ny * sizeof(double2);
    IPDE_HIP_CHECK(p->ctx, hipMemcpyAsync(h, p->cbuf[slot], bytes, hipMemcpyDeviceToHost,
                                          p->ctx->stream));
    return IPDE_OK;
}
int finish_sync(ipde_fft_plan* p, int loc) {
    if (loc == IPDE_HOST) IPDE_HIP_CHECK(p->ctx, hipStreamSynchronize(p->ctx->stream));
    return IPDE_OK;
}

template <int SYM>
int scalar_solve(ipde_fft_plan* p, int loc, double k2h, const double* f, double* u, double* uhat) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_CHECK_ARG(ctx, f && u);
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double* d_f;
    double* d_u;
    double2* d_uh;
    IPDE_TRY(stage_real_in(p, loc, 0, f, &d_f));
    IPDE_TRY(stage_real_out(p, loc, 1, u, &d_u));
    IPDE_TRY(stage_cplx_out(p, loc, 0, uhat, &d_uh));
    if (p->fast.ready && ctx->opt_fft2d && !uhat) {
        // three hand-written kernels: rows r2c, fused column FFT * symbol * inverse FFT, rows c2r
        IPDE_TRY(fft2d_scalar_solve(ctx, p->fast, SYM, k2h, d_f, d_u, p->keep_spec));
        p->have_spec = p->keep_spec;
        p->have_spec_general = false;
        IPDE_TRY(finish_real_out(p, loc, 1, u));
        return finish_sync(p, loc);
    }
    IPDE_TRY(exec(p, p->r2c, (void*)d_f, p->spec[0]));
    const double N = (double)p->nx * (double)p->ny;
    hipLaunchKernelGGL((scalar_symbol_kernel<SYM>), dim3(nblk(p->nx * p->nyh)), dim3(256), 0,
                       ctx->stream, p->spec[0], p->nx, p->ny, p->nyh,
                       2.0 * M_PI / (p->nx * p->hx), 2.0 * M_PI / (p->ny * p->hy), k2h, 1.0 / N,
                       d_uh);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    p->have_spec = false;
    p->have_spec_general = false;
    if (p->keep_spec && grid_interp_general_supported(p->nx, p->ny)) {
        // (the inverse transform may overwrite its input: the spectrum is copied aside)
        const size_t bytes = (size_t)p->nx * p->nyh * sizeof(double2);
        if (!p->kept_half) IPDE_HIP_CHECK(ctx, hipMalloc((void**)&p->kept_half, bytes));
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(p->kept_half, p->spec[0], bytes, hipMemcpyDeviceToDevice, ctx->stream));
        p->have_spec_general = true;
    }
    IPDE_TRY(exec(p, p->c2r, p->spec[0], d_u));
    IPDE_TRY(finish_real_out(p, loc, 1, u));
    IPDE_TRY(finish_cplx_out(p, loc, 0, uhat));
    return finish_sync(p, loc);
}

}  // namespace

extern "C" int ipde_fft_plan2d_create(ipde_ctx* ctx, int64_t nx, int64_t ny, double hx, double hy,
                                      ipde_fft_plan** out) {
    if (!ctx || !out) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, nx >= 2 && ny >= 2 && hx > 0 && hy > 0);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    rocfft_setup_once();
    ipde_fft_plan* p = new ipde_fft_plan();
    p->ctx = ctx;
    p->nx = nx;
    p->ny = ny;
    p->nyh = ny / 2 + 1;
    p->hx = hx;
    p->hy = hy;
    size_t len[2] = {(size_t)ny, (size_t)nx};  // rocFFT: fastest dimension first
    int st = make_plan(ctx, &p->r2c, rocfft_transform_type_real_forward, 2, len, 1);
    if (st == IPDE_OK) st = make_plan(ctx, &p->c2r, rocfft_transform_type_real_inverse, 2, len, 1);
    if (st == IPDE_OK && rocfft_execution_info_create(&p->info) != rocfft_status_success)
        st = IPDE_ERR_FFT;
    if (st == IPDE_OK) st = ensure_work(p, p->r2c);
    if (st == IPDE_OK) st = ensure_work(p, p->c2r);
    for (int i = 0; i < 3 && st == IPDE_OK; ++i) {
        if (hipMalloc((void**)&p->spec[i], (size_t)nx * p->nyh * sizeof(double2)) != hipSuccess)
            st = IPDE_ERR_ALLOC;
    }
    if (st == IPDE_OK && fft2d_supported(nx, ny)) st = fft2d_plan_init(ctx, p->fast, nx, ny, hx, hy);
    if (st != IPDE_OK) {
        ipde_fft_plan2d_destroy(p);
        return st;
    }
    *out = p;
    return IPDE_OK;
}

extern "C" int ipde_fft_plan2d_destroy(ipde_fft_plan* p) {
    if (!p) return IPDE_ERR_INVALID;
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    if (p->r2c) rocfft_plan_destroy(p->r2c);
    if (p->c2r) rocfft_plan_destroy(p->c2r);
    if (p->c2c_f) rocfft_plan_destroy(p->c2c_f);
    if (p->c2c_b) rocfft_plan_destroy(p->c2c_b);
    if (p->info) rocfft_execution_info_destroy(p->info);
    if (p->work) hipFree(p->work);
    fft2d_plan_free(p->fast);
    grid_interp_destroy(p->interp);
    grid_interp_destroy(p->interp_general);
    if (p->kept_half) hipFree(p->kept_half);
    for (auto& q : p->gspec)
        if (q) hipFree(q);
    for (auto& s : p->spec)
        if (s) hipFree(s);
    for (auto& s : p->rbuf)
        if (s) hipFree(s);
    for (auto& s : p->cbuf)
        if (s) hipFree(s);
    delete p;
    return IPDE_OK;
}

extern "C" int ipde_fft_plan2d_keep_spectrum(ipde_fft_plan* p, int on, int* supported) {
    if (!p) return IPDE_ERR_INVALID;
    const bool ok = (p->fast.ready && p->ctx->opt_fft2d && grid_interp_supported(p->nx, p->ny)) ||
                    grid_interp_general_supported(p->nx, p->ny);
    if (supported) *supported = ok ? 1 : 0;
    p->keep_spec = ok && on;
    if (!p->keep_spec) p->have_spec = p->have_spec_general = false;
    return IPDE_OK;
}

// The interpolation state (fine-grid plan, window factors, work grids: ~60 ms of allocation
// and host quadrature at 2048^2) ahead of the first ipde_grid_interp / ipde_grid_interp_fields —
// the path a scalar grid solve on this plan will leave its spectrum for.
extern "C" int ipde_grid_interp_prepare(ipde_fft_plan* p) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (p->fast.ready && ctx->opt_fft2d && grid_interp_supported(p->nx, p->ny)) {
        if (!p->interp) {
            grid_interp_force_shifted(ctx->opt_interp_shifted != 0);
            IPDE_TRY(grid_interp_create(ctx, p->nx, p->ny, p->hx, p->hy, &p->interp));
        }
    } else if (grid_interp_general_supported(p->nx, p->ny)) {
        if (!p->interp_general)
            IPDE_TRY(grid_interp_create(ctx, p->nx, p->ny, p->hx, p->hy, &p->interp_general, true));
    }
    return IPDE_OK;
}

extern "C" int ipde_grid_interp(ipde_fft_plan* p, int loc, int64_t np, const double* x, const double* y,
                                double* out3) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, np >= 0 && (np == 0 || (x && y && out3)));
    if (!p->have_spec && !p->have_spec_general) {
        IPDE_SET_ERR(ctx, "ipde_grid_interp: no kept spectrum (ipde_fft_plan2d_keep_spectrum, then a "
                          "scalar grid solve on this plan)");
        return IPDE_ERR_INVALID;
    }
    if (np == 0) return IPDE_OK;
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (p->have_spec_general) {
        if (!p->interp_general)
            IPDE_TRY(grid_interp_create(ctx, p->nx, p->ny, p->hx, p->hy, &p->interp_general, true));
        return grid_interp_eval_general(p->interp_general, p->kept_half, loc, np, x, y,
                                        2.0 * M_PI / (p->nx * p->hx), 2.0 * M_PI / (p->ny * p->hy), out3);
    }
    if (!p->interp) {
        grid_interp_force_shifted(ctx->opt_interp_shifted != 0);
        IPDE_TRY(grid_interp_create(ctx, p->nx, p->ny, p->hx, p->hy, &p->interp));
    }
    return grid_interp_eval(p->interp, p->fast, loc, np, x, y, 2.0 * M_PI / (p->nx * p->hx),
                            2.0 * M_PI / (p->ny * p->hy), out3);
}

extern "C" int ipde_grid_interp_fields(ipde_fft_plan* p, int loc, int nin, const double* const* fields,
                                       int nout, const int* term_start, const int* term_src,
                                       const int* term_der, const double* term_coef, int64_t np,
                                       const double* x, const double* y, double* out) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_CHECK_ARG(ctx, nin >= 1 && nin <= 3 && nout >= 1 && nout <= 8 && fields && term_start &&
                            term_src && term_der && term_coef);
    IPDE_CHECK_ARG(ctx, np >= 0 && (np == 0 || (x && y && out)));
    const bool fast = p->fast.ready && ctx->opt_fft2d && grid_interp_supported(p->nx, p->ny);
    if (!fast && !grid_interp_general_supported(p->nx, p->ny)) {
        IPDE_SET_ERR(ctx, "ipde_grid_interp_fields: no fft2d path for a %lld x %lld grid",
                     (long long)p->nx, (long long)p->ny);
        return IPDE_ERR_INVALID;
    }
    if (np == 0) return IPDE_OK;
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double* d_f[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < nin; ++k) {
        IPDE_CHECK_ARG(ctx, fields[k] != nullptr);
        IPDE_TRY(stage_real_in(p, loc, k, fields[k], &d_f[k]));
    }
    if (!fast) {
        // any grid size: rocFFT forward transforms, then the general-size interpolation
        const void* specs[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < nin; ++k) {
            if (!p->gspec[k])
                IPDE_HIP_CHECK(ctx, hipMalloc((void**)&p->gspec[k], (size_t)p->nx * p->nyh * sizeof(double2)));
            IPDE_TRY(exec(p, p->r2c, (void*)d_f[k], p->gspec[k]));
            specs[k] = p->gspec[k];
        }
        if (!p->interp_general)
            IPDE_TRY(grid_interp_create(ctx, p->nx, p->ny, p->hx, p->hy, &p->interp_general, true));
        return grid_interp_fields_general(p->interp_general, nin, specs, nout, term_start, term_src, term_der,
                                          term_coef, loc, np, x, y, 2.0 * M_PI / (p->nx * p->hx),
                                          2.0 * M_PI / (p->ny * p->hy), out);
    }
    if (!p->interp) {
        grid_interp_force_shifted(ctx->opt_interp_shifted != 0);
        IPDE_TRY(grid_interp_create(ctx, p->nx, p->ny, p->hx, p->hy, &p->interp));
    }
    return grid_interp_fields(p->interp, p->fast, nin, d_f, nout, term_start, term_src, term_der,
                              term_coef, loc, np, x, y, 2.0 * M_PI / (p->nx * p->hx),
                              2.0 * M_PI / (p->ny * p->hy), out);
}

extern "C" int ipde_poisson_grid_solve(ipde_fft_plan* p, int loc, const double* f, double* u,
                                       double* uhat) {
    return scalar_solve<SYM_POISSON>(p, loc, 0.0, f, u, uhat);
}

extern "C" int ipde_modhelm_grid_solve(ipde_fft_plan* p, int loc, double k, const double* f,
                                       double* u, double* uhat) {
    return scalar_solve<SYM_MODHELM>(p, loc, k * k, f, u, uhat);
}

extern "C" int ipde_fourier_deriv(ipde_fft_plan* p, int loc, const double* f, int axis,
                                  double* out) {
    if (!p) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(p->ctx, axis == 0 || axis == 1);
    if (axis == 0) return scalar_solve<SYM_DX>(p, loc, 0.0, f, out, nullptr);
    return scalar_solve<SYM_DY>(p, loc, 0.0, f, out, nullptr);
}

extern "C" int ipde_fourier_multiply(ipde_fft_plan* p, int loc, const double* f,
                                     const double* sym_c, double* out) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_CHECK_ARG(ctx, f && sym_c && out);
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double* d_f;
    const double2* d_s;
    double* d_o;
    IPDE_TRY(stage_real_in(p, loc, 0, f, &d_f));
    IPDE_TRY(stage_cplx_in(p, loc, 0, sym_c, &d_s));
    IPDE_TRY(stage_real_out(p, loc, 1, out, &d_o));
    IPDE_TRY(exec(p, p->r2c, (void*)d_f, p->spec[0]));
    const double N = (double)p->nx * (double)p->ny;
    hipLaunchKernelGGL(general_symbol_kernel, dim3(nblk(p->nx * p->nyh)), dim3(256), 0, ctx->stream,
                       p->spec[0], d_s, p->nx, p->ny, p->nyh, 1.0 / N);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    IPDE_TRY(exec(p, p->c2r, p->spec[0], d_o));
    IPDE_TRY(finish_real_out(p, loc, 1, out));
    return finish_sync(p, loc);
}

extern "C" int ipde_stokes_grid_solve(ipde_fft_plan* p, int loc, const double* fu,
                                      const double* fv, double* u, double* v, double* pr) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_CHECK_ARG(ctx, fu && fv && u && v && pr);
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double *d_fu, *d_fv;
    double *d_u, *d_v, *d_p;
    IPDE_TRY(stage_real_in(p, loc, 0, fu, &d_fu));
    IPDE_TRY(stage_real_in(p, loc, 1, fv, &d_fv));
    IPDE_TRY(stage_real_out(p, loc, 2, u, &d_u));
    IPDE_TRY(stage_real_out(p, loc, 3, v, &d_v));
    IPDE_TRY(stage_real_out(p, loc, 4, pr, &d_p));
    if (p->fast.ready && ctx->opt_fft2d) {
        // hand-written pipeline (fft2d.hip): the kept scalar spectrum, if any, is overwritten
        p->have_spec = false;
        IPDE_TRY(fft2d_stokes_solve(ctx, p->fast, d_fu, d_fv, d_u, d_v, d_p));
        IPDE_TRY(finish_real_out(p, loc, 2, u));
        IPDE_TRY(finish_real_out(p, loc, 3, v));
        IPDE_TRY(finish_real_out(p, loc, 4, pr));
        return finish_sync(p, loc);
    }
    IPDE_TRY(exec(p, p->r2c, (void*)d_fu, p->spec[0]));
    IPDE_TRY(exec(p, p->r2c, (void*)d_fv, p->spec[1]));
    const double N = (double)p->nx * (double)p->ny;
    hipLaunchKernelGGL(stokes_symbol_kernel, dim3(nblk(p->nx * p->nyh)), dim3(256), 0, ctx->stream,
                       p->spec[0], p->spec[1], p->spec[2], p->nx, p->ny, p->nyh,
                       2.0 * M_PI / (p->nx * p->hx), 2.0 * M_PI / (p->ny * p->hy), 1.0 / N);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    IPDE_TRY(exec(p, p->c2r, p->spec[0], d_u));
    IPDE_TRY(exec(p, p->c2r, p->spec[1], d_v));
    IPDE_TRY(exec(p, p->c2r, p->spec[2], d_p));
    IPDE_TRY(finish_real_out(p, loc, 2, u));
    IPDE_TRY(finish_real_out(p, loc, 3, v));
    IPDE_TRY(finish_real_out(p, loc, 4, pr));
    return finish_sync(p, loc);
}

extern "C" int ipde_fft2_r2c_full(ipde_fft_plan* p, int loc, const double* in_r, double* out_c) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_CHECK_ARG(ctx, in_r && out_c);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double* d_f;
    double2* d_o;
    IPDE_TRY(stage_real_in(p, loc, 0, in_r, &d_f));
    IPDE_TRY(stage_cplx_out(p, loc, 0, out_c, &d_o));
    IPDE_TRY(exec(p, p->r2c, (void*)d_f, p->spec[0]));
    hipLaunchKernelGGL(expand_hermitian_kernel, dim3(nblk(p->nx * p->ny)), dim3(256), 0, ctx->stream,
                       (const double2*)p->spec[0], p->nx, p->ny, p->nyh, d_o);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    IPDE_TRY(finish_cplx_out(p, loc, 0, out_c));
    return finish_sync(p, loc);
}

extern "C" int ipde_fft2_c2c(ipde_fft_plan* p, int loc, int direction, const double* in_c,
                             double* out_c) {
    if (!p) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = p->ctx;
    IPDE_CHECK_ARG(ctx, in_c && out_c && (direction == -1 || direction == 1));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t len[2] = {(size_t)p->ny, (size_t)p->nx};
    rocfft_plan* pl = direction < 0 ? &p->c2c_f : &p->c2c_b;
    if (!*pl) {
        IPDE_TRY(make_plan(ctx, pl,
                           direction < 0 ? rocfft_transform_type_complex_forward
                                         : rocfft_transform_type_complex_inverse,
                           2, len, 1));
        IPDE_TRY(ensure_work(p, *pl));
    }
    const double2* d_i;
    double2* d_o;
    IPDE_TRY(stage_cplx_in(p, loc, 0, in_c, &d_i));
    IPDE_TRY(stage_cplx_out(p, loc, 1, out_c, &d_o));
    IPDE_TRY(exec(p, *pl, (void*)d_i, d_o));
    if (direction > 0) {
        int64_t n2 = 2 * p->nx * p->ny;
        hipLaunchKernelGGL(scale_kernel, dim3(nblk(n2)), dim3(256), 0, ctx->stream, (double*)d_o, n2,
                           1.0 / ((double)p->nx * (double)p->ny));
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    IPDE_TRY(finish_cplx_out(p, loc, 1, out_c));
    return finish_sync(p, loc);
}

extern "C" int ipde_fd4(ipde_ctx* ctx, int loc, int64_t nx, int64_t ny, double h, int axis,
                        int periodic_fix, const double* f, double* out) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, f && out && nx >= 5 && ny >= 5 && (axis == 0 || axis == 1));
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double* d_f;
    double* d_o;
    IPDE_TRY(ipde_stage_in(ctx, loc, 0, f, nx * ny, &d_f));
    IPDE_TRY(ipde_stage_out(ctx, loc, 1, out, nx * ny, &d_o));
    double iah = 1.0 / (12.0 * h);
    if (axis == 0)
        hipLaunchKernelGGL(fd4_kernel<0>, dim3(nblk(nx * ny)), dim3(256), 0, ctx->stream, d_f, nx, ny,
                           iah, periodic_fix, d_o);
    else
        hipLaunchKernelGGL(fd4_kernel<1>, dim3(nblk(nx * ny)), dim3(256), 0, ctx->stream, d_f, nx, ny,
                           iah, periodic_fix, d_o);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return ipde_stage_finish(ctx, loc, 1, out, nx * ny);
}

// ---------------------------------------------------------------------------
// batched 1-D complex FFT along the last axis (annular solvers, utilities.fft)
struct Fft1Plan {
    rocfft_plan fwd = nullptr, bwd = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;
};

int ipde_fft1_get(ipde_ctx* ctx, int64_t batch, int64_t n, Fft1Plan** out) {
    rocfft_setup_once();
    std::lock_guard<std::mutex> guard(ctx->fft1_mutex);
    auto key = std::make_pair(batch, n);
    auto it = ctx->fft1_plans.find(key);
    if (it != ctx->fft1_plans.end()) {
        *out = (Fft1Plan*)it->second;
        return IPDE_OK;
    }
    std::lock_guard<std::mutex> global(g_rocfft_plan_mutex);
    Fft1Plan* fp = new Fft1Plan();
    size_t len[1] = {(size_t)n};
    IPDE_FFT_CHECK(ctx, rocfft_plan_create(&fp->fwd, rocfft_placement_notinplace,
                                           rocfft_transform_type_complex_forward,
                                           rocfft_precision_double, 1, len, (size_t)batch, nullptr));
    IPDE_FFT_CHECK(ctx, rocfft_plan_create(&fp->bwd, rocfft_placement_notinplace,
                                           rocfft_transform_type_complex_inverse,
                                           rocfft_precision_double, 1, len, (size_t)batch, nullptr));
    IPDE_FFT_CHECK(ctx, rocfft_execution_info_create(&fp->info));
    size_t a = 0, b = 0;
    IPDE_FFT_CHECK(ctx, rocfft_plan_get_work_buffer_size(fp->fwd, &a));
    IPDE_FFT_CHECK(ctx, rocfft_plan_get_work_buffer_size(fp->bwd, &b));
    fp->work_bytes = a > b ? a : b;
    if (fp->work_bytes) {
        IPDE_HIP_CHECK(ctx, hipMalloc(&fp->work, fp->work_bytes));
        IPDE_FFT_CHECK(ctx, rocfft_execution_info_set_work_buffer(fp->info, fp->work, fp->work_bytes));
    }
    ctx->fft1_plans[key] = fp;
    *out = fp;
    return IPDE_OK;
}

// device-pointer execution, unscaled; used by the annular solvers too
int ipde_fft1_exec(ipde_ctx* ctx, int64_t batch, int64_t n, int direction, const void* in,
                   void* out) {
    Fft1Plan* fp;
    IPDE_TRY(ipde_fft1_get(ctx, batch, n, &fp));
    IPDE_FFT_CHECK(ctx, rocfft_execution_info_set_stream(fp->info, ctx->stream));
    void* ib[1] = {(void*)in};
    void* ob[1] = {out};
    IPDE_FFT_CHECK(ctx, rocfft_execute(direction < 0 ? fp->fwd : fp->bwd, ib, ob, fp->info));
    return IPDE_OK;
}

// Create (and run-time compile) the plan of a batched 1-D transform ahead of its first use;
// thread safe, touches no staging buffers, launches nothing.
extern "C" int ipde_fft1_prepare(ipde_ctx* ctx, int64_t batch, int64_t n) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, batch >= 1 && n >= 1);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Fft1Plan* fp;
    return ipde_fft1_get(ctx, batch, n, &fp);
}

void ipde_fft1_plans_destroy(ipde_ctx* ctx) {
    for (auto& kv : ctx->fft1_plans) {
        Fft1Plan* fp = (Fft1Plan*)kv.second;
        if (fp->fwd) rocfft_plan_destroy(fp->fwd);
        if (fp->bwd) rocfft_plan_destroy(fp->bwd);
        if (fp->info) rocfft_execution_info_destroy(fp->info);
        if (fp->work) hipFree(fp->work);
        delete fp;
    }
    ctx->fft1_plans.clear();
}

// ---- noise cut of a QFS source density (ipde_amd/qfs.py: Stokes_QFS._lowpass) ------------------------------
// The collocation's singular values fall like e^{-|k| alpha h} / |k|: the spectrum of a resolved density decays,
// reaches a minimum where the amplified noise of the data takes over, and RISES again towards the Nyquist
// frequency.  The rule: band maxima B_j of max(|z_k|, |z_{-k}|) (z = x + i y carries both components), m_j their
// running minimum; the first band j <= top (0.9 Nyquist) whose whole tail up to top lies above rise * m_{j-1}, with
// m_{j-1} < floor_rel * max B, marks the turnaround: every mode above the band where the minimum was reached is
// removed.  No sustained rise after a deep minimum (an under-resolved density, an isolated high mode): nothing is.
namespace {
__global__ __launch_bounds__(256) void cut_pack_kernel(double2* __restrict__ z, const double* __restrict__ mu, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) z[i] = double2{mu[i], mu[n + i]};
}
__global__ __launch_bounds__(256) void cut_unpack_kernel(double* __restrict__ out, const double2* __restrict__ z, int64_t n,
                                                         double s) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        out[i] = z[i].x * s;
        out[n + i] = z[i].y * s;
    }
}
constexpr int CUT_MAXB = 1024;
__global__ __launch_bounds__(1024) void density_noise_cut_kernel(double2* __restrict__ Z, int n, int w, int nbands,
                                                                 double rise, double floor_rel, int kcap,
                                                                 int* __restrict__ kcut_out) {
    __shared__ double B[CUT_MAXB];
    __shared__ int s_kc;
    const int tid = threadIdx.x, H = n / 2;
    for (int j = tid; j < nbands; j += 1024) {
        double b = 0.0;
        const int k1 = (j + 1) * w - 1 < H ? (j + 1) * w - 1 : H;
        for (int k = j * w; k <= k1; ++k) {
            const double2 p = Z[k], q = Z[(n - k) % n];
            b = fmax(b, fmax(sqrt(p.x * p.x + p.y * p.y), sqrt(q.x * q.x + q.y * q.y)));
        }
        B[j] = b;
    }
    if (tid == 0) s_kc = nbands;                 // (first band that meets the rule; nbands: none)
    __syncthreads();
    // Every band j evaluates the rule for itself (band = thread: nbands <= 1024): m = min B[0 .. j-1] and the first
    // band attaining it, S = min B[j .. top], the largest band — loops over LDS with wave-uniform addresses
    // (broadcast reads), a few microseconds; one thread walking the bands took 70-90 us of LDS latency per call.
    int top = (int)(0.9 * H) / w;
    if (top > nbands - 1) top = nbands - 1;
    const int j = tid;
    double gmax = 0.0, m = B[0], S = 1.0e300;
    int jmin = 0;
    for (int i = 0; i < nbands; ++i) {
        const double b = B[i];
        gmax = fmax(gmax, b);
        if (i >= 1 && i < j && b < m) {
            m = b;
            jmin = i;
        }
        if (i >= j && i <= top) S = fmin(S, b);
    }
    if (j >= 1 && j <= top && m < floor_rel * gmax && S > rise * m) atomicMin(&s_kc, j);
    __syncthreads();
    const int jfirst = s_kc;
    __syncthreads();
    if (jfirst == nbands) {
        if (tid == 0) s_kc = H < kcap ? H : kcap;
    } else if (j == jfirst) {
        int kc = (jmin + 1) * w - 1;
        if (kc > kcap) kc = kcap;
        if (kc > H) kc = H;
        s_kc = kc;
    }
    __syncthreads();
    if (tid == 0 && kcut_out) *kcut_out = s_kc;
    const int kc = s_kc;
    for (int k = kc + 1 + tid; k <= H; k += 1024) {
        Z[k] = double2{0.0, 0.0};
        Z[n - k] = double2{0.0, 0.0};
    }
}
}  // namespace

extern "C" int ipde_density_noise_cut(ipde_ctx* ctx, int64_t n, const double* mu, double* out, double rise,
                                      double floor_rel, double max_keep, int* kcut) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, mu && out && n >= 16 && n < (1ll << 30) && rise > 1.0 && floor_rel > 0.0 && max_keep > 0.0);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->cut_work, 2 * (size_t)n * sizeof(double2)));
    double2* z = (double2*)ctx->cut_work.p;
    double2* zh = z + n;
    const int H = (int)(n / 2);
    int w = (H + 1 + 511) / 512;
    if (w < 4) w = 4;
    const int nbands = (H + w) / w;                       // bands cover k = 0 .. H
    const int kcap = max_keep >= 1.0 ? H : (int)(max_keep * H);
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(cut_pack_kernel, dim3(nblk(n)), dim3(256), 0, st, z, mu, n);
    IPDE_TRY(ipde_fft1_exec(ctx, 1, n, -1, z, zh));
    hipLaunchKernelGGL(density_noise_cut_kernel, dim3(1), dim3(1024), 0, st, zh, (int)n, w, nbands, rise, floor_rel,
                       kcap, kcut);
    IPDE_TRY(ipde_fft1_exec(ctx, 1, n, +1, zh, z));
    hipLaunchKernelGGL(cut_unpack_kernel, dim3(nblk(n)), dim3(256), 0, st, out, (const double2*)z, n, 1.0 / (double)n);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_fft1_c2c(ipde_ctx* ctx, int loc, int64_t batch, int64_t n, int direction,
                             const double* in_c, double* out_c) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, in_c && out_c && batch >= 1 && n >= 1 && (direction == -1 || direction == 1));
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double* d_i;
    double* d_o;
    IPDE_TRY(ipde_stage_in(ctx, loc, 0, in_c, 2 * batch * n, &d_i));
    IPDE_TRY(ipde_stage_out(ctx, loc, 1, out_c, 2 * batch * n, &d_o));
    IPDE_TRY(ipde_fft1_exec(ctx, batch, n, direction, d_i, d_o));
    if (direction > 0) {
        hipLaunchKernelGGL(scale_kernel, dim3(nblk(2 * batch * n)), dim3(256), 0, ctx->stream, d_o,
                           2 * batch * n, 1.0 / (double)n);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    return ipde_stage_finish(ctx, loc, 1, out_c, 2 * batch * n);
}
