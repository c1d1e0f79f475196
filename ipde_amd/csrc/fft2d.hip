// Hand-written 2-D real FFT pipeline for the periodic grid operators on
// power-of-two grids: three kernels per "ifft2(fft2(f) * S).real" instead of rocFFT's nine.
//
//   row_r2c   : every grid row (ny reals) -> packed half spectrum row (ny/2 complex, the real
//               Nyquist entry in the imaginary part of entry 0), one pass over HBM: the
//               packed-real trick (a length ny/2 complex FFT + one butterfly with the
//               mirrored entry) entirely in registers / LDS.
//   col_kernel: for a block of C = 4 adjacent ky columns: forward length-nx FFT, multiply by
//               the (Hermitian-symmetrised, scaled) operator symbol, inverse FFT — the
//               spectrum never goes back to HBM between the two transforms.  Global access is
//               64-byte row segments of the row-major half spectrum (measured 5.5 / 6.0 TB/s
//               read / write with enough workgroups, profiles/r02_segment_bw_probe.txt);
//               blocks are handed out XCD-aware so neighbouring blocks share L2 lines.
//   row_c2r   : packed half spectrum rows -> real rows (mirror butterfly + length ny/2 FFT).
//
// HBM traffic: 3 x (read + write of one field) = 201 MB at 2048^2 against 604 MB for the
// rocFFT pipeline it replaces (profiles/r02_fft_rocfft_breakdown.json).  Roofline: HBM.
//
// The FFT itself: Stockham autosort, three passes (radix 16 / 8 / 4 butterflies in
// registers), a thread holds P = 16 (8) points in the "strided natural" layout
// a[t + T q]; between passes the points travel through LDS (one complex slot per point,
// ds_write_b128 / ds_read_b128), positions padded by one slot every 16 so that the radix-16
// scatter (lane stride 16 slots) is bank-conflict free.  First-pass inputs and last-pass
// outputs stay in registers in natural order, so operator symbols and the real-transform
// butterflies index plainly.  64-thread transforms own their LDS region: no barriers.
//
// Sizes: nx in {512, 1024, 2048, 4096}, ny in {1024, 2048, 4096, 8192}; everything else
// stays on the rocFFT path (spectral.hip).
#include <atomic>

#include "ipde_common.h"
#include "fft2d.h"
#include "fft_core.h"

namespace {

using namespace fftcore;

// ---- operator symbols (the arithmetic of spectral.hip's scalar_symbol_kernel) --------------
__device__ __forceinline__ double wavenumber(int i, int n, double dk) {
    int s = (i < (n + 1) / 2) ? i : i - n;
    return (double)s * dk;
}
// 1 / x to ~1 ulp in 5 instructions: the v_rcp_f64 seed (about 27 bits) and two Newton steps,
// against ~14 for the IEEE division (no denormal / inf handling needed: |x| is O(1 .. 1e7))
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// Hermitian-symmetrised effective symbol (S(k) + conj S(-k)) / 2: what `.real` of the complex
// pipeline amounts to.  `-k` as the complex pipeline sees it maps a Nyquist index onto itself,
// so the even real symbols are unchanged and the odd imaginary ones lose their Nyquist row /
// column (spectral.hip's scalar_symbol_kernel evaluates the same thing the long way).
template <int SYM>
__device__ __forceinline__ cd effective_symbol(int i, int j, int nx, int ny, double dkx, double dky,
                                               double k2h) {
    const double kx = wavenumber(i, nx, dkx), ky = wavenumber(j, ny, dky);
    if (SYM == FFT2D_SYM_POISSON) {
        return (i == 0 && j == 0) ? cd{0.0, 0.0} : cd{fast_rcp(-kx * kx - ky * ky), 0.0};
    } else if (SYM == FFT2D_SYM_MODHELM) {
        return cd{fast_rcp(k2h - (-kx * kx - ky * ky)), 0.0};
    } else if (SYM == FFT2D_SYM_DX) {
        return cd{0.0, (2 * i == nx) ? 0.0 : kx};
    } else if (SYM == FFT2D_SYM_DY) {
        return cd{0.0, (2 * j == ny) ? 0.0 : ky};
    }
    return cd{1.0, 0.0};
}

// The half spectrum is kept PACKED: W[row][k], k = 0 .. ny/2 - 1, with the (real) Nyquist
// entry k = ny/2 stored in the imaginary part of the (real) k = 0 entry.  Rows are then
// exactly ny/2 complex numbers (16 KiB at ny = 2048, line aligned) and the column pass has
// ny/8 blocks of four columns — 256 at ny = 2048, one per CU — instead of one more.

// ---- kernel A: rows, real -> packed half spectrum -------------------------------------------
template <int NY>
__global__ __launch_bounds__(256) void row_r2c_kernel(const double* __restrict__ f,
                                                      cd* __restrict__ W, const cd* __restrict__ tw_h,
                                                      const cd* __restrict__ tw_ny) {
    constexpr int H = NY / 2;
    using G = Cfg<H>;
    constexpr int T = G::T, P = G::P, RPW = 256 / T;
    extern __shared__ double2 lds_raw[];   // (double2: the dynamic LDS base is 16-byte aligned)
    cd* lds = (cd*)lds_raw;
    const int tid = threadIdx.x, sub = tid / T, t = tid % T;
    const int64_t row = (int64_t)blockIdx.x * RPW + sub;
    cd* buf = lds + sub * lds_slots<H>();
    const double2* src = (const double2*)(f + row * NY);
    // every table entry the thread will need, requested before the data (one latency for all)
    const PassTw<H> pw = load_pass_twiddles<H>(t, tw_h);
    const cd wt = tw_ny[t];   // exp(-2 pi i t / NY)
    cd v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        double2 a = src[t + T * q];
        v[q] = cd{a.x, a.y};
    }
    fft_regs<H, -1, (T == 64)>(v, t, pw, buf);
    cd m[P];   // Z[(H - k) mod H]
    gather_mirror<H, (T == 64)>(v, m, t, buf);
    cd* dst = W + row * H;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int k = t + T * q;
        cd zk = v[q], zm = cconj(m[q]);
        cd e = cd{0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y)};
        cd d = cd{0.5 * (zk.x - zm.x), 0.5 * (zk.y - zm.y)};
        // exp(-2 pi i k / NY) = exp(-2 pi i t / NY) exp(-2 pi i T q / NY): the second factor sits at
        // a wave-uniform address (scalar load)
        cd w = q == 0 ? wt : cmul(wt, tw_ny[T * q]);
        cd wd = cmul(w, d);
        cd X = cd{e.x + wd.y, e.y - wd.x};   // e - i w d
        if (k == 0) X = cd{zk.x + zk.y, zk.x - zk.y};   // {X[0], X[ny/2]}, both real
        dst[k] = X;
    }
}

// ---- kernel C: rows, packed half spectrum -> real --------------------------------------------
// out = (unnormalised c2r) / 2 : the missing factor 2 is folded into the symbol's scale
template <int NY>
__global__ __launch_bounds__(256) void row_c2r_kernel(const cd* __restrict__ W, double* __restrict__ out,
                                                      const cd* __restrict__ tw_h,
                                                      const cd* __restrict__ tw_ny) {
    constexpr int H = NY / 2;
    using G = Cfg<H>;
    constexpr int T = G::T, P = G::P, RPW = 256 / T;
    extern __shared__ double2 lds_raw[];   // (double2: the dynamic LDS base is 16-byte aligned)
    cd* lds = (cd*)lds_raw;
    const int tid = threadIdx.x, sub = tid / T, t = tid % T;
    const int64_t row = (int64_t)blockIdx.x * RPW + sub;
    cd* buf = lds + sub * lds_slots<H>();
    const cd* src = W + row * H;
    const PassTw<H> pw = load_pass_twiddles<H>(t, tw_h);
    const cd wt = cconj(tw_ny[t]);   // exp(+2 pi i t / NY)
    cd v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = src[t + T * q];
    cd m[P];   // X[H - k]
    gather_mirror<H, (T == 64)>(v, m, t, buf);
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int k = t + T * q;
        cd xk = v[q], xm = cconj(m[q]);
        if (k == 0) {   // unpack {X[0], X[ny/2]}
            xm = cd{xk.y, 0.0};
            xk = cd{xk.x, 0.0};
        }
        cd e = cd{0.5 * (xk.x + xm.x), 0.5 * (xk.y + xm.y)};
        cd d = cd{0.5 * (xk.x - xm.x), 0.5 * (xk.y - xm.y)};
        cd w = q == 0 ? wt : cmul(wt, cconj(tw_ny[T * q]));   // exp(+2 pi i k / NY), k = t + T q
        cd wd = cmul(w, d);
        v[q] = cd{e.x - wd.y, e.y + wd.x};   // e + i w d
    }
    fft_regs<H, +1, (T == 64)>(v, t, pw, buf);
    double2* dst = (double2*)(out + row * NY);
#pragma unroll
    for (int q = 0; q < P; ++q) dst[t + T * q] = double2{v[q].x, v[q].y};
}

// ---- kernel B: columns, forward FFT x symbol x inverse FFT in one pass ------------------------
// MODE 0: fused (forward, symbol, inverse); 1: forward only; 2: inverse only.
// Column 0 carries two real columns (ky = 0 in the real part, the Nyquist ky in the imaginary
// part): its transform G = F0 + i FH is split with the mirrored entry, F0 = (G(k) + conj
// G(-k)) / 2, FH = (G(k) - conj G(-k)) / 2i, each half gets its own symbol, and U0 + i UH
// goes back through the inverse transform (both results are real sequences again).
template <int NX, int C, int SYM, int MODE>
__global__ __launch_bounds__(C* Cfg<NX>::T) void col_kernel(cd* __restrict__ W, int pitch,
                                                           int nblocks, int ny, double dkx,
                                                           double dky, double k2h, double scale,
                                                           const cd* __restrict__ tw_x,
                                                           cd* __restrict__ spec_out) {
    using G = Cfg<NX>;
    // column regions are offset by 4 slots (16 banks) from one another: the 16 lanes of a
    // ds_read_b128 group (4 columns x 4 values of t) then hit 16 different 4-bank slots
    constexpr int T = G::T, P = G::P, NPC = lds_slots<NX>() + 4;
    extern __shared__ double2 lds_raw[];   // (double2: the dynamic LDS base is 16-byte aligned)
    cd* lds = (cd*)lds_raw;
    // XCD-aware hand-out: workgroup w runs on XCD w % 8; give each XCD a contiguous range of
    // column blocks so that blocks sharing 128-byte lines share an L2
    const int per = (nblocks + 7) / 8;
    const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (blk >= nblocks) return;
    // Thread layout: column = tid % C, t = tid / C.  A load instruction then covers 64 / C rows
    // x C columns (64-byte row segments), and the values a thread loads, rows t + T i of its
    // column, ARE the strided natural layout of that column's transform: no staging permute on
    // the way in or out.  (The T threads of a column sit in every wave: all LDS exchanges are
    // workgroup-wide.)
    const int tid = threadIdx.x;
    const int col = tid % C, t = tid / C;
    const int j0 = blk * C;
    cd* base = W + j0 + col;
    const PassTw<NX> pw = load_pass_twiddles<NX>(t, tw_x);   // (both transforms; before the data)
    cd v[P];
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = base[(int64_t)(t + T * i) * pitch];
    cd* buf = lds + col * NPC;
    if (MODE != 2) fft_regs<NX, -1, false>(v, t, pw, buf);
    if (MODE == 0) {
        const int j = j0 + col;
        if (blk == 0) {   // (uniform over the workgroup)
            cd m[P];
            gather_mirror<NX, false>(v, m, t, buf);
            if (col == 0) {
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const int i = t + T * q;
                    cd g = v[q], gm = cconj(m[q]);
                    cd f0 = cd{0.5 * (g.x + gm.x), 0.5 * (g.y + gm.y)};
                    cd fh = cd{0.5 * (g.y - gm.y), -0.5 * (g.x - gm.x)};   // (g - gm) / 2i
                    cd u0 = cmul(f0, effective_symbol<SYM>(i, 0, NX, ny, dkx, dky, k2h));
                    cd uh = cmul(fh, effective_symbol<SYM>(i, ny / 2, NX, ny, dkx, dky, k2h));
                    v[q] = cd{(u0.x - uh.y) * scale, (u0.y + uh.x) * scale};   // u0 + i uh
                }
            }
        }
        if (!(blk == 0 && col == 0)) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                cd S = effective_symbol<SYM>(t + T * q, j, NX, ny, dkx, dky, k2h);
                cd o = cmul(v[q], S);
                v[q] = cd{o.x * scale, o.y * scale};
            }
        }
    }
    if (MODE == 0 && spec_out) {   // keep fft2(f) * symbol * scale (packed) for the interpolation
        cd* so = spec_out + j0 + col;
#pragma unroll
        for (int q = 0; q < P; ++q) so[(int64_t)(t + T * q) * pitch] = v[q];
    }
    if (MODE != 1) fft_regs<NX, +1, false>(v, t, pw, buf);
#pragma unroll
    for (int i = 0; i < P; ++i) base[(int64_t)(t + T * i) * pitch] = v[i];
}

// ---- Stokes symbols on packed half spectra ---------------------------------------------------
// Fu, Fv (after the forward column pass) -> U, V in place and P into Pb, all scaled:
//   ph = ilap (ikx fu + iky fv); uh = ilap (ikx ph - fu); vh = ilap (iky ph - fv)
// evaluated at k and at -k (the Nyquist index maps onto itself) and symmetrised, which is what
// `.real` of the reference's complex ifft2 amounts to (multi_boundary/stokes.py:34-45; the same
// arithmetic as spectral.hip's stokes_symbol_kernel).  Column 0 carries ky = 0 and the Nyquist
// column packed: unpacked with the mirrored row, each half gets its own symbols, packed again.
__device__ __forceinline__ double wavenumber_neg(int i, int n, double dk) {
    if ((n & 1) == 0 && i == n / 2) return wavenumber(i, n, dk);
    return -wavenumber(i, n, dk);
}
struct Sym2 {
    cd a, b;   // out = a fu + b fv
};
__device__ __forceinline__ void stokes_symbols(double kx, double ky, bool is00, Sym2& P, Sym2& U,
                                               Sym2& V) {
    const double il = is00 ? 0.0 : 1.0 / (-kx * kx - ky * ky);
    const cd ikx{0.0, kx}, iky{0.0, ky};
    P.a = cd{0.0, il * kx};
    P.b = cd{0.0, il * ky};
    cd t = cmul(ikx, P.a);
    U.a = cd{il * (t.x - 1.0), il * t.y};
    t = cmul(ikx, P.b);
    U.b = cd{il * t.x, il * t.y};
    t = cmul(iky, P.a);
    V.a = cd{il * t.x, il * t.y};
    t = cmul(iky, P.b);
    V.b = cd{il * (t.x - 1.0), il * t.y};
}
__device__ __forceinline__ cd sym_eff(cd s, cd sn) { return cd{0.5 * (s.x + sn.x), 0.5 * (s.y - sn.y)}; }

__device__ __forceinline__ void stokes_apply_mode(int i, int j, int nx, int ny, double dkx, double dky,
                                                  cd fu, cd fv, double scale, cd& u, cd& v, cd& p) {
    const bool is00 = (i == 0 && j == 0);
    Sym2 P, U, V, Pn, Un, Vn;
    stokes_symbols(wavenumber(i, nx, dkx), wavenumber(j, ny, dky), is00, P, U, V);
    stokes_symbols(wavenumber_neg(i, nx, dkx), wavenumber_neg(j, ny, dky), is00, Pn, Un, Vn);
    auto app = [&](const Sym2& s, const Sym2& sn) {
        cd x = cmul(sym_eff(s.a, sn.a), fu), y = cmul(sym_eff(s.b, sn.b), fv);
        return cd{(x.x + y.x) * scale, (x.y + y.y) * scale};
    };
    p = app(P, Pn);
    u = app(U, Un);
    v = app(V, Vn);
}

__global__ __launch_bounds__(256) void stokes_packed_symbol_kernel(cd* __restrict__ Fu,
                                                                   cd* __restrict__ Fv,
                                                                   cd* __restrict__ Pb, int nx, int ny,
                                                                   double dkx, double dky, double scale) {
    const int H = ny / 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)nx * H) return;
    const int i = (int)(idx / H), j = (int)(idx - (int64_t)i * H);
    cd u, v, p;
    if (j > 0) {
        stokes_apply_mode(i, j, nx, ny, dkx, dky, Fu[idx], Fv[idx], scale, u, v, p);
    } else {
        // both rows i and nx - i read each other's packed entry before either writes: the pair
        // is handled by ONE thread (the lower index; i = 0 and i = nx/2 are their own mirrors)
        const int im = (nx - i) % nx;
        if (im < i) return;
        auto unpack = [](cd g, cd gm, cd& f0, cd& fh) {
            f0 = cd{0.5 * (g.x + gm.x), 0.5 * (g.y - gm.y)};
            fh = cd{0.5 * (g.y + gm.y), -0.5 * (g.x - gm.x)};
        };
        const int64_t a = (int64_t)i * H, b = (int64_t)im * H;
        cd gu = Fu[a], gum = Fu[b], gv = Fv[a], gvm = Fv[b];
        for (int side = 0; side < (im == i ? 1 : 2); ++side) {
            const int r = side ? im : i;
            cd fu0, fuh, fv0, fvh;
            if (side == 0) {
                unpack(gu, gum, fu0, fuh);
                unpack(gv, gvm, fv0, fvh);
            } else {
                unpack(gum, gu, fu0, fuh);
                unpack(gvm, gv, fv0, fvh);
            }
            cd u0, v0, p0, uh, vh, ph;
            stokes_apply_mode(r, 0, nx, ny, dkx, dky, fu0, fv0, scale, u0, v0, p0);
            stokes_apply_mode(r, H, nx, ny, dkx, dky, fuh, fvh, scale, uh, vh, ph);
            const int64_t o = (int64_t)r * H;
            Fu[o] = cd{u0.x - uh.y, u0.y + uh.x};
            Fv[o] = cd{v0.x - vh.y, v0.y + vh.x};
            Pb[o] = cd{p0.x - ph.y, p0.y + ph.x};
        }
        return;
    }
    Fu[idx] = u;
    Fv[idx] = v;
    Pb[idx] = p;
}

// (once per kernel and device: the attribute call is ~2 us of host time, a third of a launch;
// `done` is the caller's static flag word for that kernel instantiation, one bit per device)
template <typename K>
int allow_lds(ipde_ctx* ctx, K kernel, size_t bytes, std::atomic<unsigned long long>& done) {
    if (bytes <= 48 * 1024) return IPDE_OK;
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (done.load(std::memory_order_relaxed) & bit) return IPDE_OK;
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)bytes));
    done.fetch_or(bit, std::memory_order_relaxed);
    return IPDE_OK;
}

template <int NY>
int launch_rows(ipde_ctx* ctx, const Fft2dPlan& p, bool forward, const double* f, cd* W, double* out) {
    constexpr int H = NY / 2, T = Cfg<H>::T, RPW = 256 / T;
    const size_t lds = (size_t)RPW * lds_slots<H>() * sizeof(cd);
    const dim3 grid((unsigned)(p.nx / RPW));
    static std::atomic<unsigned long long> done_f{0}, done_b{0};
    if (forward) {
        IPDE_TRY(allow_lds(ctx, row_r2c_kernel<NY>, lds, done_f));
        hipLaunchKernelGGL(row_r2c_kernel<NY>, grid, dim3(256), lds, ctx->stream, f, W,
                           (const cd*)p.tw_h, (const cd*)p.tw_ny);
    } else {
        IPDE_TRY(allow_lds(ctx, row_c2r_kernel<NY>, lds, done_b));
        hipLaunchKernelGGL(row_c2r_kernel<NY>, grid, dim3(256), lds, ctx->stream, (const cd*)W,
                           out, (const cd*)p.tw_h, (const cd*)p.tw_ny);
    }
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// columns per workgroup: 4 (64-byte row segments); 2 at nx = 4096, where four 4096-point
// columns would need 1024 threads at <= 128 VGPRs
template <int NX>
constexpr int cols_per_block() {
    return NX >= 4096 ? 2 : 4;
}

template <int NX, int SYM, int MODE>
int launch_cols_t(ipde_ctx* ctx, const Fft2dPlan& p, cd* W, double k2h, double scale, cd* spec_out,
                  int64_t ncols) {
    constexpr int C = cols_per_block<NX>(), T = Cfg<NX>::T;
    const size_t lds = (size_t)C * (lds_slots<NX>() + 4) * sizeof(cd);
    // (ncols < pitch: only the leading columns are transformed, the others are known zeros)
    const int nblocks = (int)((ncols + C - 1) / C);
    const unsigned grid = (unsigned)(((nblocks + 7) / 8) * 8);
    auto k = col_kernel<NX, C, SYM, MODE>;
    static std::atomic<unsigned long long> done{0};
    IPDE_TRY(allow_lds(ctx, k, lds, done));
    hipLaunchKernelGGL(k, dim3(grid), dim3(C * T), lds, ctx->stream, W, (int)p.pitch, nblocks, (int)p.ny,
                       2.0 * M_PI / (p.nx * p.hx), 2.0 * M_PI / (p.ny * p.hy), k2h, scale,
                       (const cd*)p.tw_x, spec_out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

template <int NX>
int launch_cols(ipde_ctx* ctx, const Fft2dPlan& p, int sym, int mode, cd* W, double k2h, double scale,
                cd* so, int64_t nc) {
    if (mode == 1) return launch_cols_t<NX, FFT2D_SYM_NONE, 1>(ctx, p, W, k2h, scale, so, nc);
    if (mode == 2) return launch_cols_t<NX, FFT2D_SYM_NONE, 2>(ctx, p, W, k2h, scale, so, nc);
    switch (sym) {
        case FFT2D_SYM_POISSON: return launch_cols_t<NX, FFT2D_SYM_POISSON, 0>(ctx, p, W, k2h, scale, so, nc);
        case FFT2D_SYM_MODHELM: return launch_cols_t<NX, FFT2D_SYM_MODHELM, 0>(ctx, p, W, k2h, scale, so, nc);
        case FFT2D_SYM_DX: return launch_cols_t<NX, FFT2D_SYM_DX, 0>(ctx, p, W, k2h, scale, so, nc);
        case FFT2D_SYM_DY: return launch_cols_t<NX, FFT2D_SYM_DY, 0>(ctx, p, W, k2h, scale, so, nc);
        default: return launch_cols_t<NX, FFT2D_SYM_NONE, 0>(ctx, p, W, k2h, scale, so, nc);
    }
}

}  // namespace

bool fft2d_supported(int64_t nx, int64_t ny) {
    const bool okx = nx == 512 || nx == 1024 || nx == 2048 || nx == 4096;
    const bool oky = ny == 1024 || ny == 2048 || ny == 4096 || ny == 8192;
    return okx && oky;
}

static int upload_twiddles(ipde_ctx* ctx, void** d, int64_t n, int64_t count) {
    std::vector<double> h(2 * (size_t)count);
    for (int64_t m = 0; m < count; ++m) {
        long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
        h[2 * m] = (double)cosl(a);
        h[2 * m + 1] = (double)sinl(a);
    }
    IPDE_HIP_CHECK(ctx, hipMalloc(d, h.size() * sizeof(double)));
    IPDE_HIP_CHECK(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return IPDE_OK;
}

int fft2d_plan_init(ipde_ctx* ctx, Fft2dPlan& p, int64_t nx, int64_t ny, double hx, double hy) {
    p.nx = nx;
    p.ny = ny;
    p.hx = hx;
    p.hy = hy;
    p.pitch = ny / 2;   // packed half spectrum: the Nyquist entry rides in Im W[.][0]
    IPDE_TRY(upload_twiddles(ctx, &p.tw_x, nx, nx));
    IPDE_TRY(upload_twiddles(ctx, &p.tw_h, ny / 2, ny / 2));
    IPDE_TRY(upload_twiddles(ctx, &p.tw_ny, ny, ny / 2));
    for (auto& w : p.W) {
        const size_t bytes = (size_t)nx * p.pitch * 2 * sizeof(double);
        IPDE_HIP_CHECK(ctx, hipMalloc(&w, bytes));
    }
    p.ready = true;
    return IPDE_OK;
}

void fft2d_plan_free(Fft2dPlan& p) {
    for (void* q : {p.tw_x, p.tw_h, p.tw_ny})
        if (q) (void)hipFree(q);
    for (auto& w : p.W)
        if (w) (void)hipFree(w);
    p = Fft2dPlan();
}

int fft2d_rows_forward(ipde_ctx* ctx, const Fft2dPlan& p, const double* f, int slot) {
    cd* W = (cd*)p.W[slot];
    switch (p.ny) {
        case 1024: return launch_rows<1024>(ctx, p, true, f, W, nullptr);
        case 2048: return launch_rows<2048>(ctx, p, true, f, W, nullptr);
        case 4096: return launch_rows<4096>(ctx, p, true, f, W, nullptr);
        case 8192: return launch_rows<8192>(ctx, p, true, f, W, nullptr);
    }
    return IPDE_ERR_INVALID;
}

int fft2d_rows_inverse(ipde_ctx* ctx, const Fft2dPlan& p, int slot, double* out) {
    cd* W = (cd*)p.W[slot];
    switch (p.ny) {
        case 1024: return launch_rows<1024>(ctx, p, false, nullptr, W, out);
        case 2048: return launch_rows<2048>(ctx, p, false, nullptr, W, out);
        case 4096: return launch_rows<4096>(ctx, p, false, nullptr, W, out);
        case 8192: return launch_rows<8192>(ctx, p, false, nullptr, W, out);
    }
    return IPDE_ERR_INVALID;
}

int fft2d_cols(ipde_ctx* ctx, const Fft2dPlan& p, int slot, int sym, int mode, double k2h, double scale,
               int spec_slot, int64_t ncols) {
    cd* W = (cd*)p.W[slot];
    cd* so = spec_slot >= 0 ? (cd*)p.W[spec_slot] : nullptr;
    const int64_t nc = ncols > 0 ? ncols : p.pitch;
    switch (p.nx) {
        case 512: return launch_cols<512>(ctx, p, sym, mode, W, k2h, scale, so, nc);
        case 1024: return launch_cols<1024>(ctx, p, sym, mode, W, k2h, scale, so, nc);
        case 2048: return launch_cols<2048>(ctx, p, sym, mode, W, k2h, scale, so, nc);
        case 4096: return launch_cols<4096>(ctx, p, sym, mode, W, k2h, scale, so, nc);
    }
    return IPDE_ERR_INVALID;
}

// u = ifft2(fft2(f) * S).real for the analytic scalar symbols: rows, fused columns, rows.
// keep_spectrum: W[1] receives fft2(f) * S * 2 / (nx ny), packed, for fft2d-based interpolation.
int fft2d_scalar_solve(ipde_ctx* ctx, const Fft2dPlan& p, int sym, double k2h, const double* f,
                       double* u, bool keep_spectrum) {
    IPDE_TRY(fft2d_rows_forward(ctx, p, f, 0));
    // row_c2r returns half of the unnormalised inverse: 2 / (nx ny) in all
    IPDE_TRY(fft2d_cols(ctx, p, 0, sym, 0, k2h, 2.0 / ((double)p.nx * (double)p.ny),
                        keep_spectrum ? 1 : -1, 0));
    return fft2d_rows_inverse(ctx, p, 0, u);
}

// (uc, vc, pc) of the Stokes grid solve: two forward transforms, the symbols, three inverse.
int fft2d_stokes_solve(ipde_ctx* ctx, const Fft2dPlan& p, const double* fu, const double* fv, double* u,
                       double* v, double* pr) {
    IPDE_TRY(fft2d_rows_forward(ctx, p, fu, 0));
    IPDE_TRY(fft2d_rows_forward(ctx, p, fv, 1));
    IPDE_TRY(fft2d_cols(ctx, p, 0, FFT2D_SYM_NONE, 1, 0.0, 1.0));
    IPDE_TRY(fft2d_cols(ctx, p, 1, FFT2D_SYM_NONE, 1, 0.0, 1.0));
    const int64_t n = p.nx * (p.ny / 2);
    hipLaunchKernelGGL(stokes_packed_symbol_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
                       ctx->stream, (cd*)p.W[0], (cd*)p.W[1], (cd*)p.W[2], (int)p.nx, (int)p.ny,
                       2.0 * M_PI / (p.nx * p.hx), 2.0 * M_PI / (p.ny * p.hy),
                       2.0 / ((double)p.nx * (double)p.ny));
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    for (int k = 0; k < 3; ++k) IPDE_TRY(fft2d_cols(ctx, p, k, FFT2D_SYM_NONE, 2, 0.0, 1.0));
    IPDE_TRY(fft2d_rows_inverse(ctx, p, 0, u));
    IPDE_TRY(fft2d_rows_inverse(ctx, p, 1, v));
    return fft2d_rows_inverse(ctx, p, 2, pr);
}
