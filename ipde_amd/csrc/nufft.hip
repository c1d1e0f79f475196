// Grid -> scattered points: u, du/dx, du/dy of the periodic grid solution at the interface
// nodes, from the spectrum the grid solve has just produced (reference
// ipde/solvers/multi_boundary/scalar.py:80-88, three finufft type-2 transforms there).
//
// A type-2 non-uniform FFT built on the hand-written pipeline of fft2d.hip:
//   pad     : the packed half spectrum (nx, ny/2) of fft2(f) * symbol, times the field's
//             multiplier (1, i kx, i ky), divided by the transform of the window, goes into
//             the (2 nx, 2 ny) oversampled spectrum; the two Nyquist lines are split half /
//             half between +N/2 and -N/2 and the corner mode goes to the (+,+)/(-,-) pair,
//             which is what `.real` of the complex sum with fft-ordered wavenumbers amounts to;
//   inverse : column pass over the ny/2 + 1 non-zero columns only, then rows c2r, 4096^2 for a
//             2048^2 grid;
//   gather  : one wavefront per point, a w x w patch of the fine grid against the window
//             "exponential of semicircle" exp(beta (sqrt(1 - z^2) - 1)), beta = 2.30 w (the
//             FINUFFT window: its transform is computed by Gauss-Legendre quadrature at set-up).
// O(n^2 log n) instead of the O(N_b n^2) complex GEMM it replaces (ipde_amd/interp.py keeps the
// GEMM as the checker and for grids beyond 2048 x 4096 that are not powers of two).  Three
// variants share the window and the gather: power-of-two grids feed the packed spectrum of the
// fft2d grid solve into a 2x fine grid; 4096-point sides use four half-cell-shifted coarse
// transforms instead of an 8192-point one; every other size takes rocFFT's half spectrum and a
// fine grid of the next power of two >= 2 n (oversampling 2 .. 4, window shape to match).  Derivatives are separate
// transforms of i k F: differentiating the window instead would amplify the aliasing error by
// N_fine / k.
// Roofline: HBM (the fine-grid passes); the gather touches 3 w^2 doubles per point.
//
// BAND form (round 4; context option "interp_band", default 1).  The points sit in a band around a few curves,
// so a whole fine grid per field is transformed to read w^2 values per point of it.  Instead: the oversampled
// transform along x ONLY — for the kept columns ky = 0 .. ny/2, two half-cell-shifted length-nx column passes
// (power-of-two grids, packed spectra; no zeros moved) or one zero-padded length-nfx pass (other sizes) — gives
// g[x_fine][ky], and a point's value is  sum over its w fine rows of psi_x * Re sum_ky eps_ky g[row][ky] e^{i ky y}:
// the sum along y is EXACT (no window, no oversampling in y, any ny).  The points are bucketed by their first
// fine row on the device (one small kernel), a workgroup takes a fine row with up to 16 of the points whose
// windows cover it, holds the row in registers and forms the dense sums with phases e^{2 pi i k f} =
// table[k f * 256] x (short Taylor polynomial), k f reduced exactly (fma), advanced by recurrence over k += 256.
// Every (point, row) pair is written once and the w partials of a point are added in order: results are
// reproducible bit for bit.  Traffic per field at 4096^2 x 8192 points: 134 MB of spectrum in, 2 x 269 MB
// through the column pass, 269 MB read by the gather — against twelve full 4096^2 inverse transforms.
#include "ipde_common.h"
#include "fft2d.h"
#include "fft_core.h"
#include <cmath>

namespace {

struct cd {
    double x, y;
};

// one output field = sum of up to three terms coef * D^der [spectrum src], der 0: value, 1: d/dx,
// 2: d/dy (e.g. the Stokes stress component 2 du/dx - p)
struct Combo {
    int n;
    const cd* src[3];
    int der[3];
    double coef[3];
};

// Gauss-Legendre nodes / weights on [-1, 1] (Newton on P_n)
void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
    x.resize(n);
    w.resize(n);
    for (int i = 0; i < n; ++i) {
        long double z = cosl(3.14159265358979323846264338327950288L * (i + 0.75L) / (n + 0.5L));
        long double pp = 1;
        for (int it = 0; it < 100; ++it) {
            long double p1 = 1, p2 = 0;
            for (int j = 1; j <= n; ++j) {
                long double p3 = p2;
                p2 = p1;
                p1 = ((2 * j - 1) * z * p2 - (j - 1) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1);
            long double dz = p1 / pp;
            z -= dz;
            if (fabsl(dz) < 1e-19L) break;
        }
        x[i] = (double)z;
        w[i] = (double)(2 / ((1 - z * z) * pp * pp));
    }
}

// h_f / psihat(k): psi(x) = phi(x / a), phi(z) = exp(beta (sqrt(1 - z^2) - 1)) on |z| <= 1,
// a = w h_f / 2; psihat(k) = a int_{-1}^{1} phi(z) cos(k a z) dz.  Index units: the box is
// [0, 2 pi), h_f = 2 pi / nf, k integer.
void window_factors(int64_t nf, int w, double beta, int64_t nk, std::vector<double>& r) {
    std::vector<double> gx, gw;
    gauss_legendre(160, gx, gw);
    const long double hf = 2.0L * 3.14159265358979323846264338327950288L / (long double)nf;
    const long double a = 0.5L * w * hf;
    r.resize(nk);
    for (int64_t k = 0; k < nk; ++k) {
        long double s = 0;
        for (size_t q = 0; q < gx.size(); ++q) {
            long double z = gx[q];
            s += (long double)gw[q] * expl((long double)beta * (sqrtl(1 - z * z) - 1)) * cosl((long double)k * a * z);
        }
        r[k] = (double)(hf / (a * s));
    }
}

// Packed coarse spectra -> packed fine half spectrum D (2 nx, ny) [columns 0 .. ny/2 written, all
// 2 nx rows] of one output field (a Combo), one thread per (fine row, column).
__global__ __launch_bounds__(256) void nufft_pad_kernel(Combo cb, cd* __restrict__ D, int nx, int ny,
                                                        const double* __restrict__ rx,
                                                        const double* __restrict__ ry, double dkx,
                                                        double dky) {
    const int H = ny / 2;             // coarse packed width
    const int ncol = H + 1;           // fine columns that receive data: ky = 0 .. ny/2
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)2 * nx * ncol) return;
    const int fi = (int)(idx / ncol), j = (int)(idx - (int64_t)fi * ncol);
    // fine row fi carries kx = fi (fi <= nx/2) or fi - 2 nx (fi >= 3 nx / 2); nothing between
    int kx;
    if (fi <= nx / 2)
        kx = fi;
    else if (fi >= 2 * nx - nx / 2)
        kx = fi - 2 * nx;
    else {
        D[(int64_t)fi * ny + j] = cd{0.0, 0.0};
        return;
    }
    const int ci = (kx + nx) % nx;    // coarse row (both +nx/2 and -nx/2 read the Nyquist row nx/2)
    double wgt = rx[kx < 0 ? -kx : kx] * ry[j];
    const bool nyqx = 2 * (kx < 0 ? -kx : kx) == nx, nyqy = (j == H);
    if (nyqx && nyqy) {
        // the corner mode: Re(F e^{-i(Nx/2 x + Ny/2 y)}) = F cos(Nx/2 x + Ny/2 y) is the pair
        // (+,+) / (-,-) alone — half at (+Nx/2, +Ny/2), nothing at (-Nx/2, +Ny/2)
        wgt = kx > 0 ? 0.5 * wgt : 0.0;
    } else if (nyqx || nyqy) {
        wgt *= 0.5;   // a Nyquist line: half at +N/2, half at -N/2 (cos(N/2 x) times the rest)
    }
    cd acc{0.0, 0.0};
    for (int t = 0; t < cb.n; ++t) {
        const cd* S = cb.src[t];
        cd v;
        if (j == 0 || j == H) {
            // unpack column 0: G = U0 + i UH, U0 = (G + conj Gm)/2, UH = (G - conj Gm)/2i
            cd g = S[(int64_t)ci * H], gm = S[(int64_t)((nx - ci) % nx) * H];
            v = (j == 0) ? cd{0.5 * (g.x + gm.x), 0.5 * (g.y - gm.y)}
                         : cd{0.5 * (g.y + gm.y), -0.5 * (g.x - gm.x)};
        } else {
            v = S[(int64_t)ci * H + j];
        }
        const double c = wgt * cb.coef[t];
        v.x *= c;
        v.y *= c;
        if (cb.der[t] == 1) v = cd{-v.y * (kx * dkx), v.x * (kx * dkx)};
        if (cb.der[t] == 2) v = cd{-v.y * (j * dky), v.x * (j * dky)};
        acc.x += v.x;
        acc.y += v.y;
    }
    // fine column 0 is packed too: its imaginary part is the fine Nyquist column, which is zero;
    // U0 is Hermitian in kx, so storing it as is keeps the packing consistent
    D[(int64_t)fi * ny + j] = acc;
}


// General-size variant of the pad kernel: the coarse grid (nx, ny) is arbitrary (the spectra
// are rocFFT's unpacked half spectra, (nx, ny/2 + 1)), the fine grid (nfx, nfy) is the next
// power of two >= twice the coarse size (oversampling between 2 and 4).  Same Nyquist rules;
// an odd size has no Nyquist line.
__global__ __launch_bounds__(256) void nufft_pad_general_kernel(Combo cb, cd* __restrict__ D, int nx,
                                                                int ny, int nfx, int nfy,
                                                                const double* __restrict__ rx,
                                                                const double* __restrict__ ry,
                                                                double dkx, double dky) {
    const int nyh = ny / 2 + 1, Hf = nfy / 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)nfx * nyh) return;
    const int fi = (int)(idx / nyh), j = (int)(idx - (int64_t)fi * nyh);
    const int kmaxp = nx / 2;                 // even: the split Nyquist; odd: (nx - 1) / 2
    int kx;
    if (fi <= kmaxp)
        kx = fi;
    else if (fi >= nfx - kmaxp)
        kx = fi - nfx;
    else {
        D[(int64_t)fi * Hf + j] = cd{0.0, 0.0};
        return;
    }
    const int akx = kx < 0 ? -kx : kx;
    const bool nyqx = (nx % 2 == 0) && 2 * akx == nx, nyqy = (ny % 2 == 0) && 2 * j == ny;
    const int ci = ((kx % nx) + nx) % nx;
    double wgt = rx[akx] * ry[j];
    if (nyqx && nyqy)
        wgt = kx > 0 ? 0.5 * wgt : 0.0;
    else if (nyqx || nyqy)
        wgt *= 0.5;
    cd acc{0.0, 0.0};
    for (int t = 0; t < cb.n; ++t) {
        cd v = cb.src[t][(int64_t)ci * nyh + j];
        const double c = wgt * cb.coef[t];
        v.x *= c;
        v.y *= c;
        if (cb.der[t] == 1) v = cd{-v.y * (kx * dkx), v.x * (kx * dkx)};
        if (cb.der[t] == 2) v = cd{-v.y * (j * dky), v.x * (j * dky)};
        acc.x += v.x;
        acc.y += v.y;
    }
    D[(int64_t)fi * Hf + j] = acc;
}

// --- the same fine-grid samples WITHOUT a fine-grid transform ---------------------------------
// A 2x oversampled inverse transform of a zero-padded spectrum is four coarse-size inverse
// transforms: fine sample (2m + a, 2n + b) = coarse sample (m, n) of the field shifted by
// (a h/2, b h/2), i.e. of the spectrum times e^{i pi (a kx / nx + b ky / ny)}.  Used when the
// fine size is beyond the FFT kernels (coarse 4096: BASELINE configs[3], [4]): same window, same
// gather, no 8192-point transforms and no zeros moved through HBM.
//
// D (nx, ny/2) packed <- S packed, for shift (a, b) and field (0: value, 1: d/dx, 2: d/dy).
// The Nyquist lines are cosines: cos(N/2 x) sampled at the half-shifted points vanishes and its
// derivative survives only there; the corner mode F cos(Nx/2 x + Ny/2 y) (see the pad kernel)
// depends on a + b.
__device__ __forceinline__ cd cis_pi(double t) {   // e^{i pi t}
    double sn, cs;
    sincospi(t, &sn, &cs);
    return cd{cs, sn};
}
__device__ __forceinline__ cd cmulz(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

__global__ __launch_bounds__(256) void nufft_shift_kernel(Combo cb, cd* __restrict__ D, int nx, int ny,
                                                          const double* __restrict__ rx,
                                                          const double* __restrict__ ry, int a, int b,
                                                          double dkx, double dky) {
    const int H = ny / 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)nx * H) return;
    const int i = (int)(idx / H), j = (int)(idx - (int64_t)i * H);
    const int kx = (i < nx / 2) ? i : i - nx;        // (row nx/2: kx = -nx/2, handled as a cosine)
    const bool nyqx = (2 * i == nx);
    const double wx = rx[kx < 0 ? -kx : kx];
    // factor of a 1-D mode along x: interior e^{i pi a kx / nx} (times i kx for d/dx); the Nyquist
    // cosine: value 1 / 0 at shift 0 / 1, derivative 0 / -(nx/2) dkx
    auto xfac = [&](int der) -> cd {
        if (nyqx) {
            if (!der) return cd{a ? 0.0 : wx, 0.0};
            return cd{a ? -0.5 * nx * dkx * wx : 0.0, 0.0};
        }
        cd ph = a ? cis_pi((double)kx / nx) : cd{1.0, 0.0};
        ph.x *= wx;
        ph.y *= wx;
        return der ? cd{-ph.y * (kx * dkx), ph.x * (kx * dkx)} : ph;
    };
    auto yfac = [&](int jj, int der) -> cd {         // jj in 0 .. H; jj == H: the Nyquist cosine
        const double wy = ry[jj];
        if (jj == H) {
            if (!der) return cd{b ? 0.0 : wy, 0.0};
            return cd{b ? -0.5 * ny * dky * wy : 0.0, 0.0};
        }
        cd ph = b ? cis_pi((double)jj / ny) : cd{1.0, 0.0};
        ph.x *= wy;
        ph.y *= wy;
        return der ? cd{-ph.y * (jj * dky), ph.x * (jj * dky)} : ph;
    };
    cd acc0{0.0, 0.0}, acch{0.0, 0.0};
    for (int t = 0; t < cb.n; ++t) {
        const cd* S = cb.src[t];
        const int dx = (cb.der[t] == 1), dy = (cb.der[t] == 2);
        const double c = cb.coef[t];
        if (j > 0) {
            cd v = cmulz(cmulz(S[idx], xfac(dx)), yfac(j, dy));
            acc0.x += c * v.x;
            acc0.y += c * v.y;
            continue;
        }
        // column 0 carries ky = 0 (U0) and the Nyquist column (UH): unpack with the mirrored row,
        // treat both, pack again (both stay Hermitian in kx: the factors are e^{i odd(kx)} or real)
        cd g = S[(int64_t)i * H], gm = S[(int64_t)((nx - i) % nx) * H];
        cd u0 = cd{0.5 * (g.x + gm.x), 0.5 * (g.y - gm.y)};
        cd uh = cd{0.5 * (g.y + gm.y), -0.5 * (g.x - gm.x)};
        u0 = cmulz(cmulz(u0, xfac(dx)), yfac(0, dy));
        if (nyqx) {
            // the corner: F cos(A + B), A = pi m + pi a/2, B = pi n + pi b/2
            const int sft = a + b;
            double f;
            if (cb.der[t] == 0)
                f = sft == 0 ? 1.0 : (sft == 1 ? 0.0 : -1.0);
            else
                f = sft == 1 ? -(cb.der[t] == 1 ? 0.5 * nx * dkx : 0.5 * ny * dky) : 0.0;
            f *= wx * ry[H];
            uh = cd{uh.x * f, uh.y * f};
        } else {
            uh = cmulz(cmulz(uh, xfac(dx)), yfac(H, dy));
        }
        acc0.x += c * u0.x;
        acc0.y += c * u0.y;
        acch.x += c * uh.x;
        acch.y += c * uh.y;
    }
    D[idx] = (j > 0) ? acc0 : cd{acc0.x - acch.y, acc0.y + acch.x};   // U0 + i UH
}

// SUB: the fine grid is stored as four coarse sub-grids [2 a + b][m][n] (fine (2m + a, 2n + b))
template <int W, bool SUB>
__global__ __launch_bounds__(64) void nufft_gather_kernel(const double* __restrict__ g0,
                                                          const double* __restrict__ g1,
                                                          const double* __restrict__ g2, int nf,
                                                          int nfx, int nfy,
                                                          const double* __restrict__ px,
                                                          const double* __restrict__ py, int64_t np,
                                                          double betax, double betay,
                                                          double* __restrict__ out) {
    const int64_t p = blockIdx.x;
    const int lane = threadIdx.x;
    const double TWO_PI = 6.283185307179586476925286766559;
    const double hfx = TWO_PI / nfx, hfy = TWO_PI / nfy;
    double x = px[p], y = py[p];
    x -= TWO_PI * floor(x / TWO_PI);
    y -= TWO_PI * floor(y / TWO_PI);
    // first node of the patch: the W nodes nearest to the point
    const int ix0 = (int)ceil(x / hfx - 0.5 * W), iy0 = (int)ceil(y / hfy - 0.5 * W);
    auto psi = [&](double z, double beta) {
        double q = 1.0 - z * z;
        return q > 0.0 ? exp(beta * (sqrt(q) - 1.0)) : 0.0;
    };
    // lane -> column b = lane % 16 (W <= 16: one column per lane of a 16-lane row group),
    // rows a = lane / 16 + 4 r
    constexpr int WB = 16;
    static_assert(W <= WB, "patch wider than a 16-lane group");
    const int b = lane % WB, a0 = lane / WB;
    const double wy = (b < W) ? psi((y - (iy0 + b) * hfy) / (0.5 * W * hfy), betay) : 0.0;
    const int jy = ((iy0 + b) % nfy + nfy) % nfy;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < (W + 3) / 4; ++r) {
        const int a = a0 + 4 * r;
        if (a < W && b < W) {
            const double wx = psi((x - (ix0 + a) * hfx) / (0.5 * W * hfx), betax);
            const int jx = ((ix0 + a) % nfx + nfx) % nfx;
            const int64_t o = SUB ? ((int64_t)(2 * (jx & 1) + (jy & 1)) * (nfx / 2) + (jx >> 1)) * (nfy / 2) + (jy >> 1)
                                  : (int64_t)jx * nfy + jy;
            const double ww = wx * wy;
            s0 = fma(ww, g0[o], s0);
            if (nf > 1) s1 = fma(ww, g1[o], s1);
            if (nf > 2) s2 = fma(ww, g2[o], s2);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s0 += __shfl_xor(s0, m);
        s1 += __shfl_xor(s1, m);
        s2 += __shfl_xor(s2, m);
    }
    if (lane == 0) {
        out[p] = s0;
        if (nf > 1) out[np + p] = s1;
        if (nf > 2) out[2 * np + p] = s2;
    }
}


// ---- the band form ----------------------------------------------------------------------------------
// The pad kernel's value at fine wavenumber kx (|kx| <= nx/2), column j (0 .. ny/2) of a Combo over PACKED coarse
// spectra, with the y window factor 1: the same Nyquist rules (see nufft_pad_kernel).
__device__ __noinline__ cd band_value(const Combo& cb, int kx, int j, int nx, int H, const double* __restrict__ rx,
                                         double dkx, double dky) {
    const int akx = kx < 0 ? -kx : kx;
    const int ci = (kx + nx) % nx;
    double wgt = rx[akx];
    const bool nyqx = 2 * akx == nx, nyqy = (j == H);
    if (nyqx && nyqy)
        wgt = kx > 0 ? 0.5 * wgt : 0.0;
    else if (nyqx || nyqy)
        wgt *= 0.5;
    cd acc{0.0, 0.0};
    for (int t = 0; t < cb.n; ++t) {
        const cd* S = cb.src[t];
        cd v;
        if (j == 0 || j == H) {
            cd g = S[(int64_t)ci * H], gm = S[(int64_t)((nx - ci) % nx) * H];
            v = (j == 0) ? cd{0.5 * (g.x + gm.x), 0.5 * (g.y - gm.y)} : cd{0.5 * (g.y + gm.y), -0.5 * (g.x - gm.x)};
        } else {
            v = S[(int64_t)ci * H + j];
        }
        const double c = wgt * cb.coef[t];
        v.x *= c;
        v.y *= c;
        if (cb.der[t] == 1) v = cd{-v.y * (kx * dkx), v.x * (kx * dkx)};
        if (cb.der[t] == 2) v = cd{-v.y * (j * dky), v.x * (j * dky)};
        acc.x += v.x;
        acc.y += v.y;
    }
    return acc;
}

// The 2x oversampled inverse transform along x of columns j0 .. j0 + C - 1, straight from the packed coarse
// spectra: D, the fine-row-interleaved layout (2 nx rows of `pitch` complex, row 2 i + a = sample (i + a / 2) h_x),
// gets the length-nx inverse transform of value(kx, j) (a = 0) and of value(kx, j) e^{i pi kx / nx} (a = 1; the x
// Nyquist row holds +nx/2 and -nx/2 together: v+ + v- and i (v+ - v-)).  The workgroup is col_kernel's (fft2d.hip):
// thread = (column, t), its values rows t + T q — the transform's own register layout, 64-byte row segments in
// and out, XCD-aware hand-out of the column blocks.  Nothing padded or shifted ever goes through HBM.
template <int NX, int C>
__global__ __launch_bounds__(C* fftcore::Cfg<NX>::T) void band_col_kernel(Combo cb, cd* __restrict__ D, int ny, int pitch,
                                                                         int nblocks, const double* __restrict__ rx,
                                                                         double dkx, double dky,
                                                                         const fftcore::cd* __restrict__ tw_x) {
    using namespace fftcore;
    using G = Cfg<NX>;
    constexpr int T = G::T, P = G::P, NPC = lds_slots<NX>() + 4;
    extern __shared__ double2 lds_raw[];
    fftcore::cd* lds = (fftcore::cd*)lds_raw;
    const int per = (nblocks + 7) / 8;
    const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (blk >= nblocks) return;
    const int tid = threadIdx.x;
    const int col = tid % C, t = tid / C;
    const int H = ny / 2;
    const int j = blk * C + col;
    const bool live = j < H;                  // (column H rides in column 0, see below: H columns, H / C whole blocks)
    const PassTw<NX> pw = load_pass_twiddles<NX>(t, tw_x);
    fftcore::cd* buf = lds + col * NPC;
    // e^{i pi (t + T q) / nx} = e^{i pi t / nx} (e^{i pi T / nx})^q
    double sn, cs;
    sincospi((double)t / NX, &sn, &cs);
    const fftcore::cd e0{cs, sn};
    sincospi((double)T / NX, &sn, &cs);
    const fftcore::cd est{cs, sn};
    // Columns 0 and ny/2 are packed together in the coarse spectra (unpacked with the mirrored row), and they leave
    // this kernel packed together again: both are spectra of REAL sequences along x (the combos' terms are fields
    // and their x derivatives with real coefficients; y derivatives are left to the gather), so their transforms
    // D_0 and D_H are real and ONE transform of value(kx, 0) + i value(kx, H) carries them as Re and Im of column 0
    // — the gather reads them back as such.  H columns instead of H + 1: at 2048^2 that is 256 workgroups of four
    // columns on 256 CUs where 257 took two rounds (73 -> 40 us), at 4096^2 1024 instead of 1025 (five rounds -> four).
    // The values go through the transform's LDS slots in a ROLLED loop (eight rows at a time: eight loads per term in
    // flight; the packed column loads each row's mirror as well) — unrolled sixteen times next to the transform's
    // own registers the loader spilled 1 KB per lane.  (A first version sent the packed column through band_value
    // row by row: its workgroup then took 110 us where the others take 40.)
    // One mode of column H is not Hermitian: the corner (kx, ky) = (+nx/2, ny/2) carries its whole weight on kx > 0
    // (the convention of the dense sums this interpolation reproduces).  It stays out of the packed transform; its
    // own transform is known in closed form — v_c e^{i pi r / 2} on fine row r — and is stored in the spare slot
    // D[r][H], which the gather adds to D_H.
    const bool special = live && j == 0;
    auto packed = [&](int kx) {
        const ::cd v0 = band_value(cb, kx, 0, NX, H, rx, dkx, dky);
        ::cd vh{0.0, 0.0};
        if (2 * kx != NX && 2 * kx != -NX) vh = band_value(cb, kx, H, NX, H, rx, dkx, dky);
        return ::cd{v0.x - vh.y, v0.y + vh.x};
    };
    ::cd vcorner{0.0, 0.0};
    if (special) vcorner = band_value(cb, NX / 2, H, NX, H, rx, dkx, dky);
#pragma unroll 1
    for (int a = 0; a < 2; ++a) {
        if (live) {
            fftcore::cd ph = e0;
            constexpr int LB = P >= 8 ? 8 : P;      // rows per batch of the loader
            fftcore::cd estb = est;                 // est^LB
#pragma unroll
            for (int m = 1; m < LB; m <<= 1) estb = cmul(estb, estb);
#pragma unroll 1
            for (int q0 = 0; q0 < P; q0 += LB) {
                fftcore::cd acc[LB];
#pragma unroll
                for (int q = 0; q < LB; ++q) acc[q] = fftcore::cd{0.0, 0.0};
                for (int tm = 0; tm < cb.n; ++tm) {
                    const fftcore::cd* S = (const fftcore::cd*)cb.src[tm] + j;
                    const int der = cb.der[tm];
                    const double cf = cb.coef[tm];
                    fftcore::cd raw[LB];
                    double wr[LB];
#pragma unroll
                    for (int q = 0; q < LB; ++q) {
                        const int i = t + T * (q0 + q);
                        const int kx = (i < NX / 2) ? i : i - NX;
                        raw[q] = S[(int64_t)i * H];
                        wr[q] = rx[kx < 0 ? -kx : kx];
                    }
                    if (special) {
                        // the packed column: value(kx, 0) + i value(kx, H) = w (X_0 + i X_H / 2) — column H carries
                        // its Nyquist half — with X_0 = (P + conj P-) / 2, i X_H = (P - conj P-) / 2 of the packed entry
                        // P(kx) and its mirror P- = P(-kx):  3/4 P + 1/4 conj P-
#pragma unroll
                        for (int q = 0; q < LB; ++q) {
                            const int i = t + T * (q0 + q);
                            const fftcore::cd m = S[(int64_t)((NX - i) & (NX - 1)) * H];
                            raw[q] = fftcore::cd{0.75 * raw[q].x + 0.25 * m.x, 0.75 * raw[q].y - 0.25 * m.y};
                        }
                    }
#pragma unroll
                    for (int q = 0; q < LB; ++q) {
                        const int i = t + T * (q0 + q);
                        const int kx = (i < NX / 2) ? i : i - NX;
                        const double w = wr[q] * cf;
                        fftcore::cd u{raw[q].x * w, raw[q].y * w};
                        if (der == 1) u = fftcore::cd{-u.y * (kx * dkx), u.x * (kx * dkx)};
                        if (der == 2) u = fftcore::cd{-u.y * (j * dky), u.x * (j * dky)};
                        acc[q].x += u.x;
                        acc[q].y += u.y;
                    }
                }
                fftcore::cd pq = ph;
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    const int i = t + T * (q0 + q);
                    fftcore::cd val = acc[q];
                    if (a == 1) {
                        val = cmul(val, pq);
                        if (i > NX / 2) val = fftcore::cd{-val.x, -val.y};      // e^{i pi (i - nx) / nx} = -e^{i pi i / nx}
                    }
                    buf[padpos(i)] = val;
                    pq = cmul(pq, est);
                }
                ph = cmul(ph, estb);
            }
            // the x Nyquist row (i = nx/2, thread t = 0) holds +nx/2 and -nx/2 together
            if (t == 0) {
                const ::cd vp = special ? packed(NX / 2) : band_value(cb, NX / 2, j, NX, H, rx, dkx, dky);
                const ::cd vm = special ? packed(-(NX / 2)) : band_value(cb, -(NX / 2), j, NX, H, rx, dkx, dky);
                buf[padpos(NX / 2)] = a == 0 ? fftcore::cd{vp.x + vm.x, vp.y + vm.y}
                                             : fftcore::cd{-(vp.y - vm.y), vp.x - vm.x};
            }
        } else {
#pragma unroll 1
            for (int q = 0; q < P; ++q) buf[padpos(t + T * q)] = fftcore::cd{0.0, 0.0};
        }
        fftcore::cd v[P];
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = buf[padpos(t + T * q)];      // (its own slots: nobody else's)
        __syncthreads();                                                  // (before anybody's exchange writes)
        fft_regs<NX, +1, false>(v, t, pw, buf);
        if (live) {
            fftcore::cd* base = (fftcore::cd*)D + j;
#pragma unroll
            for (int q = 0; q < P; ++q) base[(int64_t)(2 * (t + T * q) + a) * pitch] = v[q];
            if (special) {                               // the corner mode's transform: v_c i^r on fine row r
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const int r = 2 * (t + T * q) + a;
                    const fftcore::cd c = (r & 1) ? fftcore::cd{-vcorner.y, vcorner.x} : fftcore::cd{vcorner.x, vcorner.y};
                    base[(int64_t)r * pitch + H] = (r & 2) ? fftcore::cd{-c.x, -c.y} : c;
                }
            }
        }
        __syncthreads();
    }
}

// Bucket the points by the first fine row of their x window (one workgroup; a few 10^4 points): r0[j] = first
// row + 16 (>= 0 also for windows that wrap below 0), perm = the points in bucket order, start = the buckets'
// offsets (nrows_ext + 1 entries).  The order INSIDE a bucket depends on the atomics; nothing downstream does.
__global__ __launch_bounds__(1024) void band_sort_kernel(const double* __restrict__ px, int64_t np, int nfx, int W,
                                                         int nrows_ext, int* __restrict__ cnt,
                                                         int* __restrict__ start, int* __restrict__ perm,
                                                         int* __restrict__ r0v, int* __restrict__ rank) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const double TWO_PI = 6.283185307179586476925286766559;
    const double hfx = TWO_PI / nfx;
    for (int i = tid; i <= nrows_ext; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int64_t j = tid; j < np; j += 1024) {
        double x = px[j];
        x -= TWO_PI * floor(x / TWO_PI);
        const int r0 = (int)ceil(x / hfx - 0.5 * W) + 16;
        r0v[j] = r0;
        rank[j] = atomicAdd(&cnt[r0], 1);
    }
    __syncthreads();
    const int per = (nrows_ext + 1023) / 1024;
    const int a = tid * per;
    int s = 0;
    for (int i = a; i < a + per && i < nrows_ext; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {       // inclusive scan of the 1024 partial sums
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = a; i < a + per && i < nrows_ext; ++i) {
        start[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) start[nrows_ext] = (int)np;
    __syncthreads();
    for (int64_t j = tid; j < np; j += 1024) perm[start[r0v[j]] + rank[j]] = (int)j;
}

// e^{2 pi i k f}, k >= 0 an integer, f in [0, 1): k f = p + e exactly (fma), frac(p) exact, the 8 leading bits
// of the fraction index a table of the 256th roots of unity, the rest (< 1/256 of a turn) goes through
// Taylor polynomials: |error| ~ 2e-16 whatever k.
__device__ __forceinline__ cd unit_kf(int k, double f, const cd* __restrict__ T) {
    const double p = (double)k * f;
    const double e = fma((double)k, f, -p);
    const double s = (p - floor(p)) * 256.0;
    const int idx = (int)s;
    const double z = 6.283185307179586476925286766559 * ((s - (double)idx) * 0.00390625 + e);
    const double z2 = z * z;
    const double c = fma(z2, fma(z2, fma(z2, fma(z2, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
    const double sn = z * fma(z2, fma(z2, fma(z2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
    const cd t = T[idx & 255];
    return cd{t.x * c - t.y * sn, t.x * sn + t.y * c};
}

// What a gather launch produces from its (up to three) transformed arrays: output `out` (a row of the partial
// buffer) = sum over its entries of coef * (plain sum | y-derivative sum) of array `arr`; acc: add to what an
// earlier launch left in the row.
struct BandOut {
    int out, n, acc;
    int arr[2], kind[2];      // kind 0: Re sum_k eps_k g e^{i k y};  1: d/dy = Re sum_k (i k dky) g e^{i k y}
    double coef[2];
};
struct BandRecipe {
    int nout;
    int need_dy[3];
    BandOut o[8];
};

// The gather as a block GEMM on the matrix cores.  For a tile of 16 consecutive fine rows and a chunk of 16 of the
// points whose windows meet it,  P[row][point] = Re sum_k eps_k g[row][k] e^{i k y_point}  is a (16 x ncol) x (ncol x 16)
// product: v_mfma_f64_16x16x4 with A = Re g / Im g of the rows (from LDS, staged 128 columns at a time with coalesced
// loads) and B = cos / -sin of k y — the phases are formed ONCE per (k, point) and serve all sixteen rows (a
// row-by-row gather spends most of its time on them: 181 us against 62 at 2048^2 x 4096), each lane advancing its own point's phase by e^{4 i y}
// and re-seeding it exactly at every staged tile (32 steps: 7e-15).  The y-derivative sums are two more products with
// B = k sin, k cos.  A workgroup = a row tile, its four waves take four point chunks; the epilogue applies the x
// window's weights and writes every (output, point, row) partial to its own slot
// partial[(out np + j) 16 + s], s = the row's place in the point's window; band_reduce_kernel adds the sixteen.
typedef double band_d4 __attribute__((ext_vector_type(4)));
constexpr int BAND_KSPLIT_MAX = 4;                      // column shares of a gather launch (slabs of the partial buffer)

template <int NA>
__global__ __launch_bounds__(256) void band_gather_mfma_kernel(const cd* __restrict__ D0, const cd* __restrict__ D1,
                                                              const cd* __restrict__ D2, BandRecipe rc, int64_t pitch,
                                                              int ncol, int packed0, int nfx, int nrows_ext,
                                                              const double* __restrict__ px,
                                                              const double* __restrict__ py, int64_t np,
                                                              const int* __restrict__ start,
                                                              const int* __restrict__ perm,
                                                              const int* __restrict__ r0v, double beta, double dky,
                                                              const cd* __restrict__ roots,
                                                              double* __restrict__ partial) {
    // CH point chunks per round; the four waves SPLIT THE COLUMNS of every staged tile (wave w: steps w, w + 4, ...) and
    // each carries all CH chunks — whatever the number of chunks of a tile (two, typically) all four SIMDs of the CU
    // work; with a chunk per wave half of them idled and two workgroups on a CU queued on the same two.  The waves'
    // partial products are added through LDS at the end of a round, in wave order.
    constexpr int W = 16, KT = 128, RS = KT + 1;       // (row stride 129 slots: the sixteen rows of an A read hit 16 bank groups)
    constexpr int CH = 2;                               // (four per round for NA <= 2 left single CUs with twice the average work: 165 -> 127 us at 2048^2 x 4096)
    constexpr int NACC = CH * NA * 2;                   // accumulator tiles (4 doubles per lane each)
    __shared__ cd sG[NA][16][RS];
    __shared__ cd T[256];
    __shared__ int s_phys[16];
    __shared__ cd sNyq[NA][16];                         // packed0: the rows' column-ny/2 values (Im of column 0 + corner)
    __shared__ cd sCorner[NA][16];                      // packed0: the corner mode's part of them (slot ncol of the rows)
    static_assert(sizeof(cd) * NA * 16 * RS >= sizeof(double) * 64 * 4 * NACC, "the reduction reuses the staging buffer");
    double* sacc = (double*)&sG[0][0][0];               // [NACC][4][64] after the last tile of a round
    const int R0 = 16 * (int)blockIdx.x;
    const int first = R0 - (W - 1) > 0 ? R0 - (W - 1) : 0;
    const int last = R0 + 16 < nrows_ext ? R0 + 16 : nrows_ext;
    const int lo = start[first], cnt = start[last] - lo;
    if (cnt <= 0) return;                               // (uniform over the workgroup)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, h4 = lane >> 4;
    T[tid] = roots[tid];
    if (tid < 16) s_phys[tid] = ((R0 + tid - 16) % nfx + nfx) % nfx;
    const double TWO_PI = 6.283185307179586476925286766559;
    const double hfx = TWO_PI / nfx;
    const cd* Dp[3] = {D0, D1, D2};
    // (read here, once per workgroup: inside the staging loop each of these sixteen loads was an exposed round trip —
    // the compiler waits for a conditional load where it stands)
    if (packed0 && tid < 16 * NA) {
        const int pr = ((R0 + (tid & 15) - 16) % nfx + nfx) % nfx;
        sCorner[tid >> 4][tid & 15] = Dp[tid >> 4][(int64_t)pr * pitch + ncol];
    }
    const int nchunk = (cnt + 15) / 16;
    // (blockIdx.y strides over the rounds: where the curve runs along a fine row hundreds of points meet one tile)
    for (int round = blockIdx.y; round * CH < nchunk; round += gridDim.y) {
        const int nact = nchunk - round * CH < CH ? nchunk - round * CH : CH;
        // this lane's B column in every chunk of the round: point 16 (CH round + c) + l16
        double fy[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int q = 16 * (round * CH + c) + l16;
            double f = 0.0;
            if (q < cnt) f = py[perm[lo + q]] * (1.0 / TWO_PI);
            fy[c] = f - floor(f);
        }
        band_d4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = band_d4{0.0, 0.0, 0.0, 0.0};
        // the staged tiles are double-buffered through registers: tile t + 1's loads are in flight under tile t's
        // products (eight 16-byte loads per thread and array; waited for at the top of the next pass)
        cd st[NA][8];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int e = tid + 256 * it, i = e / KT, kk = e % KT, k = k0 + kk;
                    st[a][it] = k < ncol ? Dp[a][(int64_t)s_phys[i] * pitch + k] : cd{0.0, 0.0};
                }
        };
        // blockIdx.z takes a share of the staged column tiles (an A/B knob, one share by default: see the launch;
        // the shares' sums go to slabs of their own)
        const int ntile_k = (ncol + KT - 1) / KT;
        const int kt0 = (int)(((int64_t)ntile_k * blockIdx.z) / gridDim.z), kt1 = (int)(((int64_t)ntile_k * (blockIdx.z + 1)) / gridDim.z);
        const int kbeg = kt0 * KT, kend = kt1 * KT < ncol ? kt1 * KT : ncol;
        __syncthreads();                                // (T, s_phys are in place; the previous round's sums are read)
        fetch(kbeg);
        cd st16[CH];                                    // e^{16 i y}: the phase step of a wave's consecutive k
#pragma unroll
        for (int c = 0; c < CH; ++c) st16[c] = c < nact ? unit_kf(16, fy[c], T) : cd{1.0, 0.0};
        for (int k0 = kbeg; k0 < kend; k0 += KT) {
            __syncthreads();                            // (the previous tile has been consumed)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int e = tid + 256 * it, i = e / KT, kk = e % KT;
                    cd v = st[a][it];
                    if (k0 + kk == 0) {
                        // eps_0 = 1/2 (see row_c2r).  packed0: column 0 holds D_0 + i D_H, both real, and slot ncol the
                        // one non-Hermitian mode of column H (band_col_kernel): D_0 goes through the product, D_H —
                        // one column, frequency ncol — is added by the epilogue
                        if (packed0) {
                            const cd cn = sCorner[a][i];                                // (the corner mode's part)
                            sNyq[a][i] = cd{v.y + cn.x, cn.y};
                            v.y = 0.0;
                        }
                        v = cd{0.5 * v.x, 0.5 * v.y};
                    }
                    sG[a][i][kk] = v;
                }
            __syncthreads();
            if (k0 + KT < kend) fetch(k0 + KT);
            // this wave's steps of the tile: k = k0 + 4 s + h4, s = wave, wave + 4, ...; phases re-seeded per tile
            cd ph[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) ph[c] = c < nact ? unit_kf(k0 + 4 * wave + h4, fy[c], T) : cd{1.0, 0.0};
#pragma unroll 2
            for (int it = 0; it < KT / 16; ++it) {
                const int s4 = wave + 4 * it;
                const int kk = 4 * s4 + h4;
                const double kq = (double)(k0 + kk);
                cd gv[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) gv[a] = sG[a][l16][kk];   // A entries: row l16, k = k0 + 4 s + h4
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (c >= nact) continue;            // (uniform: chunks of this round that exist)
                    const double ky = kq * ph[c].y, kx = kq * ph[c].x, ms = -ph[c].y;
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        band_d4& P = acc[(c * NA + a) * 2];
                        band_d4& Dy = acc[(c * NA + a) * 2 + 1];
                        P = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[a].x, ph[c].x, P, 0, 0, 0);
                        if (rc.need_dy[a]) Dy = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[a].x, ky, Dy, 0, 0, 0);
                    }
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        band_d4& P = acc[(c * NA + a) * 2];
                        band_d4& Dy = acc[(c * NA + a) * 2 + 1];
                        P = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[a].y, ms, P, 0, 0, 0);
                        if (rc.need_dy[a]) Dy = __builtin_amdgcn_mfma_f64_16x16x4f64(gv[a].y, kx, Dy, 0, 0, 0);
                    }
                    ph[c] = cmulz(ph[c], st16[c]);
                }
            }
        }
        // the four waves' partial products, added in wave order through the staging buffer
        for (int w = 0; w < 4; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int i = 0; i < NACC; ++i)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        double* slot = sacc + ((size_t)(i * 4 + v) * 64 + lane);
                        *slot = w == 0 ? acc[i][v] : *slot + acc[i][v];
                    }
            }
        }
        __syncthreads();
        // epilogue: wave c takes chunk c; lane (point l16, h4) holds rows h4 + 4 v of its point's column
        if (wave < CH) {
            const int c = wave;
            const int q = 16 * (round * CH + c) + l16;
            if (q < cnt) {
                const int j = perm[lo + q];
                const int r0 = r0v[j];
                double x = px[j];
                x -= TWO_PI * floor(x / TWO_PI);
                cd eH{0.0, 0.0};                        // e^{i ncol y} of this lane's point
                if (packed0) {
                    double f = py[j] * (1.0 / TWO_PI);
                    eH = unit_kf(ncol, f - floor(f), T);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r_ext = R0 + h4 + 4 * v;
                    const int sidx = r_ext - r0;
                    if (sidx < 0 || sidx >= W || r_ext >= nrows_ext) continue;
                    const double zq = (x - (double)(r_ext - 16) * hfx) / (0.5 * W * hfx);
                    const double qq = 1.0 - zq * zq;
                    const double wx = qq > 0.0 ? exp(beta * (sqrt(qq) - 1.0)) : 0.0;
                    for (int o = 0; o < rc.nout; ++o) {
                        const BandOut& bo = rc.o[o];
                        double val = 0.0;
                        for (int e = 0; e < bo.n; ++e) {
                            const int i = (c * NA + bo.arr[e]) * 2 + bo.kind[e];
                            double sum = sacc[(size_t)(i * 4 + v) * 64 + lane];
                            if (packed0 && blockIdx.z == 0) {      // + Re / ncol Im of D_H e^{i ncol y} (with column 0's share)
                                const cd dH = sNyq[bo.arr[e]][h4 + 4 * v];
                                sum += bo.kind[e] ? (double)ncol * (dH.x * eH.y + dH.y * eH.x) : dH.x * eH.x - dH.y * eH.y;
                            }
                            val += bo.coef[e] * (bo.kind[e] ? -dky * sum : sum);
                        }
                        double* dst = partial + (((int64_t)blockIdx.z * 8 + bo.out) * np + j) * W + sidx;
                        *dst = bo.acc ? *dst + wx * val : wx * val;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void band_reduce_kernel(const double* __restrict__ partial, int64_t n, int64_t np,
                                                          int nslab, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int z = 0; z < nslab; ++z) {               // (slab z: the column share of blockIdx.z, 8 outputs x np points each)
        const double* p = partial + ((int64_t)z * 8 * np + i) * 16;
#pragma unroll
        for (int m = 0; m < 16; ++m) s += p[m];
    }
    out[i] = s;
}

}  // namespace

struct GridInterp {
    ipde_ctx* ctx = nullptr;
    int64_t nx = 0, ny = 0;
    int w = 16;
    double beta = 0, betax = 0, betay = 0;
    int64_t nfx = 0, nfy = 0;     // fine grid
    bool general = false;         // arbitrary coarse size, unpacked (rocFFT) spectra, fine = next
                                  // power of two >= 2 n per axis
    bool shifted = false;         // fine size beyond the FFT kernels: four shifted coarse transforms
    Fft2dPlan fine;               // (2 nx, 2 ny): W[0..2] = the three fields' fine half spectra
    double* d_rx = nullptr;       // h_f / psihat_x(k), k = 0 .. nx/2
    double* d_ry = nullptr;       // k = 0 .. ny/2
    double* g[3] = {nullptr, nullptr, nullptr};   // fine real grids
    void* spec[3] = {nullptr, nullptr, nullptr};  // packed spectra of input fields (grid_interp_fields)
    double* stage = nullptr;      // host-call staging (points, results)
    size_t stage_bytes = 0;
    double hx = 0, hy = 0;
    bool legacy_ready = false;    // the full-fine-grid form's buffers (fine plan of the packed variant, g[])
    // band form
    cd* bandD[3] = {nullptr, nullptr, nullptr};   // packed variant: (2 nx, band_pitch) fine-row-interleaved
    int64_t band_pitch = 0;
    double* d_ones = nullptr;     // the y window factor of the pad kernels: 1
    cd* d_roots = nullptr;        // the 256th roots of unity (the gather's phase table)
    int* ibuf = nullptr;          // cnt, start (nfx + 33 each), perm, r0, rank (np each)
    int64_t ibuf_np = -1;
    double* partial = nullptr;    // (column shares, 8, np, 16)
};

bool grid_interp_supported(int64_t nx, int64_t ny) { return fft2d_supported(nx, ny); }

static int64_t pow2_at_least(int64_t n) {
    int64_t p = 1;
    while (p < n) p *= 2;
    return p;
}
static void general_fine_size(int64_t nx, int64_t ny, int64_t* nfx, int64_t* nfy) {
    *nfx = std::max<int64_t>(512, pow2_at_least(2 * nx));
    *nfy = std::max<int64_t>(1024, pow2_at_least(2 * ny));
}
bool grid_interp_general_supported(int64_t nx, int64_t ny) {
    int64_t a, b;
    general_fine_size(nx, ny, &a, &b);
    return nx >= 8 && ny >= 8 && fft2d_supported(a, b);
}

static bool g_force_shifted = false;
void grid_interp_force_shifted(bool on) { g_force_shifted = on; }

void grid_interp_destroy(GridInterp* gi) {
    if (!gi) return;
    fft2d_plan_free(gi->fine);
    if (gi->d_rx) (void)hipFree(gi->d_rx);
    if (gi->d_ry) (void)hipFree(gi->d_ry);
    for (auto& g : gi->g)
        if (g) (void)hipFree(g);
    for (auto& q : gi->spec)
        if (q) (void)hipFree(q);
    for (auto& q : gi->bandD)
        if (q) (void)hipFree(q);
    if (gi->d_ones) (void)hipFree(gi->d_ones);
    if (gi->d_roots) (void)hipFree(gi->d_roots);
    if (gi->ibuf) (void)hipFree(gi->ibuf);
    if (gi->partial) (void)hipFree(gi->partial);
    if (gi->stage) (void)hipFree(gi->stage);
    delete gi;
}

// general = true: arbitrary coarse size with unpacked spectra (see nufft_pad_general_kernel)
int grid_interp_create(ipde_ctx* ctx, int64_t nx, int64_t ny, double hx, double hy, GridInterp** out,
                       bool general) {
    GridInterp* gi = new GridInterp();
    gi->ctx = ctx;
    gi->nx = nx;
    gi->ny = ny;
    gi->w = 16;
    gi->general = general;
    if (general) {
        general_fine_size(nx, ny, &gi->nfx, &gi->nfy);
        gi->shifted = false;
    } else {
        gi->nfx = 2 * nx;
        gi->nfy = 2 * ny;
        gi->shifted = g_force_shifted || !fft2d_supported(2 * nx, 2 * ny);
    }
    // window shape for an oversampling factor sigma (2.30 w at sigma = 2, the FINUFFT choice
    // gamma pi w (1 - 1/(2 sigma)), gamma = 0.976)
    auto beta_of = [&](double sigma) { return 0.976 * M_PI * gi->w * (1.0 - 0.5 / sigma); };
    gi->betax = general ? beta_of((double)gi->nfx / nx) : 2.30 * gi->w;
    gi->betay = general ? beta_of((double)gi->nfy / ny) : 2.30 * gi->w;
    gi->beta = gi->betax;
    gi->hx = hx;
    gi->hy = hy;
    // (general sizes: the zero-padded fine spectra serve both forms; the packed variant's fine plan and the
    // fine real grids belong to the full-fine-grid form alone and are created when it first runs)
    int st = general ? fft2d_plan_init(ctx, gi->fine, gi->nfx, gi->nfy, hx * nx / gi->nfx, hy * ny / gi->nfy)
                     : IPDE_OK;
    std::vector<double> rx, ry;
    window_factors(gi->nfx, gi->w, gi->betax, nx / 2 + 1, rx);
    window_factors(gi->nfy, gi->w, gi->betay, ny / 2 + 1, ry);
    auto up = [&](double** d, const std::vector<double>& h) {
        if (st == IPDE_OK && hipMalloc((void**)d, h.size() * sizeof(double)) != hipSuccess) st = IPDE_ERR_ALLOC;
        if (st == IPDE_OK && hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            st = IPDE_ERR_HIP;
    };
    up(&gi->d_rx, rx);
    up(&gi->d_ry, ry);
    up(&gi->d_ones, std::vector<double>((size_t)ny / 2 + 2, 1.0));
    {
        std::vector<double> roots(512);
        for (int i = 0; i < 256; ++i) {
            const long double a = 2.0L * 3.14159265358979323846264338327950288L * i / 256.0L;
            roots[2 * i] = (double)cosl(a);
            roots[2 * i + 1] = (double)sinl(a);
        }
        double* r = nullptr;
        up(&r, roots);
        gi->d_roots = (cd*)r;
    }
    // the columns of the fine half spectra beyond the coarse band are never written: zero once
    for (auto& w : gi->fine.W)
        if (general && st == IPDE_OK && w &&
            hipMemset(w, 0, (size_t)gi->nfx * (gi->nfy / 2) * 2 * sizeof(double)) != hipSuccess)
            st = IPDE_ERR_HIP;
    if (st != IPDE_OK) {
        grid_interp_destroy(gi);
        return st;
    }
    *out = gi;
    return IPDE_OK;
}

// the full-fine-grid form's own buffers, at its first use
static int ensure_legacy(GridInterp* gi) {
    if (gi->legacy_ready) return IPDE_OK;
    ipde_ctx* ctx = gi->ctx;
    if (!gi->general && !gi->shifted && !gi->fine.ready) {
        IPDE_TRY(fft2d_plan_init(ctx, gi->fine, gi->nfx, gi->nfy, gi->hx * gi->nx / gi->nfx, gi->hy * gi->ny / gi->nfy));
        for (auto& w : gi->fine.W)
            IPDE_HIP_CHECK(ctx, hipMemset(w, 0, (size_t)gi->nfx * (gi->nfy / 2) * 2 * sizeof(double)));
    }
    const size_t gbytes = (size_t)gi->nfx * gi->nfy * sizeof(double);
    for (auto& g : gi->g)
        if (!g) IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g, gbytes));
    gi->legacy_ready = true;
    return IPDE_OK;
}

static int ensure_band(GridInterp* gi, int64_t np) {
    ipde_ctx* ctx = gi->ctx;
    if (!gi->general && !gi->bandD[0]) {
        const int64_t ncol = gi->ny / 2 + 1;
        gi->band_pitch = (ncol + 3) / 4 * 4;
        const size_t bytes = (size_t)2 * gi->nx * gi->band_pitch * sizeof(cd);      // (fine-row-interleaved)
        for (auto& d : gi->bandD) {
            IPDE_HIP_CHECK(ctx, hipMalloc((void**)&d, bytes));
            IPDE_HIP_CHECK(ctx, hipMemset(d, 0, bytes));     // (the padding columns go through the column pass too)
        }
    }
    if (np > gi->ibuf_np) {
        if (gi->ibuf) (void)hipFree(gi->ibuf);
        if (gi->partial) (void)hipFree(gi->partial);
        gi->ibuf = nullptr;
        gi->partial = nullptr;
        gi->ibuf_np = -1;
        const int64_t cap = np + np / 4 + 64;
        IPDE_HIP_CHECK(ctx, hipMalloc((void**)&gi->ibuf, (size_t)(2 * (gi->nfx + 33) + 3 * cap) * sizeof(int)));
        IPDE_HIP_CHECK(ctx, hipMalloc((void**)&gi->partial, (size_t)BAND_KSPLIT_MAX * 8 * cap * 16 * sizeof(double)));
        gi->ibuf_np = cap;
    }
    return IPDE_OK;
}

// (C: columns per workgroup — 64-byte row segments at 2048^2, 139 KB of LDS, one workgroup per CU; half of that,
// two workgroups per CU, measured the same at 2048^2 (51.5 against 53.3 us) and worse at 4096^2 (273 against 253):
// the phases of a workgroup do not wait for each other's CU)
template <int NX>
static int launch_band_cols(GridInterp* gi, const Fft2dPlan& coarse, const Combo& cb, cd* D, double dkx, double dky) {
    constexpr int C = NX >= 4096 ? 2 : 4;
    using namespace fftcore;
    constexpr int T = Cfg<NX>::T;
    ipde_ctx* ctx = gi->ctx;
    const size_t lds = (size_t)C * (lds_slots<NX>() + 4) * sizeof(fftcore::cd);
    const int ncol = (int)(gi->ny / 2);                  // column ny/2 rides in column 0
    const int nblocks = (ncol + C - 1) / C;
    auto k = band_col_kernel<NX, C>;
    if (lds > 48 * 1024)
        IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)(((nblocks + 7) / 8) * 8)), dim3(C * T), lds, ctx->stream, cb, D, (int)gi->ny,
                       (int)gi->band_pitch, nblocks, (const double*)gi->d_rx, dkx, dky, (const fftcore::cd*)coarse.tw_x);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// An output combo's x part (terms D^der, der in {0, 1}) and its y-derivative part (der == 2, with the i ky left to
// the gather) are arrays to transform; equal arrays up to a scalar are transformed once (the scalar solvers'
// u, du/dx, du/dy: two arrays; the Stokes solver's five fields: five).
struct BandArray {
    Combo cb;
};
static int band_find_or_add(std::vector<BandArray>& arrs, const Combo& c, double* scale) {
    for (size_t i = 0; i < arrs.size(); ++i) {
        const Combo& a = arrs[i].cb;
        if (a.n != c.n) continue;
        bool same = true;
        const double sc = c.coef[0] / a.coef[0];
        for (int t = 0; t < c.n && same; ++t)
            same = a.src[t] == c.src[t] && a.der[t] == c.der[t] &&
                   fabs(c.coef[t] - sc * a.coef[t]) <= 4e-16 * fabs(c.coef[t]);
        if (same) {
            *scale = sc;
            return (int)i;
        }
    }
    arrs.push_back(BandArray{c});
    *scale = 1.0;
    return (int)arrs.size() - 1;
}

// The band form of interp_combos (see the head of the file).
static int interp_combos_band(GridInterp* gi, const Fft2dPlan& coarse, int nout, const Combo* combos, int64_t np,
                              const double* d_px, const double* d_py, double dkx, double dky, double* d_out) {
    ipde_ctx* ctx = gi->ctx;
    const int64_t nx = gi->nx, ny = gi->ny;
    if (np < 1) return IPDE_OK;
    if (np >= (1LL << 30) || nout > 8) return IPDE_ERR_INVALID;
    IPDE_TRY(ensure_band(gi, np));
    const int nrows_ext = (int)gi->nfx + 32;
    int* cnt = gi->ibuf;
    int* start = cnt + (gi->nfx + 33);
    int* perm = start + (gi->nfx + 33);
    int* r0v = perm + gi->ibuf_np;
    int* rank = r0v + gi->ibuf_np;
    hipLaunchKernelGGL(band_sort_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_px, np, (int)gi->nfx, gi->w, nrows_ext, cnt,
                       start, perm, r0v, rank);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    // arrays and recipes
    std::vector<BandArray> arrs;
    struct Use {
        int arr[2], kind[2], n;
        double coef[2];
    } use[8];
    for (int f = 0; f < nout; ++f) {
        Combo P{0, {nullptr, nullptr, nullptr}, {0, 0, 0}, {0.0, 0.0, 0.0}}, Qd = P;
        for (int t = 0; t < combos[f].n; ++t) {
            Combo& dst = combos[f].der[t] == 2 ? Qd : P;
            dst.src[dst.n] = combos[f].src[t];
            dst.der[dst.n] = combos[f].der[t] == 2 ? 0 : combos[f].der[t];
            dst.coef[dst.n] = combos[f].coef[t];
            ++dst.n;
        }
        use[f].n = 0;
        for (int kind = 0; kind < 2; ++kind) {
            const Combo& c = kind ? Qd : P;
            if (c.n == 0) continue;
            double sc;
            const int a = band_find_or_add(arrs, c, &sc);
            use[f].arr[use[f].n] = a;
            use[f].kind[use[f].n] = kind;
            use[f].coef[use[f].n] = sc;
            ++use[f].n;
        }
    }
    // column shares per gather launch (IPDE_BAND_KSPLIT, default 1): measured with two chunks per round at 2048^2 x 4096 /
    // 4096^2 x 8192 points: 127 / 403 us with one share, 129 / 400 with two, 154 / 442 with four — the row tiles
    // alone fill the CUs, and what a workgroup waits for is not hidden by a second one on its CU
    static const int ks_env = getenv("IPDE_BAND_KSPLIT") ? atoi(getenv("IPDE_BAND_KSPLIT")) : 0;
    const int ntile_k = (int)((ny / 2 + 1 + 127) / 128);
    int ks = ks_env > 0 ? ks_env : 1;
    if (ks > BAND_KSPLIT_MAX) ks = BAND_KSPLIT_MAX;
    if (ks > ntile_k) ks = ntile_k;
    // (the packed path's column ny/2 rides in column 0 as its imaginary part: ny/2 columns there)
    const int packed0 = gi->general ? 0 : 1;
    const int ncol = (int)(ny / 2) + (packed0 ? 0 : 1);
    bool started[8] = {false, false, false, false, false, false, false, false};
    for (int a0 = 0; a0 < (int)arrs.size(); a0 += 3) {
        const int na = (int)arrs.size() - a0 < 3 ? (int)arrs.size() - a0 : 3;
        const cd* D[3] = {nullptr, nullptr, nullptr};
        int64_t pitch = 0;
        for (int f = 0; f < na; ++f) {
            const Combo& cb = arrs[a0 + f].cb;
            if (gi->general) {
                const int64_t nthreads = gi->nfx * (ny / 2 + 1);
                hipLaunchKernelGGL(nufft_pad_general_kernel, dim3((unsigned)ceil_div64(nthreads, 256)), dim3(256), 0,
                                   ctx->stream, cb, (cd*)gi->fine.W[f], (int)nx, (int)ny, (int)gi->nfx, (int)gi->nfy,
                                   (const double*)gi->d_rx, (const double*)gi->d_ones, dkx, dky);
                IPDE_HIP_CHECK(ctx, hipGetLastError());
                IPDE_TRY(fft2d_cols(ctx, gi->fine, f, FFT2D_SYM_NONE, 2, 0.0, 1.0, -1, ny / 2 + 1));
                D[f] = (const cd*)gi->fine.W[f];
                pitch = gi->fine.pitch;
            } else {
                int st = IPDE_ERR_INVALID;
                switch (nx) {
                    case 512: st = launch_band_cols<512>(gi, coarse, cb, gi->bandD[f], dkx, dky); break;
                    case 1024: st = launch_band_cols<1024>(gi, coarse, cb, gi->bandD[f], dkx, dky); break;
                    case 2048: st = launch_band_cols<2048>(gi, coarse, cb, gi->bandD[f], dkx, dky); break;
                    case 4096: st = launch_band_cols<4096>(gi, coarse, cb, gi->bandD[f], dkx, dky); break;
                }
                IPDE_TRY(st);
                D[f] = gi->bandD[f];
                pitch = gi->band_pitch;
            }
        }
        BandRecipe rc{};
        for (int f = 0; f < nout; ++f) {
            BandOut bo{};
            bo.out = f;
            for (int e = 0; e < use[f].n; ++e) {
                const int a = use[f].arr[e];
                if (a < a0 || a >= a0 + na) continue;
                bo.arr[bo.n] = a - a0;
                bo.kind[bo.n] = use[f].kind[e];
                bo.coef[bo.n] = use[f].coef[e];
                if (use[f].kind[e]) rc.need_dy[a - a0] = 1;
                ++bo.n;
            }
            if (bo.n == 0) continue;
            bo.acc = started[f] ? 1 : 0;
            started[f] = true;
            rc.o[rc.nout++] = bo;
        }
        if (rc.nout == 0) continue;
        {
            // the gather as a block GEMM on the matrix cores: a workgroup per tile of 16 fine rows
            const unsigned ntiles = (unsigned)((nrows_ext + 15) / 16);
#define BAND_MFMA(NA)                                                                                                  \
    hipLaunchKernelGGL(band_gather_mfma_kernel<NA>, dim3(ntiles, 16, ks), dim3(256), 0, ctx->stream, D[0], D[1], D[2], rc, pitch,  \
                       ncol, packed0, (int)gi->nfx, nrows_ext, d_px, d_py, np, (const int*)start, (const int*)perm,        \
                       (const int*)r0v, gi->betax, dky, (const cd*)gi->d_roots, gi->partial)
            if (na == 1)
                BAND_MFMA(1);
            else if (na == 2)
                BAND_MFMA(2);
            else
                BAND_MFMA(3);
#undef BAND_MFMA
        }
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    const int64_t n = (int64_t)nout * np;
    hipLaunchKernelGGL(band_reduce_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream,
                       (const double*)gi->partial, n, np, ks, d_out);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// The core: nout output fields, each a Combo over packed coarse spectra, at np points.
// d_px, d_py, d_out: device; out (nout, np).  Three fields per sweep (the g buffers).
static int interp_combos(GridInterp* gi, const Fft2dPlan& coarse, int nout, const Combo* combos,
                         int64_t np, const double* d_px, const double* d_py, double dkx, double dky,
                         double* d_out) {
    ipde_ctx* ctx = gi->ctx;
    const int64_t nx = gi->nx, ny = gi->ny;
    if (ctx->opt_interp_band) return interp_combos_band(gi, coarse, nout, combos, np, d_px, d_py, dkx, dky, d_out);
    IPDE_TRY(ensure_legacy(gi));
    for (int f0 = 0; f0 < nout; f0 += 3) {
        const int nf = nout - f0 < 3 ? nout - f0 : 3;
        for (int f = 0; f < nf; ++f) {
            const Combo& cb = combos[f0 + f];
            if (gi->shifted) {
                // W[2] of the coarse plan is scratch; sub-grid (a, b) lands in g[f] + (2a+b) nx ny
                const int64_t nth = nx * (ny / 2);
                for (int ab = 0; ab < 4; ++ab) {
                    hipLaunchKernelGGL(nufft_shift_kernel, dim3((unsigned)ceil_div64(nth, 256)), dim3(256),
                                       0, ctx->stream, cb, (cd*)coarse.W[2], (int)nx, (int)ny,
                                       (const double*)gi->d_rx, (const double*)gi->d_ry, ab >> 1, ab & 1,
                                       dkx, dky);
                    IPDE_HIP_CHECK(ctx, hipGetLastError());
                    IPDE_TRY(fft2d_cols(ctx, coarse, 2, FFT2D_SYM_NONE, 2, 0.0, 1.0));
                    IPDE_TRY(fft2d_rows_inverse(ctx, coarse, 2, gi->g[f] + (int64_t)ab * nx * ny));
                }
            } else if (gi->general) {
                const int64_t nthreads = gi->nfx * (ny / 2 + 1);
                hipLaunchKernelGGL(nufft_pad_general_kernel, dim3((unsigned)ceil_div64(nthreads, 256)),
                                   dim3(256), 0, ctx->stream, cb, (cd*)gi->fine.W[f], (int)nx, (int)ny,
                                   (int)gi->nfx, (int)gi->nfy, (const double*)gi->d_rx,
                                   (const double*)gi->d_ry, dkx, dky);
                IPDE_HIP_CHECK(ctx, hipGetLastError());
                IPDE_TRY(fft2d_cols(ctx, gi->fine, f, FFT2D_SYM_NONE, 2, 0.0, 1.0, -1, ny / 2 + 1));
                IPDE_TRY(fft2d_rows_inverse(ctx, gi->fine, f, gi->g[f]));
                continue;
            } else {
                const int64_t nthreads = 2 * nx * (ny / 2 + 1);
                hipLaunchKernelGGL(nufft_pad_kernel, dim3((unsigned)ceil_div64(nthreads, 256)), dim3(256), 0,
                                   ctx->stream, cb, (cd*)gi->fine.W[f], (int)nx, (int)ny,
                                   (const double*)gi->d_rx, (const double*)gi->d_ry, dkx, dky);
                IPDE_HIP_CHECK(ctx, hipGetLastError());
                IPDE_TRY(fft2d_cols(ctx, gi->fine, f, FFT2D_SYM_NONE, 2, 0.0, 1.0, -1, ny / 2 + 1));
                IPDE_TRY(fft2d_rows_inverse(ctx, gi->fine, f, gi->g[f]));
            }
        }
        double* o = d_out + (int64_t)f0 * np;
        if (gi->shifted)
            hipLaunchKernelGGL((nufft_gather_kernel<16, true>), dim3((unsigned)np), dim3(64), 0, ctx->stream,
                               (const double*)gi->g[0], (const double*)gi->g[1], (const double*)gi->g[2], nf,
                               (int)gi->nfx, (int)gi->nfy, d_px, d_py, np, gi->betax, gi->betay, o);
        else
            hipLaunchKernelGGL((nufft_gather_kernel<16, false>), dim3((unsigned)np), dim3(64), 0, ctx->stream,
                               (const double*)gi->g[0], (const double*)gi->g[1], (const double*)gi->g[2], nf,
                               (int)gi->nfx, (int)gi->nfy, d_px, d_py, np, gi->betax, gi->betay, o);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    }
    return IPDE_OK;
}

static int stage_points(GridInterp* gi, int loc, int64_t np, int nout, const double* px, const double* py,
                        double* out, const double** d_px, const double** d_py, double** d_out) {
    ipde_ctx* ctx = gi->ctx;
    *d_px = px;
    *d_py = py;
    *d_out = out;
    if (loc == IPDE_HOST) {
        const size_t need = (size_t)(2 + nout) * np * sizeof(double);
        if (need > gi->stage_bytes) {
            if (gi->stage) (void)hipFree(gi->stage);
            gi->stage = nullptr;
            gi->stage_bytes = 0;
            IPDE_HIP_CHECK(ctx, hipMalloc((void**)&gi->stage, need));
            gi->stage_bytes = need;
        }
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(gi->stage, px, np * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(gi->stage + np, py, np * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        *d_px = gi->stage;
        *d_py = gi->stage + np;
        *d_out = gi->stage + 2 * np;
    }
    return IPDE_OK;
}

static int finish_points(GridInterp* gi, int loc, int64_t np, int nout, double* out, const double* d_out) {
    ipde_ctx* ctx = gi->ctx;
    if (loc == IPDE_HOST) {
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, (size_t)nout * np * sizeof(double), hipMemcpyDeviceToHost,
                                           ctx->stream));
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return IPDE_OK;
}

// The kept spectrum of the last scalar grid solve (coarse.W[1] = fft2(f) * symbol * 2 / (nx ny),
// packed): out (3, np) = u, du/dx, du/dy at the points (box units [0, 2 pi)), derivatives in
// physical units (dkx, dky = 2 pi / box length).
int grid_interp_eval(GridInterp* gi, const Fft2dPlan& coarse, int loc, int64_t np, const double* px,
                     const double* py, double dkx, double dky, double* out) {
    const double *d_px, *d_py;
    double* d_out;
    IPDE_TRY(stage_points(gi, loc, np, 3, px, py, out, &d_px, &d_py, &d_out));
    // (the kept spectrum carries the grid solve's 2 / (nx ny): exactly what the inverse needs)
    Combo cb[3];
    for (int f = 0; f < 3; ++f) cb[f] = Combo{1, {(const cd*)coarse.W[1], nullptr, nullptr}, {f, 0, 0}, {1.0, 0.0, 0.0}};
    IPDE_TRY(interp_combos(gi, coarse, 3, cb, np, d_px, d_py, dkx, dky, d_out));
    return finish_points(gi, loc, np, 3, out, d_out);
}

// Linear combinations of real grid fields and their first derivatives at points (the Stokes
// solver's velocity and stress on the interfaces, reference multi_boundary/vector.py:66-82):
// d_fields: nin device (nx, ny) arrays; output k = sum over its terms of coef * D^der [field src].
int grid_interp_fields(GridInterp* gi, const Fft2dPlan& coarse, int nin, const double* const* d_fields,
                       int nout, const int* term_start, const int* term_src, const int* term_der,
                       const double* term_coef, int loc_points, int64_t np, const double* px,
                       const double* py, double dkx, double dky, double* out) {
    ipde_ctx* ctx = gi->ctx;
    const int64_t nx = gi->nx, ny = gi->ny;
    if (nin > 3 || nout > 8) return IPDE_ERR_INVALID;
    for (int k = 0; k < nin; ++k) {
        if (!gi->spec[k])
            IPDE_HIP_CHECK(ctx, hipMalloc(&gi->spec[k], (size_t)nx * (ny / 2) * 2 * sizeof(double)));
        Fft2dPlan tmp = coarse;          // same tables, the spectrum goes to our own buffer
        tmp.W[0] = gi->spec[k];
        IPDE_TRY(fft2d_rows_forward(ctx, tmp, d_fields[k], 0));
        IPDE_TRY(fft2d_cols(ctx, tmp, 0, FFT2D_SYM_NONE, 1, 0.0, 1.0));
    }
    Combo cb[8];
    const double norm = 2.0 / ((double)nx * (double)ny);   // forward transforms are unnormalised
    for (int f = 0; f < nout; ++f) {
        const int n = term_start[f + 1] - term_start[f];
        if (n < 1 || n > 3) return IPDE_ERR_INVALID;
        cb[f].n = n;
        for (int t = 0; t < 3; ++t) {
            const int q = term_start[f] + (t < n ? t : 0);
            if (term_src[q] < 0 || term_src[q] >= nin || term_der[q] < 0 || term_der[q] > 2) return IPDE_ERR_INVALID;
            cb[f].src[t] = (const cd*)gi->spec[term_src[q]];
            cb[f].der[t] = term_der[q];
            cb[f].coef[t] = t < n ? term_coef[q] * norm : 0.0;
        }
    }
    const double *d_px, *d_py;
    double* d_out;
    IPDE_TRY(stage_points(gi, loc_points, np, nout, px, py, out, &d_px, &d_py, &d_out));
    IPDE_TRY(interp_combos(gi, coarse, nout, cb, np, d_px, d_py, dkx, dky, d_out));
    return finish_points(gi, loc_points, np, nout, out, d_out);
}

// ---- general grid sizes: spectra are rocFFT's unpacked half spectra (nx, ny/2 + 1) ----------------
// spec: c_k = fft2(f)_k * symbol_k / (nx ny) (what spectral.hip's scalar solve leaves behind)
int grid_interp_eval_general(GridInterp* gi, const void* spec, int loc, int64_t np, const double* px,
                             const double* py, double dkx, double dky, double* out) {
    const double *d_px, *d_py;
    double* d_out;
    IPDE_TRY(stage_points(gi, loc, np, 3, px, py, out, &d_px, &d_py, &d_out));
    Combo cb[3];
    // (the fine inverse returns half of the unnormalised sum: coefficient 2)
    for (int f = 0; f < 3; ++f) cb[f] = Combo{1, {(const cd*)spec, nullptr, nullptr}, {f, 0, 0}, {2.0, 0.0, 0.0}};
    Fft2dPlan none;
    IPDE_TRY(interp_combos(gi, none, 3, cb, np, d_px, d_py, dkx, dky, d_out));
    return finish_points(gi, loc, np, 3, out, d_out);
}

// specs[k]: UNNORMALISED half spectra (rocFFT r2c) of the nin input fields
int grid_interp_fields_general(GridInterp* gi, int nin, const void* const* specs, int nout,
                               const int* term_start, const int* term_src, const int* term_der,
                               const double* term_coef, int loc_points, int64_t np, const double* px,
                               const double* py, double dkx, double dky, double* out) {
    if (nin > 3 || nout > 8) return IPDE_ERR_INVALID;
    Combo cb[8];
    const double norm = 2.0 / ((double)gi->nx * (double)gi->ny);
    for (int f = 0; f < nout; ++f) {
        const int n = term_start[f + 1] - term_start[f];
        if (n < 1 || n > 3) return IPDE_ERR_INVALID;
        cb[f].n = n;
        for (int t = 0; t < 3; ++t) {
            const int q = term_start[f] + (t < n ? t : 0);
            if (term_src[q] < 0 || term_src[q] >= nin || term_der[q] < 0 || term_der[q] > 2) return IPDE_ERR_INVALID;
            cb[f].src[t] = (const cd*)specs[term_src[q]];
            cb[f].der[t] = term_der[q];
            cb[f].coef[t] = t < n ? term_coef[q] * norm : 0.0;
        }
    }
    const double *d_px, *d_py;
    double* d_out;
    IPDE_TRY(stage_points(gi, loc_points, np, nout, px, py, out, &d_px, &d_py, &d_out));
    Fft2dPlan none;
    IPDE_TRY(interp_combos(gi, none, nout, cb, np, d_px, d_py, dkx, dky, d_out));
    return finish_points(gi, loc_points, np, nout, out, d_out);
}
