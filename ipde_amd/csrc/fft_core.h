// The in-register / LDS Stockham FFT of one workgroup (or one wavefront) on complex fp64 data —
// shared by the 2-D pipeline (fft2d.hip: rows, fused columns) and the annular solvers' fused
// transform pairs (annular.hip).  Device code only; every translation unit gets its own copy
// (anonymous namespace inside fftcore).
#pragma once
#include <hip/hip_runtime.h>

namespace fftcore {
namespace {

// (16-byte aligned: LDS exchanges are ds_read_b128 / ds_write_b128, not pairs of 8-byte accesses)
struct alignas(16) cd {
    double x, y;
};
__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd b) {
    return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ cd cconj(cd a) { return cd{a.x, -a.y}; }
// multiply by exp(SIGN * i * pi / 2) = SIGN * i
template <int SIGN>
__device__ __forceinline__ cd rot90(cd a) {
    return SIGN > 0 ? cd{-a.y, a.x} : cd{a.y, -a.x};
}
// multiply by the constant (c, SIGN * s)
template <int SIGN>
__device__ __forceinline__ cd mulc(cd a, double c, double s) {
    return SIGN > 0 ? cd{a.x * c - a.y * s, a.x * s + a.y * c} : cd{a.x * c + a.y * s, a.y * c - a.x * s};
}

#define C_PI8 0.92387953251128675613   // cos(pi/8)
#define S_PI8 0.38268343236508977173   // sin(pi/8)
#define R_HALF 0.70710678118654752440  // sqrt(1/2)

// ---- small DFTs, natural-order output, w = exp(SIGN 2 pi i / R) -----------------------
template <int SIGN>
__device__ __forceinline__ void dft4(cd& a, cd& b, cd& c, cd& d) {
    cd s0 = a + c, s1 = a - c, s2 = b + d, s3 = rot90<SIGN>(b - d);
    a = s0 + s2;
    b = s1 + s3;
    c = s0 - s2;
    d = s1 - s3;
}

template <int R, int SIGN>
struct Dft;

template <int SIGN>
struct Dft<4, SIGN> {
    static __device__ __forceinline__ void run(cd (&u)[4]) { dft4<SIGN>(u[0], u[1], u[2], u[3]); }
};

// n = n0 + 2 n1, k = k1 + 4 k0
template <int SIGN>
struct Dft<8, SIGN> {
    static __device__ __forceinline__ void run(cd (&u)[8]) {
        cd e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6];   // n0 = 0
        cd o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];   // n0 = 1
        dft4<SIGN>(e0, e1, e2, e3);
        dft4<SIGN>(o0, o1, o2, o3);
        o1 = mulc<SIGN>(o1, R_HALF, R_HALF);
        o2 = rot90<SIGN>(o2);
        o3 = mulc<SIGN>(o3, -R_HALF, R_HALF);
        u[0] = e0 + o0;
        u[4] = e0 - o0;
        u[1] = e1 + o1;
        u[5] = e1 - o1;
        u[2] = e2 + o2;
        u[6] = e2 - o2;
        u[3] = e3 + o3;
        u[7] = e3 - o3;
    }
};

// n = n0 + 4 n1, k = k1 + 4 k0
template <int SIGN>
struct Dft<16, SIGN> {
    static __device__ __forceinline__ void run(cd (&u)[16]) {
        cd y[4][4];
#pragma unroll
        for (int n0 = 0; n0 < 4; ++n0) {
            y[n0][0] = u[n0];
            y[n0][1] = u[n0 + 4];
            y[n0][2] = u[n0 + 8];
            y[n0][3] = u[n0 + 12];
            dft4<SIGN>(y[n0][0], y[n0][1], y[n0][2], y[n0][3]);
        }
        // twiddles W16^(n0 k1)
        y[1][1] = mulc<SIGN>(y[1][1], C_PI8, S_PI8);
        y[1][2] = mulc<SIGN>(y[1][2], R_HALF, R_HALF);
        y[1][3] = mulc<SIGN>(y[1][3], S_PI8, C_PI8);
        y[2][1] = mulc<SIGN>(y[2][1], R_HALF, R_HALF);
        y[2][2] = rot90<SIGN>(y[2][2]);
        y[2][3] = mulc<SIGN>(y[2][3], -R_HALF, R_HALF);
        y[3][1] = mulc<SIGN>(y[3][1], S_PI8, C_PI8);
        y[3][2] = mulc<SIGN>(y[3][2], -R_HALF, R_HALF);
        y[3][3] = mulc<SIGN>(y[3][3], -C_PI8, -S_PI8);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            dft4<SIGN>(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
            u[k1] = y[0][k1];
            u[k1 + 4] = y[1][k1];
            u[k1 + 8] = y[2][k1];
            u[k1 + 12] = y[3][k1];
        }
    }
};

// ---- pass twiddles: u[r] *= w^r, powers by products of depth <= 4 -------------------------
template <int R>
__device__ __forceinline__ void twiddle(cd (&u)[R], cd w1) {
    cd w2 = cmul(w1, w1);
    u[1] = cmul(u[1], w1);
    u[2] = cmul(u[2], w2);
    cd w3 = cmul(w2, w1);
    u[3] = cmul(u[3], w3);
    if constexpr (R > 4) {
        cd w4 = cmul(w2, w2);
        u[4] = cmul(u[4], w4);
        u[5] = cmul(u[5], cmul(w4, w1));
        u[6] = cmul(u[6], cmul(w4, w2));
        u[7] = cmul(u[7], cmul(w4, w3));
        if constexpr (R > 8) {
            cd w8 = cmul(w4, w4);
            u[8] = cmul(u[8], w8);
            u[9] = cmul(u[9], cmul(w8, w1));
            u[10] = cmul(u[10], cmul(w8, w2));
            u[11] = cmul(u[11], cmul(w8, w3));
            cd w12 = cmul(w8, w4);
            u[12] = cmul(u[12], w12);
            u[13] = cmul(u[13], cmul(w12, w1));
            u[14] = cmul(u[14], cmul(w12, w2));
            u[15] = cmul(u[15], cmul(w12, w3));
        }
    }
}

// ---- FFT configurations -------------------------------------------------------------------
template <int N>
struct Cfg;
template <>
struct Cfg<512> {
    static constexpr int T = 64, P = 8, R1 = 8, R2 = 8, R3 = 8;
};
template <>
struct Cfg<1024> {
    static constexpr int T = 64, P = 16, R1 = 16, R2 = 4, R3 = 16;
};
template <>
struct Cfg<2048> {
    static constexpr int T = 128, P = 16, R1 = 16, R2 = 8, R3 = 16;
};
template <>
struct Cfg<4096> {
    static constexpr int T = 256, P = 16, R1 = 16, R2 = 16, R3 = 16;
};

__device__ __forceinline__ int padpos(int p) { return p + (p >> 4); }
template <int N>
constexpr int lds_slots() {
    return N + N / 16;
}

// The twiddles a thread needs for passes two and three.  Butterfly i of a pass with stride NS takes
// w = exp(-2 pi i k / (NS R)), k = (t + T i) mod NS; every configuration above has NS | T, so k = t mod NS
// whatever i is: ONE table entry per pass and thread.  Loaded at the top of a kernel, next to the
// data, they cost no latency of their own (read where the pass needs them, each was a dependent
// global load: two exposed L2 round trips per transform at two waves per SIMD).
template <int N>
struct PassTw {
    cd w2, w3;
};
template <int N>
__device__ __forceinline__ PassTw<N> load_pass_twiddles(int t, const cd* __restrict__ tw) {
    using G = Cfg<N>;
    static_assert(G::T % G::R1 == 0 && G::T % (G::R1 * G::R2) == 0, "k = t mod NS needs NS | T");
    static_assert(G::R1 * G::R2 * G::R3 == N, "three passes");
    PassTw<N> p;
    p.w2 = tw[(t & (G::R1 - 1)) * (N / (G::R1 * G::R2))];   // exp(-2 pi i k / (R1 R2))
    p.w3 = tw[t & (G::R1 * G::R2 - 1)];                       // exp(-2 pi i k / N)
    return p;
}

// One Stockham pass on the thread's registers.  Slot convention: butterfly i (of P/R) takes
// slots {i + (P/R) r}; its output r goes back to the same slot.  w1: the pass twiddle of this
// thread (forward sign; unused in the first pass, NS = 1).
template <int N, int T, int P, int R, int NS, int SIGN>
__device__ __forceinline__ void pass_compute(cd (&v)[P], cd w1) {
    constexpr int nb = P / R;
    if (SIGN > 0) w1.y = -w1.y;
#pragma unroll
    for (int i = 0; i < nb; ++i) {
        cd u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = v[i + nb * r];
        if (NS > 1) twiddle<R>(u, w1);
        Dft<R, SIGN>::run(u);
#pragma unroll
        for (int r = 0; r < R; ++r) v[i + nb * r] = u[r];
    }
}

// Where output r of butterfly i of a (non-final) pass lives in the next pass's natural order.
template <int T, int R, int NS>
__device__ __forceinline__ int outpos(int t, int i, int r) {
    const int j = t + T * i;
    return (j / NS) * (NS * R) + (j & (NS - 1)) + r * NS;
}

// WAVE: the LDS region is private to one wavefront (a 64-thread transform): LDS operations
// of a wave execute in order, no workgroup barrier needed.
template <bool WAVE>
__device__ __forceinline__ void lds_sync() {
    if (WAVE)
        __builtin_amdgcn_wave_barrier();
    else
        __syncthreads();
}

// The one data-movement primitive: every thread writes its P complex values to LDS slots
// wpos(s) and then reads the values at slots rpos(q) (slots are already padded, absolute).
template <int P, bool WAVE, typename WP, typename RP>
__device__ __forceinline__ void lds_permute(const cd (&in)[P], cd (&out)[P], cd* __restrict__ buf,
                                            WP wpos, RP rpos) {
#pragma unroll
    for (int s = 0; s < P; ++s) buf[wpos(s)] = in[s];
    lds_sync<WAVE>();
#pragma unroll
    for (int q = 0; q < P; ++q) out[q] = buf[rpos(q)];
    lds_sync<WAVE>();
}

// Length-N FFT of the points held as v[q] = a[t + T q] by the T threads of one transform;
// result in the same layout, natural order, unnormalised.  tw[m] = exp(-2 pi i m / N);
// buf: this transform's lds_slots<N>() complex slots.
template <int N, int SIGN, bool WAVE>
__device__ __forceinline__ void fft_regs(cd (&v)[Cfg<N>::P], int t, const PassTw<N>& pw,
                                         cd* __restrict__ buf) {
    using G = Cfg<N>;
    constexpr int T = G::T, P = G::P;
    auto natural = [&](int q) { return padpos(t + T * q); };
    pass_compute<N, T, P, G::R1, 1, SIGN>(v, cd{1.0, 0.0});
    lds_permute<P, WAVE>(v, v, buf,
                         [&](int s) { return padpos(outpos<T, G::R1, 1>(t, s % (P / G::R1), s / (P / G::R1))); },
                         natural);
    pass_compute<N, T, P, G::R2, G::R1, SIGN>(v, pw.w2);
    lds_permute<P, WAVE>(v, v, buf,
                         [&](int s) { return padpos(outpos<T, G::R2, G::R1>(t, s % (P / G::R2), s / (P / G::R2))); },
                         natural);
    pass_compute<N, T, P, G::R3, G::R1 * G::R2, SIGN>(v, pw.w3);
}
// (the same with the twiddles read here: tw[m] = exp(-2 pi i m / N))
template <int N, int SIGN, bool WAVE>
__device__ __forceinline__ void fft_regs(cd (&v)[Cfg<N>::P], int t, const cd* __restrict__ tw,
                                         cd* __restrict__ buf) {
    const PassTw<N> pw = load_pass_twiddles<N>(t, tw);
    fft_regs<N, SIGN, WAVE>(v, t, pw, buf);
}

// value at the mirrored natural index (N - k) mod N for every k the thread holds
template <int N, bool WAVE>
__device__ __forceinline__ void gather_mirror(const cd (&v)[Cfg<N>::P], cd (&out)[Cfg<N>::P], int t,
                                              cd* __restrict__ buf) {
    constexpr int T = Cfg<N>::T, P = Cfg<N>::P;
    lds_permute<P, WAVE>(v, out, buf, [&](int q) { return padpos(t + T * q); },
                              [&](int q) { return padpos((N - (t + T * q)) & (N - 1)); });
}

}  // namespace
}  // namespace fftcore
