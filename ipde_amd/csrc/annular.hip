// Chebyshev x Fourier annular solvers on the device.
//
// Layout: every field is a (rows, n) row-major array, the tangential (Fourier /
// theta) index is the fast one, so that "one thread per column" kernels are
// coalesced and the batched 1-D rocFFTs run along contiguous rows.
//   * operator apply   : small dense Chebyshev matrices mixed over rows (A is
//                        wave-uniform -> scalar loads), pointwise metric factors,
//                        4 batched FFT calls per apply;
//   * preconditioner   : per-mode dense block K_i^{-1}; stored MODE-MINOR
//                        ([row][col][mode]) so the one-thread-per-mode matvec
//                        streams it fully coalesced.  HBM bound: bytes = |KINV|;
//   * GMRES            : right preconditioned, restarted, Krylov basis resident
//                        in HBM, CGS2 orthogonalisation (2 fused multi-dots),
//                        one host sync per iteration (the Hessenberg column).
#include <algorithm>
#include <atomic>

#include "ipde_common.h"
#include "fft_core.h"

int ipde_fft1_exec(ipde_ctx* ctx, int64_t batch, int64_t n, int direction, const void* in,
                   void* out);
extern "C" int ipde_fft1_prepare(ipde_ctx* ctx, int64_t batch, int64_t n);

namespace {

typedef double2 cd;

__device__ __forceinline__ cd cmul(cd a, cd b) {
    return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// out[a, j] = alpha * sum_b A[a,b] * in[b, j] * (colfac ? colfac[j] : 1) + beta * out[a, j]
// complex data, real A.  grid = (ceil(n/256), ra)
__global__ __launch_bounds__(256) void mixc_kernel(cd* __restrict__ out, int ldo,
                                                   const double* __restrict__ A, int ca,
                                                   const cd* __restrict__ in, int ldi, int n,
                                                   const cd* __restrict__ colfac, double alpha,
                                                   double beta) {
    int j = blockIdx.x * 256 + threadIdx.x;
    int a = blockIdx.y;
    if (j >= n) return;
    const double* Ar = A + (size_t)a * ca;
    double sr = 0.0, si = 0.0;
#pragma unroll 4
    for (int b = 0; b < ca; ++b) {
        cd v = in[(size_t)b * ldi + j];
        double w = Ar[b];
        sr = fma(w, v.x, sr);
        si = fma(w, v.y, si);
    }
    cd s{sr, si};
    if (colfac) s = cmul(s, colfac[j]);
    cd o{alpha * s.x, alpha * s.y};
    if (beta != 0.0) {
        cd p = out[(size_t)a * ldo + j];
        o.x = fma(beta, p.x, o.x);
        o.y = fma(beta, p.y, o.y);
    }
    out[(size_t)a * ldo + j] = o;
}

// real-space variant on the REAL PART of complex input rows, real output rows:
// out[a, j] = beta*out[a,j] + alpha * P[a,j] * sum_b A[a,b] * Q[b,j] * Re(in[b, j])
// P, Q nullable real fields.
__global__ __launch_bounds__(256) void mixr_kernel(double* __restrict__ out, int ldo,
                                                   const double* __restrict__ A, int ca,
                                                   const cd* __restrict__ in, int ldi, int n,
                                                   const double* __restrict__ Q, int ldq,
                                                   const double* __restrict__ P, int ldp,
                                                   double alpha, double beta) {
    int j = blockIdx.x * 256 + threadIdx.x;
    int a = blockIdx.y;
    if (j >= n) return;
    const double* Ar = A + (size_t)a * ca;
    double s = 0.0;
    for (int b = 0; b < ca; ++b) {
        double v = in[(size_t)b * ldi + j].x;
        if (Q) v *= Q[(size_t)b * ldq + j];
        s = fma(Ar[b], v, s);
    }
    if (P) s *= P[(size_t)a * ldp + j];
    double o = alpha * s;
    if (beta != 0.0) o = fma(beta, out[(size_t)a * ldo + j], o);
    out[(size_t)a * ldo + j] = o;
}

// same, but the input rows are real arrays
__global__ __launch_bounds__(256) void mixrr_kernel(double* __restrict__ out, int ldo,
                                                    const double* __restrict__ A, int ca,
                                                    const double* __restrict__ in, int ldi, int n,
                                                    const double* __restrict__ P, int ldp,
                                                    double alpha, double beta) {
    int j = blockIdx.x * 256 + threadIdx.x;
    int a = blockIdx.y;
    if (j >= n) return;
    const double* Ar = A + (size_t)a * ca;
    double s = 0.0;
    for (int b = 0; b < ca; ++b) s = fma(Ar[b], in[(size_t)b * ldi + j], s);
    if (P) s *= P[(size_t)a * ldp + j];
    double o = alpha * s;
    if (beta != 0.0) o = fma(beta, out[(size_t)a * ldo + j], o);
    out[(size_t)a * ldo + j] = o;
}

// complex rows times a real field (and a constant): x[a,j] *= s * F[a,j]
__global__ __launch_bounds__(256) void cscale_field_kernel(cd* __restrict__ x, int rows, int n,
                                                           const double* __restrict__ F, double s) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * n) return;
    double f = s * F[idx];
    cd v = x[idx];
    x[idx] = cd{v.x * f, v.y * f};
}

// y[a,j] += alpha * x[a,j] (complex)
__global__ __launch_bounds__(256) void caxpy_kernel(cd* __restrict__ y, const cd* __restrict__ x,
                                                    int64_t n, double alpha) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    cd a = y[idx], b = x[idx];
    y[idx] = cd{fma(alpha, b.x, a.x), fma(alpha, b.y, a.y)};
}

// The scalar annular operator's radial mixes, fused per stage (one launch where there were two):
// a job is  out[a][j] = colfac0[j] (A0 in0)[a][j]  (+ (A1 in1)[a][j], added the way the second of two
// mixc_kernel launches with beta = 1 added it)  (- sub[a][j] for a < sub_rows, the way caxpy_kernel
// subtracted it); blockIdx.z picks the job.  Products, sums and their order are those of the
// separate kernels: bitwise the same operator (tests/test_annular_gpu.py goldens).
struct MixcTerm {
    const double* A;      // (rows, ca) row-major
    const cd* in;         // (ca, n) complex, leading dimension n
    const cd* colfac;     // nullable (n): factor on the sum
    int ca;
};
struct MixcJob {
    cd* out;
    int rows;
    int nterms;           // 1 or 2
    MixcTerm t[2];
    const cd* sub;        // nullable: out[a][j] -= sub[a][j] for a < sub_rows
    int sub_rows;
};
struct MixcBatch {
    MixcJob j[2];
};
__global__ __launch_bounds__(256) void mixc_fused_kernel(MixcBatch B, int n) {
    const MixcJob& J = B.j[blockIdx.z];
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int a = blockIdx.y;
    if (j >= n || a >= J.rows) return;
    cd o{0.0, 0.0};
    for (int k = 0; k < J.nterms; ++k) {
        const MixcTerm& T = J.t[k];
        const double* Ar = T.A + (size_t)a * T.ca;
        double sr = 0.0, si = 0.0;
#pragma unroll 4
        for (int b = 0; b < T.ca; ++b) {
            cd v = T.in[(size_t)b * n + j];
            double w = Ar[b];
            sr = fma(w, v.x, sr);
            si = fma(w, v.y, si);
        }
        cd s{sr, si};
        if (T.colfac) s = cmul(s, T.colfac[j]);
        cd t{1.0 * s.x, 1.0 * s.y};
        if (k == 0) {
            o = t;
        } else {
            o.x = fma(1.0, o.x, t.x);
            o.y = fma(1.0, o.y, t.y);
        }
    }
    if (J.sub && a < J.sub_rows) {
        const cd b = J.sub[(size_t)a * n + j];
        o.x = fma(-1.0, b.x, o.x);
        o.y = fma(-1.0, b.y, o.y);
    }
    J.out[(size_t)a * n + j] = o;
}

// x[r][j] *= s * (r < rows0 ? F0 : F1)[r][j]: the two field multiplications of a stage in one launch
__global__ __launch_bounds__(256) void cscale_field2_kernel(cd* __restrict__ x, int rows0, int rows1, int n,
                                                            const double* __restrict__ F0,
                                                            const double* __restrict__ F1, double s) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n0 = (int64_t)rows0 * n;
    if (idx >= n0 + (int64_t)rows1 * n) return;
    double f = s * (idx < n0 ? F0[idx] : F1[idx - n0]);
    cd v = x[idx];
    x[idx] = cd{v.x * f, v.y * f};
}

// real (rows,n) -> complex with zero imaginary part, times s
__global__ __launch_bounds__(256) void r2c_copy_kernel(cd* __restrict__ y,
                                                       const double* __restrict__ x, int64_t n,
                                                       double s) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    y[idx] = cd{s * x[idx], 0.0};
}
__global__ __launch_bounds__(256) void c2r_real_kernel(double* __restrict__ y,
                                                       const cd* __restrict__ x, int64_t n,
                                                       double s) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    y[idx] = s * x[idx].x;
}

// Nyquist handling of ipde/utilities.py:78-99.  ns = n-1, N2 = n/2.
// splat: (rows, ns) -> (rows, n) with a zero column at N2, times colfac[js] (nullable)
__global__ __launch_bounds__(256) void splat_kernel(cd* __restrict__ out,
                                                    const cd* __restrict__ in, int rows, int n,
                                                    const cd* __restrict__ colfac) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * n) return;
    int a = (int)(idx / n), j = (int)(idx - (int64_t)a * n);
    const int N2 = n / 2, ns = n - 1;
    cd v{0.0, 0.0};
    if (j != N2) {
        int js = j < N2 ? j : j - 1;
        v = in[(size_t)a * ns + js];
        if (colfac) v = cmul(v, colfac[js]);
    }
    out[idx] = v;
}
// desplat: (rows, n) -> (rows, ns) dropping column N2, times colfac[js], times s
__global__ __launch_bounds__(256) void desplat_kernel(cd* __restrict__ out,
                                                      const cd* __restrict__ in, int rows, int n,
                                                      const cd* __restrict__ colfac, double s) {
    const int ns = n - 1, N2 = n / 2;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * ns) return;
    int a = (int)(idx / ns), js = (int)(idx - (int64_t)a * ns);
    int j = js < N2 ? js : js + 1;
    cd v = in[(size_t)a * n + j];
    if (colfac) v = cmul(v, colfac[js]);
    out[idx] = cd{s * v.x, s * v.y};
}

// ---- preconditioners -------------------------------------------------------
// scalar: Kt real [M][M][n] mode-minor.  out[j, i] = sum_k Kt[j][k][i] x[k, i]
// grid = (ceil(n/256), M)
__global__ __launch_bounds__(256) void prec_scalar_kernel(cd* __restrict__ out,
                                                          const double* __restrict__ Kt,
                                                          const cd* __restrict__ x, int M, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (i >= n) return;
    const double* K = Kt + (size_t)j * M * n + i;
    double sr = 0.0, si = 0.0;
#pragma unroll 4
    for (int k = 0; k < M; ++k) {
        double w = K[(size_t)k * n];
        cd v = x[(size_t)k * n + i];
        sr = fma(w, v.x, sr);
        si = fma(w, v.y, si);
    }
    out[(size_t)j * n + i] = cd{sr, si};
}
// complex blocks: Kt complex [B][B][ns] mode-minor, x/out as (B, ns) row-major
__global__ __launch_bounds__(256) void prec_cplx_kernel(cd* __restrict__ out,
                                                        const cd* __restrict__ Kt,
                                                        const cd* __restrict__ x, int B, int ns) {
    int i = blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (i >= ns) return;
    const cd* K = Kt + (size_t)j * B * ns + i;
    double sr = 0.0, si = 0.0;
#pragma unroll 4
    for (int k = 0; k < B; ++k) {
        cd w = K[(size_t)k * ns];
        cd v = x[(size_t)k * ns + i];
        sr = fma(w.x, v.x, sr);
        sr = fma(-w.y, v.y, sr);
        si = fma(w.x, v.y, si);
        si = fma(w.y, v.x, si);
    }
    out[(size_t)j * ns + i] = cd{sr, si};
}

// The same two products with the Arnoldi normalisation folded in (option "gmres_fused_scale"): the
// input is w, x = w / sqrt(nrm2) is formed on the fly — rounded to a double before it enters the
// product, so x and the product are the bits the separate normalisation kernel gave — and thread
// (i, j) also stores x[j, i] as the new basis vector.  A vanished norm (breakdown) leaves the basis
// vector alone, as cscale_copy_rnorm_kernel does.
__global__ __launch_bounds__(256) void prec_scalar_scaled_kernel(cd* __restrict__ out, const double* __restrict__ Kt,
                                                                 const cd* __restrict__ w, int M, int n,
                                                                 const cd* __restrict__ nrm2, cd* __restrict__ vout) {
    int i = blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (i >= n) return;
    const double hn = sqrt(fmax(nrm2->x, 0.0));
    const bool ok = hn > 0.0;
    const double s = ok ? 1.0 / hn : 0.0;
    const double* K = Kt + (size_t)j * M * n + i;
    double sr = 0.0, si = 0.0;
#pragma unroll 4
    for (int k = 0; k < M; ++k) {
        double kw = K[(size_t)k * n];
        cd v = w[(size_t)k * n + i];
        const double vx = s * v.x, vy = s * v.y;
        if (k == j && ok) vout[(size_t)k * n + i] = cd{vx, vy};
        sr = fma(kw, vx, sr);
        si = fma(kw, vy, si);
    }
    out[(size_t)j * n + i] = cd{sr, si};
}
__global__ __launch_bounds__(256) void prec_cplx_scaled_kernel(cd* __restrict__ out, const cd* __restrict__ Kt,
                                                               const cd* __restrict__ w, int B, int ns,
                                                               const cd* __restrict__ nrm2, cd* __restrict__ vout) {
    int i = blockIdx.x * 256 + threadIdx.x;
    int j = blockIdx.y;
    if (i >= ns) return;
    const double hn = sqrt(fmax(nrm2->x, 0.0));
    const bool ok = hn > 0.0;
    const double s = ok ? 1.0 / hn : 0.0;
    const cd* K = Kt + (size_t)j * B * ns + i;
    double sr = 0.0, si = 0.0;
#pragma unroll 4
    for (int k = 0; k < B; ++k) {
        cd kw = K[(size_t)k * ns];
        cd v = w[(size_t)k * ns + i];
        const double vx = s * v.x, vy = s * v.y;
        if (k == j && ok) vout[(size_t)k * ns + i] = cd{vx, vy};
        sr = fma(kw.x, vx, sr);
        sr = fma(-kw.y, vy, sr);
        si = fma(kw.x, vy, si);
        si = fma(kw.y, vx, si);
    }
    out[(size_t)j * ns + i] = cd{sr, si};
}

// ---- GMRES vector kernels --------------------------------------------------
// h[i] = <V_i, w> = sum conj(V_i) * w, i = blockIdx.x < nv.  Every vector is split over
// MD_SPLIT blocks (one block per vector read 2 x 2 MB through a single CU: 27 us); each
// writes its partial sum, the block that finishes last (ticket counter) adds the partials
// IN INDEX ORDER and resets the counter — one launch, and the result does not depend on the
// order in which the blocks ran.
constexpr int MD_SPLIT = 16;
// Cross-workgroup reduction tails (multidot_kernel, multiaxpy_norm_kernel): every block leaves
// its partial sum and draws a ticket, the block that draws the last one adds the partials up.
// Hand-off per the CDNA4 guide (Guideline 16, "last adder" row of the sc1 table): the partials
// are written by ONE lane with 8-byte agent-scope relaxed atomic stores (write-through), drained
// with s_waitcnt vmcnt(0) before that lane's agent-scope ticket add, and read by the last block
// with agent-scope relaxed atomic loads only after its own add has returned (other waves: after
// the workgroup barrier).  No __threadfence: two of them (an L2 write-back and an L1
// invalidate each, ~3.5 us) were most of these kernels' run time at the sizes of the annular
// systems (multiaxpy_norm 23 us against 8 us for the same update without the norm).
typedef __attribute__((address_space(1))) unsigned long long ann_gu64;
typedef __attribute__((address_space(1))) unsigned int ann_gu32;
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store((ann_gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((ann_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned draw_ticket(unsigned* t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's partial sums have left
    return __hip_atomic_fetch_add((ann_gu32*)t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void multidot_kernel(const cd* __restrict__ V, int64_t ld,
                                                       const cd* __restrict__ w, int64_t n,
                                                       cd* __restrict__ h, cd* __restrict__ partial,
                                                       unsigned* __restrict__ ticket) {
    const cd* v = V + (size_t)blockIdx.x * ld;
    const int64_t per = (n + MD_SPLIT - 1) / MD_SPLIT;
    const int64_t k0 = (int64_t)blockIdx.y * per, k1 = min(n, k0 + per);
    double sr = 0.0, si = 0.0;
    // (unrolled: four pairs of loads in flight per thread; the sums keep their order)
#pragma unroll 4
    for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
        cd a = v[k], b = w[k];
        sr = fma(a.x, b.x, sr);
        sr = fma(a.y, b.y, sr);
        si = fma(a.x, b.y, si);
        si = fma(-a.y, b.x, si);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sr += __shfl_xor(sr, o);
        si += __shfl_xor(si, o);
    }
    __shared__ double red[4][2];
    __shared__ bool last;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[wv][0] = sr;
        red[wv][1] = si;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        double b = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        double* mine = (double*)(partial + (size_t)blockIdx.x * MD_SPLIT + blockIdx.y);
        st_agent(mine, a);
        st_agent(mine + 1, b);
        last = draw_ticket(&ticket[blockIdx.x]) == MD_SPLIT - 1;
        if (last) {        // (this lane's add has returned: every block's partials are out)
            a = 0.0;
            b = 0.0;
            const double* p = (const double*)(partial + (size_t)blockIdx.x * MD_SPLIT);
            double pr[MD_SPLIT], pi[MD_SPLIT];
#pragma unroll
            for (int i = 0; i < MD_SPLIT; ++i) {
                pr[i] = ld_agent(p + 2 * i);
                pi[i] = ld_agent(p + 2 * i + 1);
            }
#pragma unroll
            for (int i = 0; i < MD_SPLIT; ++i) {
                a += pr[i];
                b += pi[i];
            }
            h[blockIdx.x] = cd{a, b};
            __hip_atomic_store((ann_gu32*)&ticket[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// w -= sum_i h[i] V_i
__global__ __launch_bounds__(256) void multiaxpy_kernel(cd* __restrict__ w,
                                                        const cd* __restrict__ V, int64_t ld,
                                                        const cd* __restrict__ h, int nv,
                                                        int64_t n) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    cd acc = w[k];
    for (int i = 0; i < nv; ++i) {
        cd c = h[i];
        cd v = V[(size_t)i * ld + k];
        acc.x -= c.x * v.x - c.y * v.y;
        acc.y -= c.x * v.y + c.y * v.x;
    }
    w[k] = acc;
}
// The same, and ||w||^2 of the result into *nrm2 (the CGS2 sequence ends with this norm: one
// launch less per iteration).  Block partial sums, the last block to finish (ticket) adds
// them in a fixed order — thread t takes partials t, t + 256, ... in order, then a fixed tree.
__global__ __launch_bounds__(256) void multiaxpy_norm_kernel(cd* __restrict__ w, const cd* __restrict__ V,
                                                             int64_t ld, const cd* __restrict__ h, int nv,
                                                             int64_t n, double* __restrict__ partial,
                                                             unsigned* __restrict__ ticket,
                                                             cd* __restrict__ nrm2,
                                                             const cd* __restrict__ col_src = nullptr,
                                                             cd* __restrict__ col_host = nullptr, int ncol = 0) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (k < n) {
        cd acc = w[k];
#pragma unroll 4
        for (int i = 0; i < nv; ++i) {
            cd c = h[i];
            cd v = V[(size_t)i * ld + k];
            acc.x -= c.x * v.x - c.y * v.y;
            acc.y -= c.x * v.y + c.y * v.x;
        }
        w[k] = acc;
        s = fma(acc.x, acc.x, acc.y * acc.y);
    }
    __shared__ double red[4];
    __shared__ bool last;
    auto block_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    double bs = block_sum(s);
    if (threadIdx.x == 0) {
        st_agent(&partial[blockIdx.x], bs);
        last = draw_ticket(ticket) == gridDim.x - 1;
    }
    __syncthreads();       // the last block's other waves read only behind this barrier
    if (!last) return;
    double t = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) t += ld_agent(&partial[i]);
    double tot = block_sum(t);
    if (threadIdx.x == 0) {
        *nrm2 = cd{tot, 0.0};
        __hip_atomic_store((ann_gu32*)ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The iteration's Hessenberg column (both Gram-Schmidt passes' inner products, written by the two
    // multidot launches before this one, and the norm just formed) goes to the host's pinned area from
    // HERE: the last block stores it into the mapped host memory, instead of a copy node of its own
    // behind this kernel (one launch less per iteration; the host waits for the event recorded next).
    if (col_host) {
        for (int i = threadIdx.x; i < ncol; i += 256) {
            cd v = col_src[i];
            if (col_src + i == nrm2) v = cd{tot, 0.0};
            col_host[i] = v;
        }
    }
}
// y = sum_i c[i] V_i  (c on device)
__global__ __launch_bounds__(256) void lincomb_kernel(cd* __restrict__ y, const cd* __restrict__ V,
                                                      int64_t ld, const cd* __restrict__ c, int nv,
                                                      int64_t n) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    cd acc{0.0, 0.0};
    for (int i = 0; i < nv; ++i) {
        cd ci = c[i];
        cd v = V[(size_t)i * ld + k];
        acc.x += ci.x * v.x - ci.y * v.y;
        acc.y += ci.x * v.y + ci.y * v.x;
    }
    y[k] = acc;
}
__global__ __launch_bounds__(256) void cscale_copy_kernel(cd* __restrict__ y,
                                                          const cd* __restrict__ x, int64_t n,
                                                          double s) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    cd v = x[k];
    y[k] = cd{s * v.x, s * v.y};
}

// y = x / sqrt(nrm2->x): the Arnoldi normalisation with the squared norm read where
// multiaxpy_norm_kernel left it — the same sqrt and division the host used to do, so the same
// value, and the whole iteration can be enqueued (or replayed from a graph) without the host.
// A vanished norm (breakdown) leaves y alone, as the host path did.
__global__ __launch_bounds__(256) void cscale_copy_rnorm_kernel(cd* __restrict__ y, const cd* __restrict__ x,
                                                                int64_t n, const cd* __restrict__ nrm2) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double hn = sqrt(fmax(nrm2->x, 0.0));
    if (!(hn > 0.0)) return;
    const double s = 1.0 / hn;
    cd v = x[k];
    y[k] = cd{s * v.x, s * v.y};
}

inline unsigned nb256(int64_t n) { return (unsigned)ceil_div64(n, 256); }

// ---------------------------------------------------------------------------
struct LinOp {
    ipde_ctx* ctx = nullptr;
    int64_t NB = 0;
    virtual int apply(const cd* in, cd* out) = 0;
    virtual int precond(const cd* in, cd* out) = 0;
    // vout = w / sqrt(nrm2->x), out = M^{-1} vout (the solvers override it with one kernel)
    virtual int precond_scaled(const cd* w, const cd* nrm2, cd* vout, cd* out);
    virtual ~LinOp() {}
};

int LinOp::precond_scaled(const cd* w, const cd* nrm2, cd* vout, cd* out) {
    hipLaunchKernelGGL(cscale_copy_rnorm_kernel, dim3(nb256(NB)), dim3(256), 0, ctx->stream, vout, w, NB, nrm2);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return precond(vout, out);
}

struct GmresWork {
    cd* V = nullptr;   // (restart+1, NB)
    cd* w = nullptr;   // NB
    cd* z = nullptr;   // NB
    cd* x = nullptr;   // NB
    cd* t = nullptr;   // NB
    cd* hdev = nullptr;  // 2*(restart+2)
    cd* mdpart = nullptr;     // (restart+2) * MD_SPLIT partial sums of multidot_kernel
    unsigned* mdticket = nullptr;  // (restart+3) counters, zero between launches (last: norm)
    double* nrmpart = nullptr;     // ceil(NB/256) block sums of multiaxpy_norm_kernel
    int restart_cap = 0;
    int64_t nb_cap = 0;
    // one executable graph per inner iteration index j (the launches of an iteration depend on j
    // and on nothing a solve changes: all buffers belong to the solver handle)
    std::vector<hipGraphExec_t> iter_graph;
    long long graph_sig = -1;     // (restart, stream, options) the graphs were captured for
    int graph_failures = 0;       // captures that did not work out; three and the handle stays eager
    // Captures run on a NON-blocking stream of their own (nothing executes during a capture): the
    // context's stream is a blocking one, and while such a stream is capturing every legacy-stream
    // operation of ANY thread — a hipMemset in another solver's set-up, a host framework's
    // default-stream kernel — fails with "would make the legacy stream depend on a capturing
    // blocking stream".  The graph is then launched on the context's stream.
    hipStream_t cap_stream = nullptr;
    // look-ahead (option "gmres_lookahead"): the Hessenberg column of inner iteration j reaches the
    // host through pinned area j & 1, signalled by col_ev[j & 1], while iteration j + 1 is
    // already in the stream
    hipEvent_t col_ev[2] = {nullptr, nullptr};
};

void gmres_drop_graphs(GmresWork& g) {
    for (hipGraphExec_t e : g.iter_graph)
        if (e) hipGraphExecDestroy(e);
    g.iter_graph.clear();
    g.graph_sig = -1;
}

int gmres_reserve(ipde_ctx* ctx, GmresWork& g, int64_t NB, int restart) {
    if (g.V && restart <= g.restart_cap && NB <= g.nb_cap) return IPDE_OK;
    hipStreamSynchronize(ctx->stream);
    gmres_drop_graphs(g);
    for (cd** p : {&g.V, &g.w, &g.z, &g.x, &g.t, &g.hdev})
        if (*p) {
            hipFree(*p);
            *p = nullptr;
        }
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.V, (size_t)(restart + 1) * NB * sizeof(cd)));
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.w, NB * sizeof(cd)));
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.z, NB * sizeof(cd)));
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.x, NB * sizeof(cd)));
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.t, NB * sizeof(cd)));
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.hdev, (size_t)(2 * restart + 8) * sizeof(cd)));
    if (g.mdpart) hipFree(g.mdpart);
    if (g.mdticket) hipFree(g.mdticket);
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.mdpart, (size_t)(restart + 2) * MD_SPLIT * sizeof(cd)));
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.mdticket, (size_t)(restart + 3) * sizeof(unsigned)));
    IPDE_HIP_CHECK(ctx, hipMemsetAsync(g.mdticket, 0, (size_t)(restart + 3) * sizeof(unsigned), ctx->stream));
    if (g.nrmpart) hipFree(g.nrmpart);
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)&g.nrmpart, (size_t)nb256(NB) * sizeof(double)));
    g.restart_cap = restart;
    g.nb_cap = NB;
    return IPDE_OK;
}
void gmres_free(GmresWork& g) {
    gmres_drop_graphs(g);
    if (g.cap_stream) hipStreamDestroy(g.cap_stream);
    g.cap_stream = nullptr;
    for (hipEvent_t& e : g.col_ev) {
        if (e) hipEventDestroy(e);
        e = nullptr;
    }
    for (cd** p : {&g.V, &g.w, &g.z, &g.x, &g.t, &g.hdev, &g.mdpart})
        if (*p) {
            hipFree(*p);
            *p = nullptr;
        }
    if (g.mdticket) hipFree(g.mdticket);
    g.mdticket = nullptr;
    if (g.nrmpart) hipFree(g.nrmpart);
    g.nrmpart = nullptr;
}

struct hc {
    double re, im;
};
inline hc hmul(hc a, hc b) { return hc{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline hc hconj(hc a) { return hc{a.re, -a.im}; }
inline double habs(hc a) { return hypot(a.re, a.im); }

// Right-preconditioned restarted GMRES for A x = b (x0 = 0).  The result is left
// in g.x.  Stops when ||r|| <= tol*||b||.  iters = total inner iterations.
// iters_done > 0: g.x holds the estimate a first cycle left (the device-side cycle) — the solve
// continues with the next restart cycle from it.
int gmres_solve(LinOp& op, GmresWork& g, const cd* b, double tol, int maxiter, int restart,
                int* iters_out, double* resid_out, int iters_done = 0) {
    ipde_ctx* ctx = op.ctx;
    const int64_t NB = op.NB;
    hipStream_t st = ctx->stream;
    if (restart < 1) restart = 1;
    if (maxiter < 1) maxiter = 1;
    if (restart > maxiter) restart = maxiter;   // a cycle never runs longer than maxiter
    // one cycle's Hessenberg column travels through the context's pinned buffer
    IPDE_CHECK_ARG(ctx, (size_t)(2 * restart + 4) * sizeof(cd) + 16 <= ctx->h_pinned_bytes);   // (last 8 bytes: dense.hip)
    IPDE_CHECK_ARG(ctx, tol > 0.0);
    IPDE_TRY(gmres_reserve(ctx, g, NB, restart));
    if (iters_done == 0) IPDE_HIP_CHECK(ctx, hipMemsetAsync(g.x, 0, NB * sizeof(cd), st));
    cd* hp = (cd*)ctx->h_pinned;  // pinned: [restart+2] entries used per transfer
    cd* hp_dev = hp;              // the same buffer as the device sees it (hipHostMalloc memory is mapped)
    {
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, hp, 0) == hipSuccess && d) hp_dev = (cd*)d;
        else (void)hipGetLastError();
    }
    // ||b||.  With the look-ahead (default) the host does NOT wait for it: v_0 = b / ||b|| is formed on
    // the device from the device-resident squared norm (the same sqrt and division, so the same
    // bits), the squared norm travels to a pinned slot of its own, and the host first reads it when
    // it waits for the first Hessenberg column anyway — one synchronisation (~50 us of idle GPU per
    // annular solve) less.  The launch-per-wait paths and a continued solve read it at once.
    hipLaunchKernelGGL(multidot_kernel, dim3(1, MD_SPLIT), dim3(256), 0, st, b, NB, b, NB, g.hdev, g.mdpart,
                       g.mdticket);
    const size_t hstride0 = (size_t)(2 * restart + 4);
    const bool defer_bnorm = ctx->opt_gmres_lookahead && !(ctx->opt_gmres_graphs && st != nullptr && g.graph_failures < 3) &&
                             iters_done == 0 && ((3 * hstride0 + 1) * sizeof(cd) + 16 <= ctx->h_pinned_bytes);
    cd* hp_b = defer_bnorm ? hp + 3 * hstride0 : hp;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(hp_b, g.hdev, sizeof(cd), hipMemcpyDeviceToHost, st));
    double bnorm = -1.0;      // < 0: not read yet
    if (defer_bnorm) {
        hipLaunchKernelGGL(cscale_copy_rnorm_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.V, b, NB, (const cd*)g.hdev);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
    } else {
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
        bnorm = sqrt(hp[0].x);
    }
    int iters = iters_done;
    double resid = 0.0;
    int status = IPDE_OK;
    if (!defer_bnorm && !(bnorm > 0.0)) {
        *iters_out = 0;
        *resid_out = 0.0;
        return IPDE_OK;
    }
    // graphs of the inner iterations: valid for one (restart, stream, option set); the legacy
    // default stream cannot be captured
    bool use_graphs = ctx->opt_gmres_graphs && st != nullptr && g.graph_failures < 3;
    // look-ahead of one inner iteration (default): three pinned areas — columns of even / odd
    // iterations, the cycle's y — and an event per area
    const bool fused = ctx->opt_gmres_fused_scale != 0;
    const size_t hstride = (size_t)(2 * restart + 4);
    bool lookahead = ctx->opt_gmres_lookahead && !use_graphs &&
                     (3 * hstride * sizeof(cd) + 16 <= ctx->h_pinned_bytes);
    for (int i = 0; lookahead && i < 2; ++i)
        if (!g.col_ev[i] && hipEventCreateWithFlags(&g.col_ev[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            g.col_ev[i] = nullptr;
            lookahead = false;
        }
    if (bnorm < 0.0 && !lookahead) {      // (no look-ahead after all: read ||b|| now)
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
        bnorm = sqrt(hp_b[0].x);
        if (!(bnorm > 0.0)) {
            *iters_out = 0;
            *resid_out = 0.0;
            return IPDE_OK;
        }
    }
    {
        const long long sig = ((long long)restart << 8) ^ ((long long)(uintptr_t)st << 20) ^
                              (ctx->opt_annular_grouped ? 1 : 0) ^ (ctx->opt_annular_fused_fft ? 2 : 0) ^
                              (ctx->opt_gmres_fused_scale ? 8 : 0);
        if (g.graph_sig != sig) {
            gmres_drop_graphs(g);
            g.graph_sig = sig;
        }
    }
    std::vector<hc> H((size_t)(restart + 1) * restart), cs(restart), sn(restart), gv(restart + 1);
    bool converged = false;
    bool first_cycle = iters_done == 0;
    while (!converged && iters < maxiter) {
        // r = b - A x   (x = 0 in the first cycle)
        double beta;
        if (first_cycle) {
            if (!defer_bnorm)
                hipLaunchKernelGGL(cscale_copy_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.V, b, NB,
                                   1.0 / bnorm);
            beta = bnorm;         // (deferred: filled in at the first column, below)
            first_cycle = false;
        } else {
            IPDE_TRY(op.apply(g.x, g.w));
            hipLaunchKernelGGL(cscale_copy_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.t, b, NB, 1.0);
            hipLaunchKernelGGL(caxpy_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.t,
                               (const cd*)g.w, NB, -1.0);
            hipLaunchKernelGGL(multidot_kernel, dim3(1, MD_SPLIT), dim3(256), 0, st, (const cd*)g.t, NB,
                               (const cd*)g.t, NB, g.hdev, g.mdpart, g.mdticket);
            IPDE_HIP_CHECK(ctx, hipMemcpyAsync(hp, g.hdev, sizeof(cd), hipMemcpyDeviceToHost, st));
            IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
            beta = sqrt(hp[0].x);
            if (beta <= tol * bnorm) {
                resid = beta / bnorm;
                converged = true;
                break;
            }
            hipLaunchKernelGGL(cscale_copy_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.V,
                               (const cd*)g.t, NB, 1.0 / beta);
        }
        for (auto& v : gv) v = hc{0.0, 0.0};
        gv[0] = hc{beta, 0.0};
        int j = 0;
        int enq = 0;                                   // inner iterations of this cycle already in the stream
        double r_prev = bnorm < 0.0 ? 1.0 : beta / bnorm, r_prev2 = 0.0;   // residual history for the look-ahead's guess
        for (; j < restart && iters < maxiter; ++j) {
            // Inner iteration j, everything up to the host's look at the new Hessenberg column:
            // preconditioner, operator, CGS2, the column's way to pinned memory, and the
            // normalisation of v_{j+1} (from the device-resident norm) — ~20 launches, no host value
            // in between.  Option "gmres_graphs": the sequence is captured once per j into a hipGraph
            // and replayed.  Measured (profiles/r02_gmres_graph_ab.txt): no gain — warm Poisson 2048^2
            // 8.75 / 9.11 ms eager vs 9.02 / 8.94 replayed, 3-body Stokes 20.4 / 21.7 vs 20.4 / 20.7;
            // the critical stream is bound by its own chain of 5-25 us kernels (the outer Stokes body:
            // 21 kernels, 255 us per iteration), not by the host's launch rate.  Off by default.
            auto hpj_dev = [&](cd* host_ptr) -> cd* { return hp_dev + (host_ptr - hp); };
            auto enqueue = [&](hipStream_t st, int j, cd* hpj, hipEvent_t done) -> int {
                cd* vj = g.V + (size_t)j * NB;
                // v_j = w / ||w|| of the previous iteration is formed INSIDE the preconditioner's
                // kernel (fused: no normalisation launch at the end of an iteration); v_0 is the
                // cycle's scaled residual
                if (fused && j > 0)
                    IPDE_TRY(op.precond_scaled(g.w, g.hdev + j, vj, g.z));
                else
                    IPDE_TRY(op.precond(vj, g.z));
                IPDE_TRY(op.apply(g.z, g.w));
                // CGS2
                cd* h1 = g.hdev;
                cd* h2 = g.hdev + (restart + 2);
                hipLaunchKernelGGL(multidot_kernel, dim3(j + 1, MD_SPLIT), dim3(256), 0, st, (const cd*)g.V, NB,
                                   (const cd*)g.w, NB, h1, g.mdpart, g.mdticket);
                hipLaunchKernelGGL(multiaxpy_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.w,
                                   (const cd*)g.V, NB, (const cd*)h1, j + 1, NB);
                hipLaunchKernelGGL(multidot_kernel, dim3(j + 1, MD_SPLIT), dim3(256), 0, st, (const cd*)g.V, NB,
                                   (const cd*)g.w, NB, h2, g.mdpart, g.mdticket);
                hipLaunchKernelGGL(multiaxpy_norm_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.w, (const cd*)g.V, NB,
                                   (const cd*)h2, j + 1, NB, g.nrmpart, g.mdticket + (restart + 2),
                                   h1 + (j + 1), (const cd*)g.hdev, hpj_dev(hpj), 2 * restart + 4);
                IPDE_HIP_CHECK(ctx, hipGetLastError());
                if (done) IPDE_HIP_CHECK(ctx, hipEventRecord(done, st));
                if (!fused) {
                    hipLaunchKernelGGL(cscale_copy_rnorm_kernel, dim3(nb256(NB)), dim3(256), 0, st,
                                       g.V + (size_t)(j + 1) * NB, (const cd*)g.w, NB, (const cd*)(h1 + (j + 1)));
                    IPDE_HIP_CHECK(ctx, hipGetLastError());
                }
                return IPDE_OK;
            };
            const cd* hpj = hp;
            if (lookahead) {
                // Iteration j is in the stream already if the previous pass put it there.  Then
                // iteration j + 1 goes in BEFORE the host waits for column j — unless the residual
                // history says column j will end the solve (an iteration enqueued in vain costs more
                // than the wait it hides; either way the results are the same bits: a surplus
                // iteration writes only buffers the solution update does not read).
                if (enq <= j) {
                    IPDE_TRY(enqueue(st, j, hp + (size_t)(j & 1) * hstride, g.col_ev[j & 1]));
                    enq = j + 1;
                }
                const double rho = (r_prev2 > 0.0) ? fmin(1.0, r_prev / r_prev2) : 1.0;
                if (j + 1 < restart && iters + 1 < maxiter && r_prev * rho > 10.0 * tol) {
                    IPDE_TRY(enqueue(st, j + 1, hp + (size_t)((j + 1) & 1) * hstride, g.col_ev[(j + 1) & 1]));
                    enq = j + 2;
                }
                IPDE_HIP_CHECK(ctx, hipEventSynchronize(g.col_ev[j & 1]));
                hpj = hp + (size_t)(j & 1) * hstride;
                if (bnorm < 0.0) {      // the deferred ||b||: its copy precedes column 0's in the stream
                    bnorm = sqrt(hp_b[0].x);
                    if (!(bnorm > 0.0)) {     // b = 0: x = 0 (the iteration that ran wrote nothing a caller reads)
                        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
                        IPDE_HIP_CHECK(ctx, hipMemsetAsync(g.x, 0, NB * sizeof(cd), st));
                        *iters_out = 0;
                        *resid_out = 0.0;
                        return IPDE_OK;
                    }
                    gv[0] = hc{bnorm, 0.0};
                }
            } else {
            bool replayed = false;
            if (use_graphs) {
                if ((int)g.iter_graph.size() <= j) g.iter_graph.resize(j + 1, nullptr);
                hipGraphExec_t ex = g.iter_graph[j];
                if (!ex) {
                    bool ok = g.cap_stream != nullptr ||
                              hipStreamCreateWithFlags(&g.cap_stream, hipStreamNonBlocking) == hipSuccess;
                    if (ok) ok = hipStreamBeginCapture(g.cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        ctx->stream = g.cap_stream;        // the operator's launches follow the context
                        const int s_enq = enqueue(g.cap_stream, j, hp, nullptr);
                        ctx->stream = st;
                        hipGraph_t graph = nullptr;
                        ok = hipStreamEndCapture(g.cap_stream, &graph) == hipSuccess && s_enq == IPDE_OK &&
                             graph != nullptr;
                        if (ok) ok = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) == hipSuccess;
                        if (graph) hipGraphDestroy(graph);
                    }
                    if (ok) {
                        g.iter_graph[j] = ex;
                    } else {
                        (void)hipGetLastError();       // the failed capture's error is not the solve's
                        ex = nullptr;
                        use_graphs = false;            // this solve goes on eagerly
                        ++g.graph_failures;
                    }
                }
                if (ex) {
                    IPDE_HIP_CHECK(ctx, hipGraphLaunch(ex, st));
                    replayed = true;
                }
            }
            if (!replayed) IPDE_TRY(enqueue(st, j, hp, nullptr));
            IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
            }
            hc* col = &H[(size_t)j * (restart + 1)];
            for (int i = 0; i <= j; ++i)
                col[i] = hc{hpj[i].x + hpj[restart + 2 + i].x, hpj[i].y + hpj[restart + 2 + i].y};
            double hn = sqrt(fmax(hpj[j + 1].x, 0.0));
            col[j + 1] = hc{hn, 0.0};
            // Givens
            for (int i = 0; i < j; ++i) {
                hc a = col[i], bb = col[i + 1];
                hc t1 = hmul(hconj(cs[i]), a);
                hc t2 = hmul(hconj(sn[i]), bb);
                hc n1{t1.re + t2.re, t1.im + t2.im};
                hc t3 = hmul(sn[i], a);
                hc t4 = hmul(cs[i], bb);
                hc n2{-t3.re + t4.re, -t3.im + t4.im};
                col[i] = n1;
                col[i + 1] = n2;
            }
            {
                hc a = col[j], bb = col[j + 1];
                double den = hypot(habs(a), habs(bb));
                if (den == 0.0) den = 1.0;
                cs[j] = hc{a.re / den, a.im / den};
                sn[j] = hc{bb.re / den, bb.im / den};
                col[j] = hc{den, 0.0};
                col[j + 1] = hc{0.0, 0.0};
                hc gj = gv[j];
                gv[j] = hmul(hconj(cs[j]), gj);
                hc tmp = hmul(sn[j], gj);
                gv[j + 1] = hc{-tmp.re, -tmp.im};
            }
            ++iters;
            resid = habs(gv[j + 1]) / bnorm;
            r_prev2 = r_prev;
            r_prev = resid;
            if (resid <= tol || hn == 0.0) {
                converged = resid <= tol || hn == 0.0;
                ++j;
                break;
            }
        }
        // solve the j x j triangular system, x += M^{-1} V y
        int m = j;
        std::vector<hc> y(m);
        for (int i = m - 1; i >= 0; --i) {
            hc s = gv[i];
            for (int k = i + 1; k < m; ++k) {
                hc t = hmul(H[(size_t)k * (restart + 1) + i], y[k]);
                s.re -= t.re;
                s.im -= t.im;
            }
            hc d = H[(size_t)i * (restart + 1) + i];
            double dd = d.re * d.re + d.im * d.im;
            if (dd == 0.0) {
                y[i] = hc{0.0, 0.0};
            } else {
                y[i] = hc{(s.re * d.re + s.im * d.im) / dd, (s.im * d.re - s.re * d.im) / dd};
            }
        }
        // (look-ahead: a surplus iteration's column may still be on its way into areas 0 / 1)
        cd* hy = lookahead ? hp + 2 * hstride : hp;
        for (int i = 0; i < m; ++i) hy[i] = cd{y[i].re, y[i].im};
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(g.hdev, hy, (size_t)m * sizeof(cd),
                                           hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lincomb_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.t, (const cd*)g.V, NB,
                           (const cd*)g.hdev, m, NB);
        IPDE_TRY(op.precond(g.t, g.z));
        hipLaunchKernelGGL(caxpy_kernel, dim3(nb256(NB)), dim3(256), 0, st, g.x, (const cd*)g.z, NB,
                           1.0);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        // the pinned buffer is reused by the next cycle
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    if (!converged) status = IPDE_ERR_NOCONV;
    *iters_out = iters;
    *resid_out = resid;
    return status;
}

int upload(ipde_ctx* ctx, double** d, const double* h, size_t n) {
    IPDE_HIP_CHECK(ctx, hipMalloc((void**)d, n * sizeof(double)));
    IPDE_HIP_CHECK(ctx, hipMemcpy(*d, h, n * sizeof(double), hipMemcpyHostToDevice));
    return IPDE_OK;
}

int set_field(ipde_ctx* ctx, int loc, double* dst, const double* src, size_t n) {
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, n * sizeof(double),
                                       loc == IPDE_HOST ? hipMemcpyHostToDevice
                                                        : hipMemcpyDeviceToDevice,
                                       ctx->stream));
    if (loc == IPDE_HOST) IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return IPDE_OK;
}

}  // namespace

// ===========================================================================
// A stage of the scalar operator is  rows -> inverse FFT -> times a metric field / n -> forward FFT.
// For power-of-two n up to 4096 the three launches (two batched rocFFT transforms with a pointwise
// kernel between them, each 5-10 us for 20-40 rows and a dependent kernel boundary apiece) are ONE
// kernel: a workgroup takes a row, both transforms run in its registers and LDS (fft_core.h, the
// Stockham passes of the 2-D pipeline), the field is applied in between, in place.
template <int N>
__global__ __launch_bounds__(fftcore::Cfg<N>::T) void fft_pair_kernel(cd* __restrict__ x,
                                                                        const fftcore::cd* __restrict__ tw,
                                                                        const double* __restrict__ F0,
                                                                        const double* __restrict__ F1, int rows0,
                                                                        double s) {
    using G = fftcore::Cfg<N>;
    constexpr int T = G::T, P = G::P;
    extern __shared__ double2 pair_lds[];
    fftcore::cd* buf = (fftcore::cd*)pair_lds;
    const int t = threadIdx.x, row = blockIdx.x;
    fftcore::cd* r = (fftcore::cd*)x + (size_t)row * N;
    const double* F = row < rows0 ? F0 + (size_t)row * N : F1 + (size_t)(row - rows0) * N;
    fftcore::cd v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = r[t + T * q];
    fftcore::fft_regs<N, +1, (T == 64)>(v, t, tw, buf);
    fftcore::lds_sync<(T == 64)>();
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const double f = s * F[t + T * q];
        v[q] = fftcore::cd{v[q].x * f, v[q].y * f};
    }
    fftcore::fft_regs<N, -1, (T == 64)>(v, t, tw, buf);
#pragma unroll
    for (int q = 0; q < P; ++q) r[t + T * q] = v[q];
}

template <int N>
int launch_fft_pair(ipde_ctx* ctx, cd* x, const void* tw, int rows, const double* F0, const double* F1, int rows0,
                    double s) {
    const size_t lds = (size_t)fftcore::lds_slots<N>() * sizeof(fftcore::cd);
    if (lds > 48 * 1024)
        IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fft_pair_kernel<N>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fft_pair_kernel<N>, dim3((unsigned)rows), dim3(fftcore::Cfg<N>::T), lds, ctx->stream, x,
                       (const fftcore::cd*)tw, F0, F1, rows0, s);
    return IPDE_OK;
}

// n = 8192 (BASELINE configs[3]'s boundary): the same stage as two 4096-point halves and one radix-2 level on either
// side.  512 threads: group g = t / 256 owns the samples of parity g.  Inverse transform, decimation in time:
// E = FFT(even samples), O = FFT(odd samples), X[k] = E[k] + w^k O[k], X[k + 4096] = E[k] - w^k O[k] — thread
// (0, tt) and thread (1, tt) hold E and O at the SAME k = tt + 256 q, so the level is a pairwise exchange through
// LDS, after which group g holds the physical samples j = 4096 g + k.  Forward transform, decimation in frequency:
// a[j] = y[j] + y[j + 4096] -> X[2 m], b[j] = (y[j] - y[j + 4096]) w'^j -> X[2 m + 1]: the same exchange, then a
// 4096-point transform per group, and group g stores the outputs of parity g.  tw: the 4096-point table.
__global__ __launch_bounds__(512) void fft_pair8192_kernel(cd* __restrict__ x, const fftcore::cd* __restrict__ tw,
                                                           const double* __restrict__ F0,
                                                           const double* __restrict__ F1, int rows0, double s) {
    constexpr int N = 8192, H = 4096, T = 256, P = 16;
    extern __shared__ double2 pair_lds[];
    const int t = threadIdx.x, g = t >> 8, tt = t & 255, row = blockIdx.x;
    fftcore::cd* buf = (fftcore::cd*)pair_lds + (size_t)g * fftcore::lds_slots<H>();
    fftcore::cd* other = (fftcore::cd*)pair_lds + (size_t)(1 - g) * fftcore::lds_slots<H>();
    fftcore::cd* r = (fftcore::cd*)x + (size_t)row * N;
    const double* F = row < rows0 ? F0 + (size_t)row * N : F1 + (size_t)(row - rows0) * N;
    fftcore::cd v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = r[2 * (tt + T * q) + g];
    fftcore::fft_regs<H, +1, false>(v, tt, tw, buf);
    // w^k = e^{+2 pi i k / 8192}, k = tt + 256 q: e^{i pi tt / 4096} (e^{i pi / 16})^q
    double sn, cs;
    sincospi((double)tt / 4096.0, &sn, &cs);
    const fftcore::cd w0{cs, sn};
    sincospi(1.0 / 16.0, &sn, &cs);
    const fftcore::cd wst{cs, sn};
    {
        fftcore::cd w = w0;
        __syncthreads();                                   // (the transforms' last exchange has been read)
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (g == 1) v[q] = fftcore::cmul(v[q], w);   // w^k O[k]
            buf[fftcore::padpos(tt + T * q)] = v[q];
            w = fftcore::cmul(w, wst);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const fftcore::cd o = other[fftcore::padpos(tt + T * q)];
            v[q] = g == 0 ? v[q] + o : o - v[q];          // E + w O  |  E - w O
        }
    }
    // physical sample j = 4096 g + tt + 256 q: the metric field, then the forward transform
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const double f = s * F[H * g + tt + T * q];
        v[q] = fftcore::cd{v[q].x * f, v[q].y * f};
    }
    {
        __syncthreads();                                   // (everybody has read the first exchange)
#pragma unroll
        for (int q = 0; q < P; ++q) buf[fftcore::padpos(tt + T * q)] = v[q];
        __syncthreads();
        fftcore::cd w = fftcore::cd{w0.x, -w0.y};         // w'^j = e^{-2 pi i j / 8192}, j = tt + 256 q
        const fftcore::cd wstc{wst.x, -wst.y};
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const fftcore::cd o = other[fftcore::padpos(tt + T * q)];
            v[q] = g == 0 ? v[q] + o : fftcore::cmul(o - v[q], w);      // y[j] + y[j + H]  |  (y[j] - y[j + H]) w'^j
            w = fftcore::cmul(w, wstc);
        }
        __syncthreads();                                   // (before the transforms' exchanges reuse the buffers)
    }
    fftcore::fft_regs<H, -1, false>(v, tt, tw, buf);
#pragma unroll
    for (int q = 0; q < P; ++q) r[2 * (tt + T * q) + g] = v[q];
}

int launch_fft_pair8192(ipde_ctx* ctx, cd* x, const void* tw, int rows, const double* F0, const double* F1, int rows0,
                        double s) {
    const size_t lds = (size_t)2 * fftcore::lds_slots<4096>() * sizeof(fftcore::cd);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fft_pair8192_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds));
    hipLaunchKernelGGL(fft_pair8192_kernel, dim3((unsigned)rows), dim3(512), lds, ctx->stream, x, (const fftcore::cd*)tw,
                       F0, F1, rows0, s);
    return IPDE_OK;
}

inline bool fft_pair_supported(int n) { return n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192; }

int fft_pair(ipde_ctx* ctx, int n, cd* x, const void* tw, int rows, const double* F0, const double* F1, int rows0,
             double s) {
    switch (n) {
        case 512: return launch_fft_pair<512>(ctx, x, tw, rows, F0, F1, rows0, s);
        case 1024: return launch_fft_pair<1024>(ctx, x, tw, rows, F0, F1, rows0, s);
        case 2048: return launch_fft_pair<2048>(ctx, x, tw, rows, F0, F1, rows0, s);
        case 4096: return launch_fft_pair<4096>(ctx, x, tw, rows, F0, F1, rows0, s);
        case 8192: return launch_fft_pair8192(ctx, x, tw, rows, F0, F1, rows0, s);
    }
    return IPDE_ERR_INVALID;
}

#include "annular_gmres_persist.h"

// persistent cycles in flight on this process's devices: a grid whose workgroups are only partly
// resident next to other partly resident ones would wait for ever (until its time-out), so at most
// PG_MAX_IN_FLIGHT run side by side (64 workgroups of one per CU each: 192 of the 256 CUs)
std::atomic<int> g_pg_in_flight{0};
constexpr int PG_MAX_IN_FLIGHT = 3;

// ===========================================================================
// scalar (modified Helmholtz / Poisson) annular solver
struct ipde_annular_scalar : public LinOp {
    int M = 0, n = 0;
    double k2 = 0.0;
    double *R01 = nullptr, *R12 = nullptr, *D01 = nullptr, *D12 = nullptr, *R02 = nullptr;
    double* Bmat = nullptr;  // (M, M): [k^2 R02; ibc; obc]
    double* Kt = nullptr;    // [M][M][n]
    cd* iks = nullptr;       // (n)
    double *psi1 = nullptr, *ipsi1 = nullptr, *ipsi2 = nullptr;
    cd *T = nullptr, *U = nullptr;  // 2(M-1) x n work
    cd *bvec = nullptr;
    double* rwork = nullptr;        // (M, n) real work
    cd *hin = nullptr, *hout = nullptr;  // host-call staging
    GmresWork gw;
    bool have_geom = false;
    void* tw = nullptr;      // exp(-2 pi i m / n), m < n: the fused transform pairs (power-of-two n <= 4096)
    // device-side GMRES cycle (annular_gmres_persist.h): [counter, time-out word, pad | result (4) |
    // partial norms (G) | partial inner products (G, PG_RMAX + 2)]
    void* pg_ws = nullptr;
    int pg_G = 0;
    bool pg_disabled = false;
    int persistent_cycle(const cd* b, double tol, int maxiter, int restart, int* iters, double* resid,
                         int* converged);

    int apply(const cd* uh, cd* out) override {
        hipStream_t st = ctx->stream;
        const int m1 = M - 1, m2 = M - 2;
        const bool fused = tw != nullptr && ctx->opt_annular_fused_fft;
        dim3 b(256);
        // T1 = R01 (uh * iks), T2 = D01 uh
        {
            MixcBatch Bq{};
            Bq.j[0] = MixcJob{T, m1, 1, {MixcTerm{R01, uh, iks, M}, MixcTerm{}}, nullptr, 0};
            Bq.j[1] = MixcJob{T + (size_t)m1 * n, m1, 1, {MixcTerm{D01, uh, nullptr, M}, MixcTerm{}}, nullptr, 0};
            hipLaunchKernelGGL(mixc_fused_kernel, dim3(nb256(n), m1, 2), b, 0, st, Bq, n);
        }
        if (fused) {
            IPDE_TRY(fft_pair(ctx, n, T, tw, 2 * m1, ipsi1, psi1, m1, 1.0 / n));
        } else {
            IPDE_TRY(ipde_fft1_exec(ctx, 2 * m1, n, +1, T, U));
            hipLaunchKernelGGL(cscale_field2_kernel, dim3(nb256((int64_t)2 * m1 * n)), b, 0, st, U, m1, m1, n,
                               (const double*)ipsi1, (const double*)psi1, 1.0 / n);
            IPDE_TRY(ipde_fft1_exec(ctx, 2 * m1, n, -1, U, T));
        }
        // S = R12 (T1 * iks) + D12 T2   -> U[0:m2]
        {
            MixcBatch Bq{};
            Bq.j[0] = MixcJob{U, m2, 2, {MixcTerm{R12, T, iks, m1}, MixcTerm{D12, T + (size_t)m1 * n, nullptr, m1}},
                              nullptr, 0};
            hipLaunchKernelGGL(mixc_fused_kernel, dim3(nb256(n), m2, 1), b, 0, st, Bq, n);
        }
        if (fused) {
            IPDE_TRY(fft_pair(ctx, n, U, tw, m2, ipsi2, ipsi2, m2, 1.0 / n));
        } else {
            IPDE_TRY(ipde_fft1_exec(ctx, m2, n, +1, U, T));
            hipLaunchKernelGGL(cscale_field_kernel, dim3(nb256((int64_t)m2 * n)), b, 0, st, T, m2, n,
                               (const double*)ipsi2, 1.0 / n);
            IPDE_TRY(ipde_fft1_exec(ctx, m2, n, -1, T, U));
        }
        // out = B uh ; out[0:m2] -= luh
        {
            MixcBatch Bq{};
            Bq.j[0] = MixcJob{out, M, 1, {MixcTerm{Bmat, uh, nullptr, M}, MixcTerm{}}, U, m2};
            hipLaunchKernelGGL(mixc_fused_kernel, dim3(nb256(n), M, 1), b, 0, st, Bq, n);
        }
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        return IPDE_OK;
    }
    int precond(const cd* in, cd* out) override {
        hipLaunchKernelGGL(prec_scalar_kernel, dim3(nb256(n), M), dim3(256), 0, ctx->stream, out,
                           (const double*)Kt, in, M, n);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        return IPDE_OK;
    }
    int precond_scaled(const cd* w, const cd* nrm2, cd* vout, cd* out) override {
        hipLaunchKernelGGL(prec_scalar_scaled_kernel, dim3(nb256(n), M), dim3(256), 0, ctx->stream, out,
                           (const double*)Kt, w, M, n, nrm2, vout);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        return IPDE_OK;
    }
    ~ipde_annular_scalar() override {
        for (double** p : {&R01, &R12, &D01, &D12, &R02, &Bmat, &Kt, &psi1, &ipsi1, &ipsi2, &rwork})
            if (*p) hipFree(*p);
        for (cd** p : {&iks, &T, &U, &bvec, &hin, &hout})
            if (*p) hipFree(*p);
        if (tw) hipFree(tw);
        if (pg_ws) hipFree(pg_ws);
        gmres_free(gw);
    }
};

template <int N, int MR>
int launch_gmres_persistent_mr(ipde_ctx* ctx, const PgArgs& A) {
    const size_t lds = (size_t)A.lds_cd * sizeof(cd);
    IPDE_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)gmres_scalar_persistent<N, MR>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gmres_scalar_persistent<N, MR>), dim3((unsigned)A.G), dim3(fftcore::Cfg<N>::T), lds,
                       ctx->stream, A);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}
template <int N>
int launch_gmres_persistent(ipde_ctx* ctx, const PgArgs& A) {
    if (A.M <= 4 * PG_RS) return launch_gmres_persistent_mr<N, 4>(ctx, A);
    if (A.M <= 5 * PG_RS) return launch_gmres_persistent_mr<N, 5>(ctx, A);
    return launch_gmres_persistent_mr<N, PG_MR_MAX>(ctx, A);
}

// One GMRES cycle (x0 = 0, at most `restart` inner iterations) in one launch; the solution estimate
// is left in gw.x.  *converged = 0: the cycle ran out (the caller goes on with the launch-per-stage
// cycles from gw.x); returns IPDE_ERR_INVALID when the configuration is not eligible or the kernel
// timed out (the caller then runs the whole solve the launch-per-stage way).
int ipde_annular_scalar::persistent_cycle(const cd* b, double tol, int maxiter, int restart, int* iters,
                                          double* resid, int* converged) {
    if (pg_disabled || !tw || !ctx->opt_gmres_persistent || !fft_pair_supported(n) || n > 4096 || M < 3 ||
        M > PG_RS * PG_MR_MAX || restart < 1 || restart > PG_RMAX)
        return IPDE_ERR_INVALID;
    const int TT = n >= 4096 ? 256 : n >= 2048 ? 128 : 64, CPB = TT / PG_RS, NW = (TT + 63) / 64;
    const int m1 = M - 1;
    const int G = std::max((n + CPB - 1) / CPB, 2 * m1);
    const int fslots = n + n / 16;
    PgArgs A{};
    A.off_colA = std::max(fslots, 2 * m1 * CPB);
    A.off_red = A.off_colA + M * CPB;
    A.off_hs = A.off_red + NW * (PG_RMAX + 2);
    A.off_H = A.off_hs + 2 * (PG_RMAX + 2);
    A.off_g = A.off_H + (PG_RMAX + 1) * PG_RMAX;
    A.off_mat = A.off_g + 4 * (PG_RMAX + 1) + 4;
    A.lds_cd = A.off_mat + (2 * m1 * M + 2 * (M - 2) * m1 + M * M + 1) / 2;
    if ((size_t)A.lds_cd * sizeof(cd) > 160 * 1024 || G > 64) return IPDE_ERR_INVALID;
    if (g_pg_in_flight.fetch_add(1) >= PG_MAX_IN_FLIGHT) {
        g_pg_in_flight.fetch_sub(1);
        return IPDE_ERR_INVALID;
    }
    struct Release {
        ~Release() { g_pg_in_flight.fetch_sub(1); }
    } release;
    hipStream_t st = ctx->stream;
    IPDE_TRY(gmres_reserve(ctx, gw, NB, restart));
    const size_t ws_bytes = 256 + (size_t)G * sizeof(double) + 16 + (size_t)G * (PG_RMAX + 2) * sizeof(cd);
    if (!pg_ws || pg_G != G) {
        if (pg_ws) IPDE_HIP_CHECK(ctx, hipFree(pg_ws));
        pg_ws = nullptr;
        IPDE_HIP_CHECK(ctx, hipMalloc(&pg_ws, ws_bytes));
        pg_G = G;
    }
    IPDE_HIP_CHECK(ctx, hipMemsetAsync(pg_ws, 0, 256, st));    // counter, time-out word, result, stage profile
    A.M = M;
    A.restart = restart;
    A.maxiter = maxiter;
    A.G = G;
    A.tol = tol;
    A.R01 = R01;
    A.R12 = R12;
    A.D01 = D01;
    A.D12 = D12;
    A.Bmat = Bmat;
    A.Kt = Kt;
    A.iks = iks;
    A.psi1 = psi1;
    A.ipsi1 = ipsi1;
    A.ipsi2 = ipsi2;
    A.tw = (const fftcore::cd*)tw;
    A.T = this->T;
    A.U = U;
    A.V = gw.V;
    A.b = b;
    A.x = gw.x;
    A.counter = (unsigned*)pg_ws;
    A.result = (double*)((char*)pg_ws + 16);
    A.partn = (double*)((char*)pg_ws + 256);
    A.part = (cd*)((char*)pg_ws + 256 + (size_t)((G * sizeof(double) + 15) / 16) * 16);
    int s;
    switch (n) {
        case 512: s = launch_gmres_persistent<512>(ctx, A); break;
        case 1024: s = launch_gmres_persistent<1024>(ctx, A); break;
        case 2048: s = launch_gmres_persistent<2048>(ctx, A); break;
        default: s = launch_gmres_persistent<4096>(ctx, A); break;
    }
    IPDE_TRY(s);
    // the result record (and the time-out word in front of it) through the context's pinned buffer
    double* hp = ctx->h_pinned;
    static const bool profile = getenv("IPDE_PG_PROFILE") != nullptr;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(hp, pg_ws, profile ? 16 + 16 * 8 : 48, hipMemcpyDeviceToHost, st));
    IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (profile) {      // 100 MHz ticks of workgroup 0 per stage kind, summed over the cycle's iterations
        const double* tk = hp + 2 + 8;
        fprintf(stderr, "ipde_hip: device-side GMRES cycle, %d iterations, us per iteration: prec+mix %.1f, pairs(T) %.1f, "
                        "mix %.1f, pairs(U) %.1f, w+dots %.1f, axpy+dots %.1f, axpy+norm %.1f, barriers %.1f\n",
                (int)hp[2], tk[0] / 100 / hp[2], tk[1] / 100 / hp[2], tk[2] / 100 / hp[2], tk[3] / 100 / hp[2],
                tk[4] / 100 / hp[2], tk[5] / 100 / hp[2], tk[6] / 100 / hp[2], tk[7] / 100 / hp[2]);
    }
    const unsigned timed_out = ((const unsigned*)hp)[1];
    if (timed_out) {
        fprintf(stderr, "ipde_hip: the device-side GMRES cycle timed out at a grid barrier (its %d workgroups were "
                        "not resident together); this solver goes back to the launch-per-stage cycle\n", G);
        pg_disabled = true;
        return IPDE_ERR_INVALID;
    }
    *iters = (int)hp[2];
    *resid = hp[3];
    *converged = hp[4] != 0.0;
    return IPDE_OK;
}

extern "C" int ipde_annular_scalar_create(ipde_ctx* ctx, int M, int n, double helmholtz_k,
                                          const double* R01, const double* R12, const double* R02,
                                          const double* D01, const double* D12, const double* ibc,
                                          const double* obc, const double* kinv,
                                          ipde_annular_scalar** out) {
    if (!ctx || !out) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, M >= 4 && M <= 256 && n >= 4);
    IPDE_CHECK_ARG(ctx, R01 && R12 && R02 && D01 && D12 && ibc && obc && kinv);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ipde_annular_scalar* h = new ipde_annular_scalar();
    h->ctx = ctx;
    h->M = M;
    h->n = n;
    h->NB = (int64_t)M * n;
    h->k2 = helmholtz_k * helmholtz_k;
    const int m1 = M - 1, m2 = M - 2;
    int st = IPDE_OK;
    auto up = [&](double** d, const double* s, size_t cnt) {
        if (st == IPDE_OK) st = upload(ctx, d, s, cnt);
    };
    up(&h->R01, R01, (size_t)m1 * M);
    up(&h->R12, R12, (size_t)m2 * m1);
    up(&h->R02, R02, (size_t)m2 * M);
    up(&h->D01, D01, (size_t)m1 * M);
    up(&h->D12, D12, (size_t)m2 * m1);
    std::vector<double> B((size_t)M * M);
    for (int a = 0; a < m2; ++a)
        for (int b = 0; b < M; ++b) B[(size_t)a * M + b] = h->k2 * R02[(size_t)a * M + b];
    for (int b = 0; b < M; ++b) {
        B[(size_t)m2 * M + b] = ibc[b];
        B[(size_t)m1 * M + b] = obc[b];
    }
    up(&h->Bmat, B.data(), B.size());
    {   // mode-minor transpose of the inverse blocks: Kt[j][k][i] = kinv[i][j][k]
        std::vector<double> Kt((size_t)M * M * n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < M; ++j)
                for (int k = 0; k < M; ++k)
                    Kt[((size_t)j * M + k) * n + i] = kinv[((size_t)i * M + j) * M + k];
        up(&h->Kt, Kt.data(), Kt.size());
    }
    if (fft_pair_supported(n)) {   // twiddles of the fused transform pairs (n = 8192: of its two 4096-point halves)
        const int nt = n == 8192 ? 4096 : n;
        std::vector<double> w(2 * (size_t)nt);
        for (int m = 0; m < nt; ++m) {
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)nt;
            w[2 * m] = (double)cosl(a);
            w[2 * m + 1] = (double)sinl(a);
        }
        double* d = nullptr;
        up(&d, w.data(), w.size());
        h->tw = d;
    }
    {   // iks = 1j * fftfreq(n, 1/n)  (annular_full.py:67-71)
        std::vector<double> v(2 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            int s = (i < (n + 1) / 2) ? i : i - n;
            v[2 * i] = 0.0;
            v[2 * i + 1] = (double)s;
        }
        up((double**)&h->iks, v.data(), v.size());
    }
    auto al = [&](void** d, size_t bytes) {
        if (st == IPDE_OK && hipMalloc(d, bytes) != hipSuccess) st = IPDE_ERR_ALLOC;
    };
    al((void**)&h->psi1, (size_t)m1 * n * sizeof(double));
    al((void**)&h->ipsi1, (size_t)m1 * n * sizeof(double));
    al((void**)&h->ipsi2, (size_t)m2 * n * sizeof(double));
    al((void**)&h->T, (size_t)2 * m1 * n * sizeof(cd));
    al((void**)&h->U, (size_t)2 * m1 * n * sizeof(cd));
    al((void**)&h->bvec, (size_t)M * n * sizeof(cd));
    al((void**)&h->rwork, (size_t)(M + 2) * n * sizeof(double));
    al((void**)&h->hin, (size_t)M * n * sizeof(cd));
    al((void**)&h->hout, (size_t)M * n * sizeof(cd));
    // the batched 1-D plans of THIS context (a private context has its own plan cache: warm
    // it here, at construction, instead of inside the first concurrent solves)
    for (int64_t b : {(int64_t)2 * m1, (int64_t)m2, (int64_t)M})
        if (st == IPDE_OK) st = ipde_fft1_prepare(ctx, b, n);
    if (st != IPDE_OK) {
        delete h;
        return st;
    }
    *out = h;
    return IPDE_OK;
}

extern "C" int ipde_annular_scalar_destroy(ipde_annular_scalar* h) {
    if (!h) return IPDE_ERR_INVALID;
    hipSetDevice(h->ctx->device);
    hipStreamSynchronize(h->ctx->stream);
    delete h;
    return IPDE_OK;
}

extern "C" int ipde_annular_scalar_set_geometry(ipde_annular_scalar* h, int loc, const double* psi1,
                                                const double* ipsi1, const double* ipsi2) {
    if (!h) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = h->ctx;
    IPDE_CHECK_ARG(ctx, psi1 && ipsi1 && ipsi2);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    IPDE_TRY(set_field(ctx, loc, h->psi1, psi1, (size_t)(h->M - 1) * h->n));
    IPDE_TRY(set_field(ctx, loc, h->ipsi1, ipsi1, (size_t)(h->M - 1) * h->n));
    IPDE_TRY(set_field(ctx, loc, h->ipsi2, ipsi2, (size_t)(h->M - 2) * h->n));
    h->have_geom = true;
    return IPDE_OK;
}

namespace {
template <class H, class F>
int run_vec_op(H* h, int loc, const double* in_c, double* out_c, int64_t NB, F f) {
    ipde_ctx* ctx = h->ctx;
    IPDE_CHECK_ARG(ctx, in_c && out_c);
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const cd* d_in = (const cd*)in_c;
    cd* d_out = (cd*)out_c;
    if (loc == IPDE_HOST) {
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(h->hin, in_c, NB * sizeof(cd), hipMemcpyHostToDevice,
                                           ctx->stream));
        d_in = h->hin;
        d_out = h->hout;
    }
    IPDE_TRY(f(d_in, d_out));
    if (loc == IPDE_HOST) {
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(out_c, h->hout, NB * sizeof(cd), hipMemcpyDeviceToHost,
                                           ctx->stream));
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return IPDE_OK;
}
}  // namespace

extern "C" int ipde_annular_scalar_apply(ipde_annular_scalar* h, int loc, const double* uh_c,
                                         double* out_c) {
    if (!h) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(h->ctx, h->have_geom);
    return run_vec_op(h, loc, uh_c, out_c, h->NB,
                      [&](const cd* a, cd* b) { return h->apply(a, b); });
}

extern "C" int ipde_annular_scalar_precondition(ipde_annular_scalar* h, int loc,
                                                const double* fh_c, double* out_c) {
    if (!h) return IPDE_ERR_INVALID;
    return run_vec_op(h, loc, fh_c, out_c, h->NB,
                      [&](const cd* a, cd* b) { return h->precond(a, b); });
}

extern "C" int ipde_annular_scalar_solve(ipde_annular_scalar* h, int loc, const double* f,
                                         const double* ig, const double* og, int negate_f,
                                         double tol, int maxiter, int restart, double* out,
                                         int* iters, double* resid) {
    if (!h) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = h->ctx;
    IPDE_CHECK_ARG(ctx, f && ig && og && out && iters && resid);
    IPDE_CHECK_ARG(ctx, h->have_geom);
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int M = h->M, n = h->n, m2 = M - 2;
    // real staging: rwork = [f (M,n) | ig | og]
    double* rw = h->rwork;
    auto kind = loc == IPDE_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rw, f, (size_t)M * n * sizeof(double), kind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rw + (size_t)M * n, ig, n * sizeof(double), kind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rw + (size_t)(M + 1) * n, og, n * sizeof(double), kind, st));
    // ff = [R02 f ; ig ; og] as real rows in U (used as real scratch), then complexify -> T
    double* ffr = (double*)h->U;
    hipLaunchKernelGGL(mixrr_kernel, dim3(nb256(n), m2), dim3(256), 0, st, ffr, n,
                       (const double*)h->R02, M, (const double*)rw, n, n, (const double*)nullptr, 0,
                       negate_f ? -1.0 : 1.0, 0.0);
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(ffr + (size_t)m2 * n, rw + (size_t)M * n,
                                       2 * (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(r2c_copy_kernel, dim3(nb256((int64_t)M * n)), dim3(256), 0, st, h->T,
                       (const double*)ffr, (int64_t)M * n, 1.0);
    IPDE_TRY(ipde_fft1_exec(ctx, M, n, -1, h->T, h->bvec));
    // the first cycle in one launch (annular_gmres_persist.h) when the configuration allows; a cycle
    // that runs out, and every other case, through the launch-per-stage cycles
    int st_g, conv = 0, it0 = 0;
    double r0 = 0.0;
    const int rs_eff = restart < 1 ? 1 : (restart > maxiter ? std::max(1, maxiter) : restart);
    if (h->persistent_cycle(h->bvec, tol, std::max(1, maxiter), rs_eff, &it0, &r0, &conv) == IPDE_OK) {
        if (conv || it0 >= maxiter) {
            *iters = it0;
            *resid = r0;
            st_g = conv ? IPDE_OK : IPDE_ERR_NOCONV;
        } else {
            st_g = gmres_solve(*h, h->gw, h->bvec, tol, maxiter, restart, iters, resid, it0);
        }
    } else {
        st_g = gmres_solve(*h, h->gw, h->bvec, tol, maxiter, restart, iters, resid);
    }
    if (st_g != IPDE_OK && st_g != IPDE_ERR_NOCONV) return st_g;
    // out = ifft(x).real
    IPDE_TRY(ipde_fft1_exec(ctx, M, n, +1, h->gw.x, h->T));
    double* d_out = loc == IPDE_HOST ? rw : out;
    hipLaunchKernelGGL(c2r_real_kernel, dim3(nb256((int64_t)M * n)), dim3(256), 0, st, d_out,
                       (const cd*)h->T, (int64_t)M * n, 1.0 / n);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    if (loc == IPDE_HOST) {
        IPDE_HIP_CHECK(ctx, hipMemcpyAsync(out, rw, (size_t)M * n * sizeof(double),
                                           hipMemcpyDeviceToHost, st));
        IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    if (st_g == IPDE_ERR_NOCONV)
        IPDE_SET_ERR(ctx, "annular scalar GMRES: no convergence in %d iterations (resid %.3e)",
                     *iters, *resid);
    return st_g;
}

// ===========================================================================
// Stokes annular solver (ns = n-1 modes)
struct ipde_annular_stokes : public LinOp {
    int M = 0, n = 0, ns = 0;
    double mu = 1.0;
    int64_t NU = 0, NP = 0;
    double *R01 = nullptr, *R12 = nullptr, *R02 = nullptr, *D01 = nullptr, *D12 = nullptr;
    double* BC = nullptr;    // (2, M): [ibcd; obcd]
    double* VI1 = nullptr;   // (M-1)
    cd* Kt = nullptr;        // [B][B][ns] complex
    cd* iks = nullptr;       // (ns)
    double *psi0 = nullptr, *psi1 = nullptr, *ipsi1 = nullptr, *ipsi2 = nullptr;
    double *combo1 = nullptr, *combo2 = nullptr, *c3 = nullptr, *c4 = nullptr, *DRpsi2 = nullptr;
    cd *A = nullptr, *Bw = nullptr;   // complex work, (6M) x n each
    cd* Cw = nullptr;                 // complex work, (3M) x n: the last transform's input (merged launches)
    double* Rw = nullptr;             // real work (4M) x n
    cd *xs = nullptr, *ys = nullptr;  // preconditioner stacking (B, ns)
    cd* bvec = nullptr;
    cd *hin = nullptr, *hout = nullptr;
    double* rstage = nullptr;
    GmresWork gw;
    bool have_geom = false;

    int apply(const cd* uuh, cd* out) override;
    int apply_grouped(const cd* uuh, cd* out);
    int precond(const cd* in, cd* out) override {
        // unknown ordering [ur (M,ns); ut (M,ns); p (M-1,ns)] is already the (B, ns)
        // row-major stacking the block matvec wants (stokes.py:200-210 transposes
        // only because its blocks are stored mode-major)
        const int B = 3 * M - 1;
        hipLaunchKernelGGL(prec_cplx_kernel, dim3(nb256(ns), B), dim3(256), 0, ctx->stream, out,
                           (const cd*)Kt, in, B, ns);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        return IPDE_OK;
    }
    int precond_scaled(const cd* w, const cd* nrm2, cd* vout, cd* out) override {
        const int B = 3 * M - 1;
        hipLaunchKernelGGL(prec_cplx_scaled_kernel, dim3(nb256(ns), B), dim3(256), 0, ctx->stream, out,
                           (const cd*)Kt, w, B, ns, nrm2, vout);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        return IPDE_OK;
    }
    ~ipde_annular_stokes() override {
        for (double** p : {&R01, &R12, &R02, &D01, &D12, &BC, &VI1, &psi0, &psi1, &ipsi1, &ipsi2,
                           &combo1, &combo2, &c3, &c4, &DRpsi2, &Rw, &rstage})
            if (*p) hipFree(*p);
        for (cd** p : {&Kt, &iks, &A, &Bw, &Cw, &xs, &ys, &bvec, &hin, &hout})
            if (*p) hipFree(*p);
        gmres_free(gw);
    }
};

namespace {
// out[j] += sum_b w[b] * in[b, 0]  for column 0 only: the pressure-mean fix
// fph[:,0] += (VI1 . ph)[0,0]   (stokes.py:383)
__global__ void pressure_mean_kernel(cd* __restrict__ fph, int ld, int rows,
                                     const double* __restrict__ w, const cd* __restrict__ ph,
                                     int ldp) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sr = 0.0, si = 0.0;
    for (int b = 0; b < rows; ++b) {
        cd v = ph[(size_t)b * ldp];
        sr = fma(w[b], v.x, sr);
        si = fma(w[b], v.y, si);
    }
    for (int a = 0; a < rows; ++a) {
        cd v = fph[(size_t)a * ld];
        fph[(size_t)a * ld] = cd{v.x + sr, v.y + si};
    }
}
// y[a,j] = alpha * F[a,j] * x[a,j] + beta * y[a,j]   (real)
__global__ __launch_bounds__(256) void rfield_axpby_kernel(double* __restrict__ y,
                                                           const double* __restrict__ x,
                                                           const double* __restrict__ F, int64_t n,
                                                           double alpha, double beta) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    double v = alpha * x[idx];
    if (F) v *= F[idx];
    if (beta != 0.0) v = fma(beta, y[idx], v);
    y[idx] = v;
}

// ---- grouped launches of the Stokes operator -------------------------------------------
// The operator is ~45 small dependent launches on one stream; a launch costs ~4 us of
// dependency latency whatever it computes (the kernels themselves average 6 us at n = 3200),
// so mutually independent launches are grouped into one and accumulation chains
// (out = beta out + alpha P (A (Q in)), five in a row into the same rows) become term lists
// evaluated by one thread per output entry — IN THE ORDER of the chain, with the chain's
// roundings: the results are bitwise those of the ungrouped sequence.
struct MixTerm {
    const double* A;     // (rows, ca) row-major; nullptr: identity (in must have `rows` rows)
    const void* in;      // complex rows (real part used) or real rows, leading dimension n
    const double* Q;     // nullable (ca, n) field on the input rows
    const double* P;     // nullable (rows, n) field on the output rows
    double alpha;
    int ca;
    int cplx;
};
constexpr int MIX_MAXT = 6, MIX_MAXO = 3, MIX_MAXC = 24;      // (MAXC: input rows of a term loaded at once; M <= 24)
struct MixOut {
    double* out;         // (rows, n); nullable when cout is given
    cd* cout;            // nullable: the same rows as complex numbers (imaginary part 0), the next
                         // transform's input (what r2c_copy_kernel made of `out` in a launch of its own)
    int rows;
    int nterms;
    MixTerm t[MIX_MAXT];
};
struct MixBatch {
    MixOut o[MIX_MAXO];
};
// grid = (ceil(n/256), max rows, outputs)
__global__ __launch_bounds__(256) void mix_multi_kernel(MixBatch B, int n) {
    const MixOut& O = B.o[blockIdx.z];
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int a = blockIdx.y;
    if (j >= n || a >= O.rows) return;
    double acc = 0.0;
    for (int k = 0; k < O.nterms; ++k) {
        const MixTerm& T = O.t[k];
        double s = 0.0;
        if (T.A) {
            const double* Ar = T.A + (size_t)a * T.ca;
            // all of the term's input rows are requested before the first is used (MIX_MAXC loads in flight: the
            // kernel is a chain of exposed round trips, not bandwidth; with four in flight it took 10 us at
            // n = 3200, M = 14); the sum keeps its order, b ascending
            double v[MIX_MAXC], q[MIX_MAXC];
            const bool hasq = T.Q != nullptr;
            if (T.ca <= MIX_MAXC) {
                if (T.cplx) {
                    const cd* in = (const cd*)T.in;
#pragma unroll
                    for (int b = 0; b < MIX_MAXC; ++b) v[b] = b < T.ca ? in[(size_t)b * n + j].x : 0.0;
                } else {
                    const double* in = (const double*)T.in;
#pragma unroll
                    for (int b = 0; b < MIX_MAXC; ++b) v[b] = b < T.ca ? in[(size_t)b * n + j] : 0.0;
                }
                if (hasq) {
#pragma unroll
                    for (int b = 0; b < MIX_MAXC; ++b) q[b] = b < T.ca ? T.Q[(size_t)b * n + j] : 0.0;
                }
#pragma unroll
                for (int b = 0; b < MIX_MAXC; ++b) {
                    if (b < T.ca) {
                        double x = v[b];
                        if (hasq) x *= q[b];
                        s = fma(Ar[b], x, s);
                    }
                }
            } else if (T.cplx) {
                const cd* in = (const cd*)T.in;
#pragma unroll 4
                for (int b = 0; b < T.ca; ++b) {
                    double x = in[(size_t)b * n + j].x;
                    if (T.Q) x *= T.Q[(size_t)b * n + j];
                    s = fma(Ar[b], x, s);
                }
            } else {
                const double* in = (const double*)T.in;
#pragma unroll 4
                for (int b = 0; b < T.ca; ++b) s = fma(Ar[b], in[(size_t)b * n + j], s);
            }
            if (T.P) s *= T.P[(size_t)a * n + j];
            s = T.alpha * s;
        } else {
            // identity term with the rounding of rfield_axpby_kernel: (alpha x) F
            s = T.alpha * ((const double*)T.in)[(size_t)a * n + j];
            if (T.P) s *= T.P[(size_t)a * n + j];
        }
        acc = (k == 0) ? s : fma(1.0, acc, s);
    }
    if (O.out) O.out[(size_t)a * n + j] = acc;
    if (O.cout) O.cout[(size_t)a * n + j] = cd{1.0 * acc, 0.0};
}

// the six splats of the operator's first stage: A rows [ur | ut | p | ik ur | ik ut | ik p]
__global__ __launch_bounds__(256) void splat6_kernel(cd* __restrict__ out, const cd* __restrict__ urh,
                                                     const cd* __restrict__ uth, const cd* __restrict__ ph,
                                                     const cd* __restrict__ iks, int M, int n) {
    const int m1 = M - 1, rows = 2 * M + m1;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)2 * rows * n) return;
    int a = (int)(idx / n), j = (int)(idx - (int64_t)a * n);
    const bool deriv = a >= rows;
    int r = deriv ? a - rows : a;
    const cd* src = r < M ? urh : (r < 2 * M ? uth : ph);
    r = r < M ? r : (r < 2 * M ? r - M : r - 2 * M);
    const int N2 = n / 2, ns = n - 1;
    cd v{0.0, 0.0};
    if (j != N2) {
        int js = j < N2 ? j : j - 1;
        v = src[(size_t)r * ns + js];
        if (deriv) v = cmul(v, iks[js]);
    }
    out[idx] = v;
}
// desplat followed by splat, in one pass on the (rows, n) layout: zero the Nyquist column,
// multiply the others by colfac
__global__ __launch_bounds__(256) void nyquist_mul_kernel(cd* __restrict__ out, const cd* __restrict__ in,
                                                          int rows, int n, const cd* __restrict__ colfac) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)rows * n) return;
    int j = (int)(idx % n);
    const int N2 = n / 2;
    cd v{0.0, 0.0};
    if (j != N2) {
        int js = j < N2 ? j : j - 1;
        cd w = cmul(in[idx], colfac[js]);
        v = cd{1.0 * w.x, 1.0 * w.y};
    }
    out[idx] = v;
}
// the three desplats of the last stage: FH rows [Fr (m2) | Ft (m2) | Fp (m1)] -> the first
// m2 / m2 / m1 rows of the three (M, ns) blocks of `out` (block stride NU)
__global__ __launch_bounds__(256) void desplat3_kernel(cd* __restrict__ out, int64_t NU,
                                                       const cd* __restrict__ in, int M, int n) {
    const int m1 = M - 1, m2 = M - 2, ns = n - 1, N2 = n / 2;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)(2 * m2 + m1) * ns) return;
    int a = (int)(idx / ns), js = (int)(idx - (int64_t)a * ns);
    int j = js < N2 ? js : js + 1;
    cd v = in[(size_t)a * n + j];
    int blk = a < m2 ? 0 : (a < 2 * m2 ? 1 : 2);
    int r = a - (blk == 0 ? 0 : (blk == 1 ? m2 : 2 * m2));
    out[(size_t)blk * NU + (size_t)r * ns + js] = cd{1.0 * v.x, 1.0 * v.y};
}
// the two boundary-condition mixc launches (blockIdx.z = 0: ur, 1: ut)
__global__ __launch_bounds__(256) void bc2_kernel(cd* __restrict__ out0, cd* __restrict__ out1, int ldo,
                                                  const double* __restrict__ A, int ca,
                                                  const cd* __restrict__ in0, const cd* __restrict__ in1,
                                                  int ldi, int n) {
    int j = blockIdx.x * 256 + threadIdx.x;
    int a = blockIdx.y;
    if (j >= n) return;
    cd* out = blockIdx.z == 0 ? out0 : out1;
    const cd* in = blockIdx.z == 0 ? in0 : in1;
    const double* Ar = A + (size_t)a * ca;
    double sr = 0.0, si = 0.0;
#pragma unroll 4
    for (int b = 0; b < ca; ++b) {
        cd v = in[(size_t)b * ldi + j];
        sr = fma(Ar[b], v.x, sr);
        si = fma(Ar[b], v.y, si);
    }
    out[(size_t)a * ldo + j] = cd{1.0 * sr, 1.0 * si};
}
// desplat3 + bc2 + pressure_mean in one launch (the three write disjoint entries of `out`, except
// that the pressure-mean fix adds to column 0 of the p block: the thread that desplats such an
// entry adds the sum itself).  Same operations in the same order as the three kernels: bitwise
// their result.  idx < nd: desplat entries; then 4 ns boundary-condition entries.
__global__ __launch_bounds__(256) void stokes_finish_kernel(cd* __restrict__ out, int64_t NU,
                                                            const cd* __restrict__ FH, int M, int n,
                                                            const double* __restrict__ BC,
                                                            const cd* __restrict__ urh, const cd* __restrict__ uth,
                                                            const double* __restrict__ VI1,
                                                            const cd* __restrict__ ph) {
    const int m1 = M - 1, m2 = M - 2, ns = n - 1, N2 = n / 2;
    const int64_t nd = (int64_t)(2 * m2 + m1) * ns;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < nd) {
        int a = (int)(idx / ns), js = (int)(idx - (int64_t)a * ns);
        int j = js < N2 ? js : js + 1;
        cd v = FH[(size_t)a * n + j];
        int blk = a < m2 ? 0 : (a < 2 * m2 ? 1 : 2);
        int r = a - (blk == 0 ? 0 : (blk == 1 ? m2 : 2 * m2));
        v = cd{1.0 * v.x, 1.0 * v.y};
        if (blk == 2 && js == 0) {
            double sr = 0.0, si = 0.0;
            for (int b = 0; b < m1; ++b) {
                cd w = ph[(size_t)b * ns];
                sr = fma(VI1[b], w.x, sr);
                si = fma(VI1[b], w.y, si);
            }
            v = cd{v.x + sr, v.y + si};
        }
        out[(size_t)blk * NU + (size_t)r * ns + js] = v;
        return;
    }
    idx -= nd;
    if (idx >= (int64_t)4 * ns) return;
    const int z = (int)(idx / (2 * ns));                 // 0: ur, 1: ut
    const int a = (int)((idx - (int64_t)z * 2 * ns) / ns);
    const int j = (int)(idx - (int64_t)z * 2 * ns - (int64_t)a * ns);
    const cd* in = z == 0 ? urh : uth;
    const double* Ar = BC + (size_t)a * M;
    double sr = 0.0, si = 0.0;
#pragma unroll 4
    for (int b = 0; b < M; ++b) {
        cd v = in[(size_t)b * ns + j];
        sr = fma(Ar[b], v.x, sr);
        si = fma(Ar[b], v.y, si);
    }
    out[(size_t)z * NU + (size_t)(m2 + a) * ns + j] = cd{1.0 * sr, 1.0 * si};
}
}  // namespace

// _apply_optim_real (ipde/annular/stokes.py:321-385)
int ipde_annular_stokes::apply(const cd* uuh, cd* out) {
    hipStream_t st = ctx->stream;
    const int m1 = M - 1, m2 = M - 2;
    const dim3 b(256);
    const cd* urh = uuh;
    const cd* uth = uuh + NU;
    const cd* ph = uuh + 2 * NU;
    cd* frh = out;
    cd* fth = out + NU;
    cd* fph = out + 2 * NU;
    const size_t Mn = (size_t)M * n;
    if (ctx->opt_annular_grouped) return apply_grouped(uuh, out);
    // --- inverse transforms: A rows [ur(M) | ut(M) | p(M-1) | dur(M) | dut(M) | dp(M-1)]
    cd* a_ur = A;
    cd* a_ut = A + Mn;
    cd* a_p = A + 2 * Mn;
    cd* a_dur = a_p + (size_t)m1 * n;
    cd* a_dut = a_dur + Mn;
    cd* a_dp = a_dut + Mn;
    hipLaunchKernelGGL(splat_kernel, dim3(nb256(Mn)), b, 0, st, a_ur, urh, M, n, (const cd*)nullptr);
    hipLaunchKernelGGL(splat_kernel, dim3(nb256(Mn)), b, 0, st, a_ut, uth, M, n, (const cd*)nullptr);
    hipLaunchKernelGGL(splat_kernel, dim3(nb256((int64_t)m1 * n)), b, 0, st, a_p, ph, m1, n,
                       (const cd*)nullptr);
    hipLaunchKernelGGL(splat_kernel, dim3(nb256(Mn)), b, 0, st, a_dur, urh, M, n, (const cd*)iks);
    hipLaunchKernelGGL(splat_kernel, dim3(nb256(Mn)), b, 0, st, a_dut, uth, M, n, (const cd*)iks);
    hipLaunchKernelGGL(splat_kernel, dim3(nb256((int64_t)m1 * n)), b, 0, st, a_dp, ph, m1, n,
                       (const cd*)iks);
    const int rowsA = 4 * M + 2 * m1;
    IPDE_TRY(ipde_fft1_exec(ctx, rowsA, n, +1, A, Bw));
    const double in = 1.0 / n;  // ifft scaling, folded into the first real-space use
    cd* ur = Bw;
    cd* ut = Bw + Mn;
    cd* p = Bw + 2 * Mn;
    cd* dur = p + (size_t)m1 * n;
    cd* dut = dur + Mn;
    cd* dp = dut + Mn;
    // --- tangential second derivatives: X = [ (R01 dur)*ipsi1 ; (R01 dut)*ipsi1 ] (real rows)
    double* X = Rw;                         // 2*m1 rows
    double* W2 = Rw + (size_t)2 * m1 * n;   // m1 rows: R01 dut (unscaled by ipsi1)
    double* Fr = W2 + (size_t)m1 * n;       // m2 rows: physical-space ur equation
    double* Ft = Fr + (size_t)m2 * n;       // m2 rows
    double* Fp = Ft + (size_t)m2 * n;       // m1 rows
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m1), b, 0, st, X, n, (const double*)R01, M,
                       (const cd*)dur, n, n, (const double*)nullptr, 0, (const double*)ipsi1, n, in, 0.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m1), b, 0, st, X + (size_t)m1 * n, n,
                       (const double*)R01, M, (const cd*)dut, n, n, (const double*)nullptr, 0,
                       (const double*)ipsi1, n, in, 0.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m1), b, 0, st, W2, n, (const double*)R01, M,
                       (const cd*)dut, n, n, (const double*)nullptr, 0, (const double*)nullptr, 0, in,
                       0.0);
    // forward, * iks (Nyquist dropped), inverse
    cd* C1 = A;                              // reuse A: 2*m1 rows complex
    cd* C2 = A + (size_t)2 * m1 * n;
    hipLaunchKernelGGL(r2c_copy_kernel, dim3(nb256((int64_t)2 * m1 * n)), b, 0, st, C1,
                       (const double*)X, (int64_t)2 * m1 * n, 1.0);
    IPDE_TRY(ipde_fft1_exec(ctx, 2 * m1, n, -1, C1, C2));
    // mfft -> *iks -> mifft == zero the Nyquist column and multiply by i k
    cd* C3 = C2 + (size_t)2 * m1 * n;        // (2 m1, ns)
    hipLaunchKernelGGL(desplat_kernel, dim3(nb256((int64_t)2 * m1 * ns)), b, 0, st, C3,
                       (const cd*)C2, 2 * m1, n, (const cd*)iks, 1.0);
    hipLaunchKernelGGL(splat_kernel, dim3(nb256((int64_t)2 * m1 * n)), b, 0, st, C1, (const cd*)C3,
                       2 * m1, n, (const cd*)nullptr);
    IPDE_TRY(ipde_fft1_exec(ctx, 2 * m1, n, +1, C1, C2));
    cd* urt2 = C2;                           // real part = d/dt (ur_t * ipsi1), unscaled by 1/n
    cd* utt2 = C2 + (size_t)m1 * n;
    // --- ur equation: Fr = mu*(-lap_ur + t1 + t2 + t3) + t4
    //   lap_ur = (D12 (D01 ur * psi1) + R12 urt2) * ipsi2
    double* G = Fp + (size_t)m1 * n;         // m1 rows scratch: D01 ur * psi1
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m1), b, 0, st, G, n, (const double*)D01, M,
                       (const cd*)ur, n, n, (const double*)nullptr, 0, (const double*)psi1, n, in, 0.0);
    hipLaunchKernelGGL(mixrr_kernel, dim3(nb256(n), m2), b, 0, st, Fr, n, (const double*)D12, m1,
                       (const double*)G, n, n, (const double*)ipsi2, n, -mu, 0.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Fr, n, (const double*)R12, m1,
                       (const cd*)urt2, n, n, (const double*)nullptr, 0, (const double*)ipsi2, n,
                       -mu * in, 1.0);
    //   t1 = R02 dut * combo1 ; t2 = R02 ur * combo2 ; t3 = R02 ut * c3 ; t4 = D12 p
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Fr, n, (const double*)R02, M,
                       (const cd*)dut, n, n, (const double*)nullptr, 0, (const double*)combo1, n,
                       mu * in, 1.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Fr, n, (const double*)R02, M,
                       (const cd*)ur, n, n, (const double*)nullptr, 0, (const double*)combo2, n,
                       mu * in, 1.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Fr, n, (const double*)R02, M,
                       (const cd*)ut, n, n, (const double*)nullptr, 0, (const double*)c3, n, mu * in,
                       1.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Fr, n, (const double*)D12, m1,
                       (const cd*)p, n, n, (const double*)nullptr, 0, (const double*)nullptr, 0, in,
                       1.0);
    // --- ut equation: Ft = mu*(-lap_ut - t1 + t2 - t3) + t4
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m1), b, 0, st, G, n, (const double*)D01, M,
                       (const cd*)ut, n, n, (const double*)nullptr, 0, (const double*)psi1, n, in, 0.0);
    hipLaunchKernelGGL(mixrr_kernel, dim3(nb256(n), m2), b, 0, st, Ft, n, (const double*)D12, m1,
                       (const double*)G, n, n, (const double*)ipsi2, n, -mu, 0.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Ft, n, (const double*)R12, m1,
                       (const cd*)utt2, n, n, (const double*)nullptr, 0, (const double*)ipsi2, n,
                       -mu * in, 1.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Ft, n, (const double*)R02, M,
                       (const cd*)dur, n, n, (const double*)nullptr, 0, (const double*)combo1, n,
                       -mu * in, 1.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Ft, n, (const double*)R02, M,
                       (const cd*)ut, n, n, (const double*)nullptr, 0, (const double*)combo2, n,
                       mu * in, 1.0);
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Ft, n, (const double*)R02, M,
                       (const cd*)ur, n, n, (const double*)nullptr, 0, (const double*)c4, n, -mu * in,
                       1.0);
    //   t4 = R12 (mifftr(ph*iks)) * ipsi2
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m2), b, 0, st, Ft, n, (const double*)R12, m1,
                       (const cd*)dp, n, n, (const double*)nullptr, 0, (const double*)ipsi2, n, in,
                       1.0);
    // --- div equation: Fp = (D01 (ur*psi0) + W2) * ipsi1
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), m1), b, 0, st, Fp, n, (const double*)D01, M,
                       (const cd*)ur, n, n, (const double*)psi0, n, (const double*)ipsi1, n, in, 0.0);
    hipLaunchKernelGGL(rfield_axpby_kernel, dim3(nb256((int64_t)m1 * n)), b, 0, st, Fp,
                       (const double*)W2, (const double*)ipsi1, (int64_t)m1 * n, 1.0, 1.0);
    // --- forward transforms of [Fr; Ft; Fp], drop Nyquist, scatter into out
    const int rowsF = 2 * m2 + m1;
    hipLaunchKernelGGL(r2c_copy_kernel, dim3(nb256((int64_t)rowsF * n)), b, 0, st, A,
                       (const double*)Fr, (int64_t)rowsF * n, 1.0);
    cd* FH = A + (size_t)rowsF * n;
    IPDE_TRY(ipde_fft1_exec(ctx, rowsF, n, -1, A, FH));
    hipLaunchKernelGGL(desplat_kernel, dim3(nb256((int64_t)m2 * ns)), b, 0, st, frh, (const cd*)FH,
                       m2, n, (const cd*)nullptr, 1.0);
    hipLaunchKernelGGL(desplat_kernel, dim3(nb256((int64_t)m2 * ns)), b, 0, st, fth,
                       (const cd*)(FH + (size_t)m2 * n), m2, n, (const cd*)nullptr, 1.0);
    hipLaunchKernelGGL(desplat_kernel, dim3(nb256((int64_t)m1 * ns)), b, 0, st, fph,
                       (const cd*)(FH + (size_t)2 * m2 * n), m1, n, (const cd*)nullptr, 1.0);
    // boundary rows (computed in Fourier space on the unknowns themselves)
    hipLaunchKernelGGL(mixc_kernel, dim3(nb256(ns), 2), b, 0, st, frh + (size_t)m2 * ns, ns,
                       (const double*)BC, M, urh, ns, ns, (const cd*)nullptr, 1.0, 0.0);
    hipLaunchKernelGGL(mixc_kernel, dim3(nb256(ns), 2), b, 0, st, fth + (size_t)m2 * ns, ns,
                       (const double*)BC, M, uth, ns, ns, (const cd*)nullptr, 1.0, 0.0);
    hipLaunchKernelGGL(pressure_mean_kernel, dim3(1), dim3(64), 0, st, fph, ns, m1,
                       (const double*)VI1, ph, ns);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// The same operator with grouped launches (mix_multi_kernel and friends above): 12 launches
// + 4 transforms instead of ~45 + 4, bitwise the same result.
int ipde_annular_stokes::apply_grouped(const cd* uuh, cd* out) {
    hipStream_t st = ctx->stream;
    const int m1 = M - 1, m2 = M - 2;
    const dim3 b(256);
    const cd* urh = uuh;
    const cd* uth = uuh + NU;
    const cd* ph = uuh + 2 * NU;
    const size_t Mn = (size_t)M * n;
    const int rowsA = 4 * M + 2 * m1;
    // annular_grouped = 2: the real-to-complex copies ride in the term-list launches and the three
    // closing launches are one (13 -> 9 launches per operator application, same bits)
    const bool merged = ctx->opt_annular_grouped >= 2;
    hipLaunchKernelGGL(splat6_kernel, dim3(nb256((int64_t)rowsA * n)), b, 0, st, A, urh, uth, ph, (const cd*)iks,
                       M, n);
    IPDE_TRY(ipde_fft1_exec(ctx, rowsA, n, +1, A, Bw));
    const double in = 1.0 / n;
    const cd* ur = Bw;
    const cd* ut = Bw + Mn;
    const cd* p = Bw + 2 * Mn;
    const cd* dur = p + (size_t)m1 * n;
    const cd* dut = dur + Mn;
    const cd* dp = dut + Mn;
    double* X = Rw;
    double* W2 = Rw + (size_t)2 * m1 * n;
    double* Fr = W2 + (size_t)m1 * n;
    double* Ft = Fr + (size_t)m2 * n;
    double* Fp = Ft + (size_t)m2 * n;
    double* Gr = Fp + (size_t)m1 * n;        // D01 ur * psi1
    double* Gt = Gr + (size_t)m1 * n;        // D01 ut * psi1
    auto term = [](const double* Amat, int ca, const void* inp, int cplx, const double* Q, const double* P,
                   double alpha) {
        MixTerm t;
        t.A = Amat;
        t.in = inp;
        t.Q = Q;
        t.P = P;
        t.alpha = alpha;
        t.ca = ca;
        t.cplx = cplx;
        return t;
    };
    {   // X = [(R01 dur) ipsi1 ; (R01 dut) ipsi1],  W2 = R01 dut
        MixBatch B{};
        B.o[0].out = X;
        B.o[1].out = X + (size_t)m1 * n;
        B.o[2].out = W2;
        for (int k = 0; k < 3; ++k) {
            B.o[k].rows = m1;
            B.o[k].nterms = 1;
        }
        B.o[0].t[0] = term(R01, M, dur, 1, nullptr, ipsi1, in);
        B.o[1].t[0] = term(R01, M, dut, 1, nullptr, ipsi1, in);
        B.o[2].t[0] = term(R01, M, dut, 1, nullptr, nullptr, in);
        if (merged) {      // X straight into the transform's input (A is free: its splats are transformed)
            B.o[0].out = nullptr;
            B.o[1].out = nullptr;
            B.o[0].cout = A;
            B.o[1].cout = A + (size_t)m1 * n;
        }
        hipLaunchKernelGGL(mix_multi_kernel, dim3(nb256(n), m1, 3), b, 0, st, B, n);
    }
    cd* C1 = A;
    cd* C2 = A + (size_t)2 * m1 * n;
    if (!merged)
        hipLaunchKernelGGL(r2c_copy_kernel, dim3(nb256((int64_t)2 * m1 * n)), b, 0, st, C1, (const double*)X,
                           (int64_t)2 * m1 * n, 1.0);
    IPDE_TRY(ipde_fft1_exec(ctx, 2 * m1, n, -1, C1, C2));
    hipLaunchKernelGGL(nyquist_mul_kernel, dim3(nb256((int64_t)2 * m1 * n)), b, 0, st, C1, (const cd*)C2, 2 * m1,
                       n, (const cd*)iks);
    IPDE_TRY(ipde_fft1_exec(ctx, 2 * m1, n, +1, C1, C2));
    const cd* urt2 = C2;
    const cd* utt2 = C2 + (size_t)m1 * n;
    {   // Gr, Gt and the divergence rows Fp = (D01 (ur psi0)) ipsi1 + W2 ipsi1
        MixBatch B{};
        B.o[0].out = Gr;
        B.o[0].rows = m1;
        B.o[0].nterms = 1;
        B.o[0].t[0] = term(D01, M, ur, 1, nullptr, psi1, in);
        B.o[1].out = Gt;
        B.o[1].rows = m1;
        B.o[1].nterms = 1;
        B.o[1].t[0] = term(D01, M, ut, 1, nullptr, psi1, in);
        B.o[2].out = Fp;
        B.o[2].rows = m1;
        B.o[2].nterms = 2;
        B.o[2].t[0] = term(D01, M, ur, 1, psi0, ipsi1, in);
        B.o[2].t[1] = term(nullptr, 0, W2, 0, nullptr, ipsi1, 1.0);
        if (merged) {      // rows [Fr | Ft | Fp] of the last transform's input: Fp from here
            B.o[2].out = nullptr;
            B.o[2].cout = Cw + (size_t)2 * m2 * n;
        }
        hipLaunchKernelGGL(mix_multi_kernel, dim3(nb256(n), m1, 3), b, 0, st, B, n);
    }
    {   // Fr = mu (-lap_ur + t1 + t2 + t3) + t4,  Ft = mu (-lap_ut - t1 + t2 - t3) + t4
        MixBatch B{};
        B.o[0].out = Fr;
        B.o[0].rows = m2;
        B.o[0].nterms = 6;
        B.o[0].t[0] = term(D12, m1, Gr, 0, nullptr, ipsi2, -mu);
        B.o[0].t[1] = term(R12, m1, urt2, 1, nullptr, ipsi2, -mu * in);
        B.o[0].t[2] = term(R02, M, dut, 1, nullptr, combo1, mu * in);
        B.o[0].t[3] = term(R02, M, ur, 1, nullptr, combo2, mu * in);
        B.o[0].t[4] = term(R02, M, ut, 1, nullptr, c3, mu * in);
        B.o[0].t[5] = term(D12, m1, p, 1, nullptr, nullptr, in);
        B.o[1].out = Ft;
        B.o[1].rows = m2;
        B.o[1].nterms = 6;
        B.o[1].t[0] = term(D12, m1, Gt, 0, nullptr, ipsi2, -mu);
        B.o[1].t[1] = term(R12, m1, utt2, 1, nullptr, ipsi2, -mu * in);
        B.o[1].t[2] = term(R02, M, dur, 1, nullptr, combo1, -mu * in);
        B.o[1].t[3] = term(R02, M, ut, 1, nullptr, combo2, mu * in);
        B.o[1].t[4] = term(R02, M, ur, 1, nullptr, c4, -mu * in);
        B.o[1].t[5] = term(R12, m1, dp, 1, nullptr, ipsi2, in);
        if (merged) {
            B.o[0].out = nullptr;
            B.o[1].out = nullptr;
            B.o[0].cout = Cw;
            B.o[1].cout = Cw + (size_t)m2 * n;
        }
        hipLaunchKernelGGL(mix_multi_kernel, dim3(nb256(n), m2, 2), b, 0, st, B, n);
    }
    const int rowsF = 2 * m2 + m1;
    if (merged) {
        cd* FH = A;        // (urt2 / utt2 in A + 2 m1 n are spent: the term lists above were their last readers)
        IPDE_TRY(ipde_fft1_exec(ctx, rowsF, n, -1, Cw, FH));
        hipLaunchKernelGGL(stokes_finish_kernel, dim3(nb256((int64_t)rowsF * ns + 4 * (int64_t)ns)), b, 0, st, out,
                           (int64_t)NU, (const cd*)FH, M, n, (const double*)BC, urh, uth, (const double*)VI1, ph);
        IPDE_HIP_CHECK(ctx, hipGetLastError());
        return IPDE_OK;
    }
    hipLaunchKernelGGL(r2c_copy_kernel, dim3(nb256((int64_t)rowsF * n)), b, 0, st, A, (const double*)Fr,
                       (int64_t)rowsF * n, 1.0);
    cd* FH = A + (size_t)rowsF * n;
    IPDE_TRY(ipde_fft1_exec(ctx, rowsF, n, -1, A, FH));
    hipLaunchKernelGGL(desplat3_kernel, dim3(nb256((int64_t)rowsF * ns)), b, 0, st, out, (int64_t)NU, (const cd*)FH,
                       M, n);
    hipLaunchKernelGGL(bc2_kernel, dim3(nb256(ns), 2, 2), b, 0, st, out + (size_t)m2 * ns,
                       out + NU + (size_t)m2 * ns, ns, (const double*)BC, M, urh, uth, ns, ns);
    hipLaunchKernelGGL(pressure_mean_kernel, dim3(1), dim3(64), 0, st, out + 2 * NU, ns, m1, (const double*)VI1, ph,
                       ns);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_annular_stokes_create(ipde_ctx* ctx, int M, int n, double mu, const double* R01,
                                          const double* R12, const double* R02, const double* D01,
                                          const double* D12, const double* ibcd, const double* obcd,
                                          const double* VI1row0, const double* kinv_c,
                                          ipde_annular_stokes** out) {
    if (!ctx || !out) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, M >= 4 && M <= 128 && n >= 4 && (n % 2) == 0);
    IPDE_CHECK_ARG(ctx, R01 && R12 && R02 && D01 && D12 && ibcd && obcd && VI1row0 && kinv_c);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ipde_annular_stokes* h = new ipde_annular_stokes();
    h->ctx = ctx;
    h->M = M;
    h->n = n;
    h->ns = n - 1;
    h->mu = mu;
    const int ns = n - 1, m1 = M - 1, m2 = M - 2, B = 3 * M - 1;
    h->NU = (int64_t)M * ns;
    h->NP = (int64_t)m1 * ns;
    h->NB = 2 * h->NU + h->NP;
    int st = IPDE_OK;
    auto up = [&](double** d, const double* s, size_t cnt) {
        if (st == IPDE_OK) st = upload(ctx, d, s, cnt);
    };
    up(&h->R01, R01, (size_t)m1 * M);
    up(&h->R12, R12, (size_t)m2 * m1);
    up(&h->R02, R02, (size_t)m2 * M);
    up(&h->D01, D01, (size_t)m1 * M);
    up(&h->D12, D12, (size_t)m2 * m1);
    std::vector<double> bc(2 * (size_t)M);
    for (int i = 0; i < M; ++i) {
        bc[i] = ibcd[i];
        bc[M + i] = obcd[i];
    }
    up(&h->BC, bc.data(), bc.size());
    up(&h->VI1, VI1row0, m1);
    {   // Kt[j][k][i] = kinv[i][j][k] (complex)
        std::vector<double> Kt((size_t)2 * B * B * ns);
        for (int i = 0; i < ns; ++i)
            for (int j = 0; j < B; ++j)
                for (int k = 0; k < B; ++k) {
                    size_t s = (((size_t)i * B + j) * B + k) * 2;
                    size_t d = (((size_t)j * B + k) * ns + i) * 2;
                    Kt[d] = kinv_c[s];
                    Kt[d + 1] = kinv_c[s + 1];
                }
        up((double**)&h->Kt, Kt.data(), Kt.size());
    }
    {   // iks = 1j*ks, ks = fftfreq(n,1/n) without the Nyquist entry (annular.py:67-71)
        std::vector<double> v(2 * (size_t)ns);
        const int N2 = n / 2;
        for (int js = 0; js < ns; ++js) {
            int j = js < N2 ? js : js + 1;
            int s = (j < (n + 1) / 2) ? j : j - n;
            v[2 * js] = 0.0;
            v[2 * js + 1] = (double)s;
        }
        up((double**)&h->iks, v.data(), v.size());
    }
    auto al = [&](void** d, size_t bytes) {
        if (st == IPDE_OK && hipMalloc(d, bytes) != hipSuccess) st = IPDE_ERR_ALLOC;
    };
    al((void**)&h->psi0, (size_t)M * n * 8);
    al((void**)&h->psi1, (size_t)m1 * n * 8);
    al((void**)&h->ipsi1, (size_t)m1 * n * 8);
    al((void**)&h->ipsi2, (size_t)m2 * n * 8);
    al((void**)&h->combo1, (size_t)m2 * n * 8);
    al((void**)&h->combo2, (size_t)m2 * n * 8);
    al((void**)&h->c3, (size_t)m2 * n * 8);
    al((void**)&h->c4, (size_t)m2 * n * 8);
    al((void**)&h->DRpsi2, (size_t)m2 * n * 8);
    al((void**)&h->A, (size_t)(6 * M + 8) * n * sizeof(cd));
    al((void**)&h->Bw, (size_t)(6 * M + 8) * n * sizeof(cd));
    al((void**)&h->Cw, (size_t)(3 * M + 8) * n * sizeof(cd));
    al((void**)&h->Rw, (size_t)(8 * M + 8) * n * 8);
    al((void**)&h->bvec, (size_t)h->NB * sizeof(cd));
    al((void**)&h->hin, (size_t)h->NB * sizeof(cd));
    al((void**)&h->hout, (size_t)h->NB * sizeof(cd));
    al((void**)&h->rstage, (size_t)(3 * M + 8) * n * 8);
    // batched 1-D plans of this context, see ipde_annular_scalar_create
    for (int64_t b : {(int64_t)4 * M + 2 * m1, (int64_t)2 * m1, (int64_t)2 * m2 + m1, (int64_t)2 * M,
                      (int64_t)2 * M + m1})
        if (st == IPDE_OK) st = ipde_fft1_prepare(ctx, b, n);
    if (st != IPDE_OK) {
        delete h;
        return st;
    }
    *out = h;
    return IPDE_OK;
}

extern "C" int ipde_annular_stokes_destroy(ipde_annular_stokes* h) {
    if (!h) return IPDE_ERR_INVALID;
    hipSetDevice(h->ctx->device);
    hipStreamSynchronize(h->ctx->stream);
    delete h;
    return IPDE_OK;
}

namespace {
// combo1 = 2 DR ipsi2^2 ; combo2 = DR^2 ipsi2^2   (stokes.py:521-522)
__global__ __launch_bounds__(256) void combos_kernel(double* __restrict__ c1, double* __restrict__ c2,
                                                     const double* __restrict__ DR,
                                                     const double* __restrict__ ipsi2, int64_t n) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    double d = DR[idx], q = ipsi2[idx];
    c1[idx] = 2.0 * d * q * q;
    c2[idx] = d * d * q * q;
}
}  // namespace

extern "C" int ipde_annular_stokes_set_geometry(ipde_annular_stokes* h, int loc, const double* psi0,
                                                const double* psi1, const double* ipsi1,
                                                const double* ipsi2, const double* DR_psi2,
                                                const double* c3, const double* c4) {
    if (!h) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = h->ctx;
    IPDE_CHECK_ARG(ctx, psi0 && psi1 && ipsi1 && ipsi2 && DR_psi2 && c3 && c4);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t n = h->n, M = h->M;
    IPDE_TRY(set_field(ctx, loc, h->psi0, psi0, M * n));
    IPDE_TRY(set_field(ctx, loc, h->psi1, psi1, (M - 1) * n));
    IPDE_TRY(set_field(ctx, loc, h->ipsi1, ipsi1, (M - 1) * n));
    IPDE_TRY(set_field(ctx, loc, h->ipsi2, ipsi2, (M - 2) * n));
    IPDE_TRY(set_field(ctx, loc, h->DRpsi2, DR_psi2, (M - 2) * n));
    IPDE_TRY(set_field(ctx, loc, h->c3, c3, (M - 2) * n));
    IPDE_TRY(set_field(ctx, loc, h->c4, c4, (M - 2) * n));
    hipLaunchKernelGGL(combos_kernel, dim3(nb256((M - 2) * n)), dim3(256), 0, ctx->stream, h->combo1,
                       h->combo2, (const double*)h->DRpsi2, (const double*)h->ipsi2,
                       (int64_t)((M - 2) * n));
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    h->have_geom = true;
    return IPDE_OK;
}

extern "C" int ipde_annular_stokes_apply(ipde_annular_stokes* h, int loc, const double* uuh_c,
                                         double* out_c) {
    if (!h) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(h->ctx, h->have_geom);
    return run_vec_op(h, loc, uuh_c, out_c, h->NB,
                      [&](const cd* a, cd* b) { return h->apply(a, b); });
}

extern "C" int ipde_annular_stokes_precondition(ipde_annular_stokes* h, int loc,
                                                const double* ffh_c, double* out_c) {
    if (!h) return IPDE_ERR_INVALID;
    return run_vec_op(h, loc, ffh_c, out_c, h->NB,
                      [&](const cd* a, cd* b) { return h->precond(a, b); });
}

extern "C" int ipde_annular_stokes_solve(ipde_annular_stokes* h, int loc, const double* fr,
                                         const double* ft, const double* irg, const double* itg,
                                         const double* org, const double* otg,
                                         const double* P10_host, double tol, int maxiter,
                                         int restart, double* ur, double* ut, double* p, int* iters,
                                         double* resid) {
    if (!h) return IPDE_ERR_INVALID;
    ipde_ctx* ctx = h->ctx;
    IPDE_CHECK_ARG(ctx, fr && ft && irg && itg && org && otg && P10_host && ur && ut && p);
    IPDE_CHECK_ARG(ctx, iters && resid && h->have_geom);
    IPDE_CHECK_ARG(ctx, loc == IPDE_HOST || loc == IPDE_DEVICE);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int M = h->M, n = h->n, ns = h->ns, m1 = M - 1, m2 = M - 2;
    const size_t Mn = (size_t)M * n;
    auto kind = loc == IPDE_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    // stage: rstage = [fr (M,n) | ft (M,n) | irg | org | itg | otg]
    double* rs = h->rstage;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rs, fr, Mn * 8, kind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rs + Mn, ft, Mn * 8, kind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rs + 2 * Mn, irg, (size_t)n * 8, kind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rs + 2 * Mn + n, org, (size_t)n * 8, kind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rs + 2 * Mn + 2 * n, itg, (size_t)n * 8, kind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(rs + 2 * Mn + 3 * n, otg, (size_t)n * 8, kind, st));
    // ffr = [R02 fr ; irg ; org], fft = [R02 ft ; itg ; otg]  as real rows in Rw (2M rows)
    double* ff = h->Rw;
    hipLaunchKernelGGL(mixrr_kernel, dim3(nb256(n), m2), dim3(256), 0, st, ff, n,
                       (const double*)h->R02, M, (const double*)rs, n, n, (const double*)nullptr, 0,
                       1.0, 0.0);
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(ff + (size_t)m2 * n, rs + 2 * Mn, 2 * (size_t)n * 8,
                                       hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(mixrr_kernel, dim3(nb256(n), m2), dim3(256), 0, st, ff + Mn, n,
                       (const double*)h->R02, M, (const double*)(rs + Mn), n, n,
                       (const double*)nullptr, 0, 1.0, 0.0);
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(ff + Mn + (size_t)m2 * n, rs + 2 * Mn + 2 * n,
                                       2 * (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(r2c_copy_kernel, dim3(nb256(2 * Mn)), dim3(256), 0, st, h->A,
                       (const double*)ff, (int64_t)(2 * Mn), 1.0);
    cd* FH = h->A + 2 * Mn;
    IPDE_TRY(ipde_fft1_exec(ctx, 2 * M, n, -1, h->A, FH));
    hipLaunchKernelGGL(desplat_kernel, dim3(nb256(2 * (int64_t)M * ns)), dim3(256), 0, st, h->bvec,
                       (const cd*)FH, 2 * M, n, (const cd*)nullptr, 1.0);
    IPDE_HIP_CHECK(ctx, hipMemsetAsync(h->bvec + 2 * h->NU, 0, h->NP * sizeof(cd), st));
    int st_g = gmres_solve(*h, h->gw, h->bvec, tol, maxiter, restart, iters, resid);
    if (st_g != IPDE_OK && st_g != IPDE_ERR_NOCONV) return st_g;
    // ur = mifft(urh).real, ut = ..., p = P10 . mifft(ph).real
    const int rows = 2 * M + m1;
    hipLaunchKernelGGL(splat_kernel, dim3(nb256((int64_t)rows * n)), dim3(256), 0, st, h->A,
                       (const cd*)h->gw.x, rows, n, (const cd*)nullptr);
    IPDE_TRY(ipde_fft1_exec(ctx, rows, n, +1, h->A, h->Bw));
    double* o = h->Rw;  // [ur | ut | p] (3M rows)
    hipLaunchKernelGGL(c2r_real_kernel, dim3(nb256(2 * Mn)), dim3(256), 0, st, o,
                       (const cd*)h->Bw, (int64_t)(2 * Mn), 1.0 / n);
    // P10 upload (M x m1) into the tail of rstage
    double* dP10 = rs;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(dP10, P10_host, (size_t)M * m1 * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mixr_kernel, dim3(nb256(n), M), dim3(256), 0, st, o + 2 * Mn, n,
                       (const double*)dP10, m1, (const cd*)(h->Bw + 2 * Mn), n, n,
                       (const double*)nullptr, 0, (const double*)nullptr, 0, 1.0 / n, 0.0);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    auto okind = loc == IPDE_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(ur, o, Mn * 8, okind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(ut, o + Mn, Mn * 8, okind, st));
    IPDE_HIP_CHECK(ctx, hipMemcpyAsync(p, o + 2 * Mn, Mn * 8, okind, st));
    IPDE_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (st_g == IPDE_ERR_NOCONV)
        IPDE_SET_ERR(ctx, "annular Stokes GMRES: no convergence in %d iterations (resid %.3e)",
                     *iters, *resid);
    return st_g;
}

// ===========================================================================
// Stokes helper algebra between the annular solve and the QFS solves (reference
// ipde/solvers/internals/vector.py:65-144: convert_rt_to_uv, get_interface_traction_uvp, the
// traction / velocity jumps), on device arrays, one library call: the tangential derivative by
// two batched 1-D transforms, everything else in two kernels.  In the multi-boundary solver this
// replaces ~55 small tensor operations per boundary issued from a Python thread.
namespace {

// geom: bnx, bny, btx, bty (boundary normal / tangent), inx, iny (interface normal): 6 rows of n
__global__ __launch_bounds__(256) void stokes_rotate_kernel(const double* __restrict__ a,
                                                            const double* __restrict__ b,
                                                            const double* __restrict__ geom, int M, int n,
                                                            int to_rt, double* __restrict__ o1,
                                                            double* __restrict__ o2) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * n) return;
    const int j = (int)(idx % n);
    const double nx = geom[j], ny = geom[n + j], tx = geom[2 * n + j], ty = geom[3 * n + j];
    const double x = a[idx], y = b[idx];
    if (to_rt) {                // (u, v) -> (r, t)   (embedded_boundary.py:242-244)
        o1[idx] = x * nx + y * ny;
        o2[idx] = x * tx + y * ty;
    } else {                    // (r, t) -> (u, v)   (:246-248)
        o1[idx] = x * nx + y * tx;
        o2[idx] = x * ny + y * ty;
    }
}

__global__ __launch_bounds__(256) void stokes_ik_kernel(cd* __restrict__ h, const double* __restrict__ rk,
                                                        int M, int n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * n) return;
    const double k = rk[idx % n];
    const cd v = h[idx];
    h[idx] = cd{-k * v.y, k * v.x};          // times i k
}

// a thread per interface node j: tractions of the annular solution on the interface, then the jumps
__global__ __launch_bounds__(256) void stokes_jump_kernel(const double* __restrict__ rr,
                                                          const double* __restrict__ tr,
                                                          const double* __restrict__ pr,
                                                          const cd* __restrict__ dth,     // ifft(i k fft(rr)), unscaled
                                                          const double* __restrict__ geom,
                                                          const double* __restrict__ rs,
                                                          const double* __restrict__ irs,
                                                          const double* __restrict__ D00,
                                                          const double* __restrict__ est,
                                                          const double* __restrict__ bdata, int M, int n,
                                                          double sign, double* __restrict__ taus,
                                                          double* __restrict__ taud) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const double invn = 1.0 / (double)n;
    double e_rr = 0.0, e_p = 0.0, e_tr = 0.0, e_rt = 0.0;
    for (int m = 0; m < M; ++m) {
        double s_r = 0.0, s_t = 0.0;
        const double* Dm = D00 + (size_t)m * M;
#pragma unroll 4
        for (int k = 0; k < M; ++k) {
            const size_t kj = (size_t)k * n + j;
            s_r = fma(Dm[k], rr[kj], s_r);                      // (D00 Ur)[m]
            s_t = fma(Dm[k], tr[kj] * irs[kj], s_t);            // (D00 (Ut / speed))[m]
        }
        const size_t mj = (size_t)m * n + j;
        const double e = est[m];
        e_rr = fma(e, s_r, e_rr);
        e_p = fma(e, pr[mj], e_p);
        e_tr = fma(e, rs[mj] * s_t, e_tr);                       // Utr = speed * D00 (Ut / speed)
        e_rt = fma(e, (dth[mj].x * invn) * irs[mj], e_rt);       // Urt = d_t Ur / speed
    }
    const double Tr = 2.0 * e_rr - e_p, Tt = e_tr + e_rt;
    const double bnx = geom[j], bny = geom[n + j], btx = geom[2 * n + j], bty = geom[3 * n + j];
    const double inx = geom[4 * n + j], iny = geom[5 * n + j];
    const double rtx = Tr * bnx + Tt * btx, rty = Tr * bny + Tt * bty;
    const double bu = bdata[j], bv = bdata[n + j], bxx = bdata[2 * n + j], bxy = bdata[3 * n + j],
                 byy = bdata[4 * n + j];
    const double gx = bxx * inx + bxy * iny, gy = bxy * inx + byy * iny;
    taus[j] = sign * (rtx - gx);
    taus[n + j] = sign * (rty - gy);
    taud[j] = sign * bu;
    taud[n + j] = sign * bv;
}

}  // namespace

extern "C" int ipde_stokes_rotate(ipde_ctx* ctx, int loc, int M, int n, const double* a, const double* b,
                                  const double* geom, int to_rt, double* o1, double* o2) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, (loc == IPDE_HOST || loc == IPDE_DEVICE) && M >= 1 && n >= 1 && a && b && geom && o1 && o2);
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double *da, *db;
    IPDE_TRY(ipde_stage_in(ctx, loc, 0, a, (size_t)M * n, &da));
    IPDE_TRY(ipde_stage_in(ctx, loc, 1, b, (size_t)M * n, &db));
    hipLaunchKernelGGL(stokes_rotate_kernel, dim3(nb256((int64_t)M * n)), dim3(256), 0, ctx->stream, da, db, geom,
                       M, n, to_rt, o1, o2);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

extern "C" int ipde_stokes_interface_jumps(ipde_ctx* ctx, int M, int n, const double* rr, const double* tr,
                                           const double* pr, const double* geom, const double* rs,
                                           const double* irs, const double* D00, const double* est,
                                           const double* rk, const double* bdata, double sign, double* ur,
                                           double* vr, double* taus, double* taud) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, M >= 1 && n >= 1 && rr && tr && pr && geom && rs && irs && D00 && est && rk && bdata);
    IPDE_CHECK_ARG(ctx, ur && vr && taus && taud && (sign == 1.0 || sign == -1.0));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)M * n * sizeof(cd);
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->r2g[0], bytes));
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->r2g[1], bytes));
    cd* A = (cd*)ctx->r2g[0].p;
    cd* B = (cd*)ctx->r2g[1].p;
    hipStream_t st = ctx->stream;
    const unsigned g = nb256((int64_t)M * n);
    // d/dt of the radial component along the annulus' rows: ifft(i k fft(.))
    hipLaunchKernelGGL(r2c_copy_kernel, dim3(g), dim3(256), 0, st, A, rr, (int64_t)M * n, 1.0);
    IPDE_TRY(ipde_fft1_exec(ctx, M, n, -1, A, B));
    hipLaunchKernelGGL(stokes_ik_kernel, dim3(g), dim3(256), 0, st, B, rk, M, n);
    IPDE_TRY(ipde_fft1_exec(ctx, M, n, +1, B, A));
    hipLaunchKernelGGL(stokes_jump_kernel, dim3(nb256(n)), dim3(256), 0, st, rr, tr, pr, (const cd*)A, geom, rs, irs,
                       D00, est, bdata, M, n, sign, taus, taud);
    hipLaunchKernelGGL(stokes_rotate_kernel, dim3(g), dim3(256), 0, st, rr, tr, geom, M, n, 0, ur, vr);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}

// The scalar counterpart (reference ipde/solvers/internals/scalar.py:76-90): interface normal
// derivative of the annular solution by the estimator row, jumps against the grid solution's value
// and gradient.  nrm: DEVICE, 2 x n (interface normal x, y); bdata: DEVICE, 3 x n (u, u_x, u_y).
namespace {
__global__ __launch_bounds__(256) void scalar_jump_kernel(const double* __restrict__ ur,
                                                          const double* __restrict__ est,
                                                          const double* __restrict__ nrm,
                                                          const double* __restrict__ bdata, int M, int n,
                                                          double sign, double* __restrict__ slp,
                                                          double* __restrict__ dlp) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double urn = 0.0;
#pragma unroll 4
    for (int m = 0; m < M; ++m) urn = fma(est[m], ur[(size_t)m * n + j], urn);
    const double ucn = bdata[n + j] * nrm[j] + bdata[2 * n + j] * nrm[n + j];
    slp[j] = sign * (urn - ucn);
    dlp[j] = sign * bdata[j];
}
}  // namespace

extern "C" int ipde_scalar_interface_jumps(ipde_ctx* ctx, int M, int n, const double* ur, const double* est,
                                           const double* nrm, const double* bdata, double sign, double* slp,
                                           double* dlp) {
    if (!ctx) return IPDE_ERR_INVALID;
    IPDE_CHECK_ARG(ctx, M >= 1 && n >= 1 && ur && est && nrm && bdata && slp && dlp && (sign == 1.0 || sign == -1.0));
    IPDE_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(scalar_jump_kernel, dim3(nb256(n)), dim3(256), 0, ctx->stream, ur, est, nrm, bdata, M, n, sign,
                       slp, dlp);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    return IPDE_OK;
}
