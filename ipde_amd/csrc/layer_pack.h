// Target bounding box + source packing kernels shared by the layer-potential
// families.  Everything stays on the device (no host round trip per apply).
#pragma once
#include "layer_common.h"

#define IPDE_BBOX_BLOCKS 256

// part[4*b + {0,1,2,3}] = per-block {xmin, xmax, ymin, ymax}
__global__ __launch_bounds__(256) static void ipde_bbox_kernel(const double* __restrict__ tx,
                                                                const double* __restrict__ ty,
                                                                int64_t nt,
                                                                double* __restrict__ part) {
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nt;
         i += (int64_t)gridDim.x * 256) {
        double x = tx[i], y = ty[i];
        xmin = fmin(xmin, x);
        xmax = fmax(xmax, x);
        ymin = fmin(ymin, y);
        ymax = fmax(ymax, y);
    }
    xmin = wave_min(xmin);
    xmax = wave_max(xmax);
    ymin = wave_min(ymin);
    ymax = wave_max(ymax);
    __shared__ double s[4][4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s[w][0] = xmin;
        s[w][1] = xmax;
        s[w][2] = ymin;
        s[w][3] = ymax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[4 * blockIdx.x + 0] = fmin(fmin(s[0][0], s[1][0]), fmin(s[2][0], s[3][0]));
        part[4 * blockIdx.x + 1] = fmax(fmax(s[0][1], s[1][1]), fmax(s[2][1], s[3][1]));
        part[4 * blockIdx.x + 2] = fmin(fmin(s[0][2], s[1][2]), fmin(s[2][2], s[3][2]));
        part[4 * blockIdx.x + 3] = fmax(fmax(s[0][3], s[1][3]), fmax(s[2][3], s[3][3]));
    }
}

struct PackArgs {
    const double* sx;
    const double* sy;
    const double* ch[8];     // nullable density channels -> record rows 2..9
    const double* mulby[8];  // nullable elementwise multiplier per channel
    double mul[8];           // constant multiplier per channel
    int pw[8];               // 1: multiply by the coordinate scale s as well
    int corr_ch;             // channel whose sum feeds ApplyParams::corr (-1: none)
    int corr2_ch;
    int use_scale;           // 0: sh = 0
    int exp_hi;              // table upper exponent
    int stokes_ng;           // 1: channel 6 = ch4*ch2 + ch5*ch3 (n.g' of the stresslet)
    double fixed_scale;      // != 0: scale coordinates by this factor instead of 2^sh
};

// (k r)^2 from which a launch whose pairs are ALL that far apart leaves the table kernel
#define IPDE_MODHELM_FAR_KR2 64.0

// single block of 1024 threads; ns_alloc = whole batches (multiple of 8) >= ns
__global__ __launch_bounds__(1024) static void ipde_pack_kernel(PackArgs a, int64_t ns,
                                                                 int64_t ns_alloc,
                                                                 const double* __restrict__ bbox_part,
                                                                 int nbbox,
                                                                 double* __restrict__ rec,
                                                                 ApplyParams* __restrict__ prm) {
    __shared__ double red[16][8];
    __shared__ int s_sh, s_win, s_force;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    int sh = 0, win = 0, force_generic = 0;
    if (a.use_scale) {
        // bounding boxes of the targets (t: from the partials of the bounding-box kernel) and of the
        // sources (s) kept apart: their union scales the tables, their GAP bounds the smallest distance
        double t[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
        double b[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
        for (int i = tid; i < nbbox; i += 1024) {
            t[0] = fmin(t[0], bbox_part[4 * i + 0]);
            t[1] = fmax(t[1], bbox_part[4 * i + 1]);
            t[2] = fmin(t[2], bbox_part[4 * i + 2]);
            t[3] = fmax(t[3], bbox_part[4 * i + 3]);
        }
        for (int64_t i = tid; i < ns; i += 1024) {
            double x = a.sx[i], y = a.sy[i];
            b[0] = fmin(b[0], x);
            b[1] = fmax(b[1], x);
            b[2] = fmin(b[2], y);
            b[3] = fmax(b[3], y);
        }
        t[0] = wave_min(t[0]);
        t[1] = wave_max(t[1]);
        t[2] = wave_min(t[2]);
        t[3] = wave_max(t[3]);
        b[0] = wave_min(b[0]);
        b[1] = wave_max(b[1]);
        b[2] = wave_min(b[2]);
        b[3] = wave_max(b[3]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[w][c] = t[c];
                red[w][4 + c] = b[c];
            }
        }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < 16; ++i) {
                t[0] = fmin(t[0], red[i][0]);
                t[1] = fmax(t[1], red[i][1]);
                t[2] = fmin(t[2], red[i][2]);
                t[3] = fmax(t[3], red[i][3]);
                b[0] = fmin(b[0], red[i][4]);
                b[1] = fmax(b[1], red[i][5]);
                b[2] = fmin(b[2], red[i][6]);
                b[3] = fmax(b[3], red[i][7]);
            }
            const double xmin = fmin(t[0], b[0]), xmax = fmax(t[1], b[1]);
            const double ymin = fmin(t[2], b[2]), ymax = fmax(t[3], b[3]);
            // distance between the two boxes (0 when they overlap or one side is empty)
            const double gx = fmax(0.0, fmax(t[0] - b[1], b[0] - t[1]));
            const double gy = fmax(0.0, fmax(t[2] - b[3], b[2] - t[3]));
            const double gap2 = (gx < INFINITY && gy < INFINITY) ? gx * gx + gy * gy : 0.0;
            double ddx = xmax - xmin, ddy = ymax - ymin;
            double D2 = ddx * ddx + ddy * ddy;
            int shv = 0, force = 0;
            if (D2 > 0.0 && D2 < INFINITY) {
                // (margin: D2 and every pair's d2 carry a few roundings; the kernels do not
                // watch the upper end of the table)
                int e = ilogb(D2 * (1.0 + 0x1p-30));   // D2 in [2^e, 2^(e+1))
                int num = a.exp_hi - 1 - e;             // need 2*sh + e + 1 <= exp_hi
                shv = (num >= 0) ? (num / 2) : -((-num + 1) / 2);
                if (shv > 400) {
                    shv = 400;
                }
                if (shv < -400) {
                    shv = -400;
                    force = 1;     // cannot be scaled under the top of the table
                }
            }
            s_sh = shv;
            s_force = force;
            // modified-Helmholtz table window (layer_modhelm.hip): smallest window whose top
            // binade 2^(11 + 2w) covers y_max = (k * diameter)^2
            int wv = 0;
            if (a.fixed_scale != 0.0 && D2 > 0.0 && D2 < INFINITY) {
                // (margin: D2 and every pair's y carry a few roundings; the kernel does not watch the
                // upper end of the table)
                int ey = ilogb(D2 * a.fixed_scale * a.fixed_scale * (1.0 + 0x1p-30)) + 1;   // y_max < 2^ey
                wv = (ey - 11 + 1) / 2;
                if (ey <= 11) wv = 0;
                if (wv > 7) wv = 8;      // beyond the last window (k * diameter > 5800): no table, generic body
                // every pair at k r >= 8: the whole result is of the size e^(-k r) at which the
                // degree-5 table's RELATIVE error (K1 part: 1.5e-12 measured on a set with k r >= 12;
                // 2e-12 at k r = 20, 7e-12 at 30) is no longer hidden under near-field values —
                // full-precision body (IPDE_MODHELM_FAR_KR2)
                if (gap2 * a.fixed_scale * a.fixed_scale >= IPDE_MODHELM_FAR_KR2) wv = 8;
            }
            s_win = wv;
        }
        __syncthreads();
        sh = a.fixed_scale != 0.0 ? 0 : s_sh;
        win = s_win;
        force_generic = s_force;
        __syncthreads();
    }
    const double s1 = a.fixed_scale != 0.0 ? a.fixed_scale : ldexp(1.0, sh);
    double sum1 = 0.0, sum2 = 0.0;
    for (int64_t i = tid; i < ns_alloc; i += 1024) {
        const bool real = i < ns;
        const int64_t is = real ? i : ns - 1;  // padding sits on the last real source
        rec[ipde_rec_index(i, 0)] = a.sx[is] * s1;
        rec[ipde_rec_index(i, 1)] = a.sy[is] * s1;
        double vals[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double v = 0.0;
            if (real && a.ch[c]) {
                v = a.ch[c][i] * a.mul[c];
                if (a.mulby[c]) v *= a.mulby[c][i];
                if (a.pw[c] == 1) v *= s1;
            }
            vals[c] = v;
            if (c == a.corr_ch) sum1 += v;
            if (c == a.corr2_ch) sum2 += v;
        }
        if (a.stokes_ng) vals[6] = fma(vals[4], vals[2], vals[5] * vals[3]);
#pragma unroll
        for (int c = 0; c < 8; ++c) rec[ipde_rec_index(i, 2 + c)] = vals[c];
    }
    sum1 = wave_sum(sum1);
    sum2 = wave_sum(sum2);
    if (lane == 0) {
        red[w][0] = sum1;
        red[w][1] = sum2;
    }
    __syncthreads();
    if (tid == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < 16; ++i) {
            t1 += red[i][0];
            t2 += red[i][1];
        }
        const double two_ln2 = 1.3862943611198906188;
        prm->sh = sh;
        prm->pad = a.fixed_scale != 0.0 ? win : force_generic;   // MH: table window; else: force generic
        prm->corr = -two_ln2 * (double)sh * t1;
        prm->corr2 = -two_ln2 * (double)sh * t2;
        prm->inv_s = ldexp(1.0, -sh);
        prm->inv_s2 = ldexp(1.0, -2 * sh);
        prm->scale = s1;
    }
}

// Sum split-source partials in a fixed order (deterministic).
__global__ __launch_bounds__(256) static void ipde_reduce_partials(const double* __restrict__ part,
                                                                    int nchunk, int64_t nt,
                                                                    double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[(size_t)c * nt + i];
    out[i] = s;
}

// bbox + pack on ctx->stream; returns device pointers to the records / params
int ipde_layer_prepare(ipde_ctx* ctx, const PackArgs& pa, int64_t ns, const double* tx,
                       const double* ty, int64_t nt, const double** rec, const ApplyParams** prm);

// One batch (8 sources) of channel row `ch`, wave-uniform -> scalar registers.
struct SrcRow {
    double v[IPDE_SRC_PAD];
    __device__ __forceinline__ void load(const double* __restrict__ rec, int batch, int ch) {
        const double* p = rec + ((size_t)batch * IPDE_SRC_NCH + ch) * IPDE_SRC_PAD;
#pragma unroll
        for (int u = 0; u < IPDE_SRC_PAD; ++u) v[u] = p[u];
    }
};
