// Probe: variants of the three kernels of the 2-D FFT pipeline (csrc/fft2d.hip) timed one by one at
// 2048^2 (and 4096^2), each checked against the library kernels' result.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_tmp/fft_probe tools/fft_probe.hip && build_tmp/fft_probe
#include "../ipde_amd/csrc/fft2d.hip"

int ipde_devbuf_reserve(ipde_ctx*, DevBuf&, size_t) { return 0; }

namespace {

// ---- rows: K rows per wave, the next row's loads in flight under the current row's transform --
template <int NY, int K, int SLEEP>
__global__ __launch_bounds__(256) void row_r2c_pipe(const double* __restrict__ f, cd* __restrict__ W,
                                                    const cd* __restrict__ tw_h,
                                                    const cd* __restrict__ tw_ny) {
    constexpr int H = NY / 2;
    using G = Cfg<H>;
    constexpr int T = G::T, P = G::P, RPW = 256 / T;
    static_assert(T == 64, "a wave per row");
    extern __shared__ double2 lds_raw[];
    cd* lds = (cd*)lds_raw;
    const int tid = threadIdx.x, sub = tid / T, t = tid % T;
    const int64_t row0 = ((int64_t)blockIdx.x * RPW + sub) * K;
    cd* buf = lds + sub * lds_slots<H>();
    if (SLEEP > 0 && (sub & 1)) __builtin_amdgcn_s_sleep(SLEEP);
    cd v[P], vn[P];
    {
        const double2* src = (const double2*)(f + row0 * NY);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            double2 a = src[t + T * q];
            v[q] = cd{a.x, a.y};
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t row = row0 + k;
        if (k + 1 < K) {
            const double2* src = (const double2*)(f + (row + 1) * NY);
#pragma unroll
            for (int q = 0; q < P; ++q) {
                double2 a = src[t + T * q];
                vn[q] = cd{a.x, a.y};
            }
        }
        fft_regs<H, -1, true>(v, t, tw_h, buf);
        cd m[P];
        gather_mirror<H, true>(v, m, t, buf);
        cd* dst = W + row * H;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int kk = t + T * q;
            cd zk = v[q], zm = cconj(m[q]);
            cd e = cd{0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y)};
            cd d = cd{0.5 * (zk.x - zm.x), 0.5 * (zk.y - zm.y)};
            cd w = tw_ny[kk];
            cd wd = cmul(w, d);
            cd X = cd{e.x + wd.y, e.y - wd.x};
            if (kk == 0) X = cd{zk.x + zk.y, zk.x - zk.y};
            dst[kk] = X;
        }
        if (k + 1 < K) {
#pragma unroll
            for (int q = 0; q < P; ++q) v[q] = vn[q];
        }
    }
}

// phase floors of the row kernel: PH = 0 load + store only, 1 transform only (no global traffic to speak of)
template <int NY, int PH>
__global__ __launch_bounds__(256) void row_phase(const double* __restrict__ f, cd* __restrict__ W,
                                                 const cd* __restrict__ tw_h, const cd* __restrict__ tw_ny) {
    constexpr int H = NY / 2;
    using G = Cfg<H>;
    constexpr int T = G::T, P = G::P, RPW = 256 / T;
    extern __shared__ double2 lds_raw[];
    cd* lds = (cd*)lds_raw;
    const int tid = threadIdx.x, sub = tid / T, t = tid % T;
    const int64_t row = (int64_t)blockIdx.x * RPW + sub;
    cd* buf = lds + sub * lds_slots<H>();
    cd v[P];
    if (PH == 0) {
        const double2* src = (const double2*)(f + row * NY);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            double2 a = src[t + T * q];
            v[q] = cd{a.x, a.y};
        }
    } else {
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = cd{(double)(t + q), (double)(row - q)};
        fft_regs<H, -1, (T == 64)>(v, t, tw_h, buf);
        cd m[P];
        gather_mirror<H, (T == 64)>(v, m, t, buf);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int k = t + T * q;
            cd zk = v[q], zm = cconj(m[q]);
            cd e = cd{0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y)};
            cd d = cd{0.5 * (zk.x - zm.x), 0.5 * (zk.y - zm.y)};
            cd w = tw_ny[k];
            cd wd = cmul(w, d);
            v[q] = cd{e.x + wd.y, e.y - wd.x};
        }
    }
    cd* dst = W + row * H;
    if (PH == 0) {
#pragma unroll
        for (int q = 0; q < P; ++q) dst[t + T * q] = v[q];
    } else {
        double s = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) s += v[q].x + v[q].y;
        if (s == 1.2345e300) dst[t] = v[0];
    }
}

__global__ void empty_kernel(int* p) {
    if (p && threadIdx.x == 12345) *p = 1;
}

// Timing by graph replay: 20 launches captured once, the graph replayed — the launch rate of the host
// and the clock ramp after an idle phase stay out of the figure (stream launches in a loop measured
// 57 .. 73 us for the three kernels of a solve that take 52.6 us from a graph).
struct Timer {
    hipEvent_t a, b;
    hipStream_t st;
    explicit Timer(hipStream_t s) : st(s) {
        hipEventCreate(&a);
        hipEventCreate(&b);
    }
    template <typename F>
    float run(F f, int reps = 20) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < reps; ++i) f();
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best = 1e30f;
        for (int w = 0; w < 8; ++w) hipGraphLaunch(ge, st);
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(a, st);
            hipGraphLaunch(ge, st);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = fminf(best, 1e3f * ms / reps);
        }
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
        return best;
    }
};

double maxdiff(const void* d_a, const void* d_b, size_t ndoubles) {
    std::vector<double> a(ndoubles), b(ndoubles);
    hipMemcpy(a.data(), d_a, ndoubles * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d_b, ndoubles * 8, hipMemcpyDeviceToHost);
    double m = 0, s = 0;
    for (size_t i = 0; i < ndoubles; ++i) {
        m = fmax(m, fabs(a[i] - b[i]));
        s = fmax(s, fabs(a[i]));
    }
    return m / s;
}

template <int N>
void probe() {
    ipde_ctx ctx;
    hipStreamCreate(&ctx.own_stream);
    ctx.stream = ctx.own_stream;
    Fft2dPlan p;
    if (fft2d_plan_init(&ctx, p, N, N, 3.0 / N, 3.0 / N) != IPDE_OK) {
        printf("plan failed: %s\n", ctx.err.c_str());
        return;
    }
    const size_t n2 = (size_t)N * N;
    std::vector<double> hf(n2);
    unsigned long long s = 88172645463325252ull;
    for (auto& x : hf) {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        x = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    double *f, *u, *u2;
    cd *Wref, *Wt;
    hipMalloc(&f, n2 * 8);
    hipMalloc(&u, n2 * 8);
    hipMalloc(&u2, n2 * 8);
    hipMalloc(&Wref, n2 * 8);
    hipMalloc(&Wt, n2 * 8);
    hipMemcpy(f, hf.data(), n2 * 8, hipMemcpyHostToDevice);
    Timer tm(ctx.own_stream);
    hipStream_t st = ctx.stream;
    printf("---- N = %d ----\n", N);
    printf("empty kernel, 512 x 256:      %7.2f us\n",
           tm.run([&] { hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(256), 0, st, (int*)nullptr); }, 100));
    float t_all = tm.run([&] { fft2d_scalar_solve(&ctx, p, FFT2D_SYM_POISSON, 0.0, f, u); });
    printf("library solve (3 kernels)     %7.2f us\n", t_all);
    float t_r = tm.run([&] { fft2d_rows_forward(&ctx, p, f, 0); });
    hipMemcpyAsync(Wref, p.W[0], n2 * 8, hipMemcpyDeviceToDevice, st);
    float t_c = tm.run([&] { fft2d_cols(&ctx, p, 0, FFT2D_SYM_POISSON, 0, 0.0, 2.0 / ((double)N * N)); });
    float t_i = tm.run([&] { fft2d_rows_inverse(&ctx, p, 0, u2); });
    printf("row_r2c %7.2f  col %7.2f  row_c2r %7.2f  sum %7.2f us\n", t_r, t_c, t_i, t_r + t_c + t_i);

    // rows: K rows per wave / staggered waves
    constexpr int H = N / 2, T = Cfg<H>::T, RPW = 256 / T;
    if constexpr (T == 64) {
        const size_t lds = (size_t)RPW * lds_slots<H>() * sizeof(cd);
        auto try_rows = [&](auto kern, int K, const char* name) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            float t = tm.run([&] {
                hipLaunchKernelGGL(kern, dim3(N / (RPW * K)), dim3(256), lds, st, (const double*)f, Wt,
                                   (const cd*)p.tw_h, (const cd*)p.tw_ny);
            });
            hipStreamSynchronize(st);
            printf("  %-28s %7.2f us   rel diff %.1e\n", name, t, maxdiff(Wt, Wref, n2));
        };
        try_rows(row_phase<N, 0>, 1, "rows: load + store only");
        try_rows(row_phase<N, 1>, 1, "rows: transform only");
        try_rows(row_r2c_pipe<N, 1, 0>, 1, "rows, twiddles read in passes");
        try_rows(row_r2c_pipe<N, 1, 16>, 1, "rows, odd waves sleep 16x64");
        try_rows(row_r2c_pipe<N, 1, 40>, 1, "rows, odd waves sleep 40x64");
        try_rows(row_r2c_pipe<N, 1, 80>, 1, "rows, odd waves sleep 80x64");
        try_rows(row_r2c_pipe<N, 1, 120>, 1, "rows, odd waves sleep 120x64");
    }
    // columns: two per workgroup (two workgroups per CU)
    if constexpr (N == 2048) {
        hipMemcpyAsync(p.W[0], Wref, n2 * 8, hipMemcpyDeviceToDevice, st);
        fft2d_cols(&ctx, p, 0, FFT2D_SYM_POISSON, 0, 0.0, 2.0 / ((double)N * N));
        hipMemcpyAsync(Wt, p.W[0], n2 * 8, hipMemcpyDeviceToDevice, st);   // reference result of the column pass
        auto try_cols = [&](auto kern, int C, const char* name) {
            const size_t lds = (size_t)C * (lds_slots<N>() + 4) * sizeof(cd);
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int nblocks = (N / 2) / C;
            hipMemcpyAsync(p.W[1], Wref, n2 * 8, hipMemcpyDeviceToDevice, st);
            hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C * Cfg<N>::T), lds, st, (cd*)p.W[1], N / 2, nblocks, N,
                               2.0 * M_PI / 3.0, 2.0 * M_PI / 3.0, 0.0, 2.0 / ((double)N * N),
                               (const cd*)p.tw_x, (cd*)nullptr);
            hipStreamSynchronize(st);
            double d = maxdiff(p.W[1], Wt, n2);
            float t = tm.run([&] {
                hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C * Cfg<N>::T), lds, st, (cd*)p.W[1], N / 2, nblocks,
                                   N, 2.0 * M_PI / 3.0, 2.0 * M_PI / 3.0, 0.0, 2.0 / ((double)N * N),
                                   (const cd*)p.tw_x, (cd*)nullptr);
            });
            printf("  %-28s %7.2f us   rel diff %.1e\n", name, t, d);
        };
        try_cols(col_kernel<N, 4, FFT2D_SYM_POISSON, 0>, 4, "cols C=4 (library)");
        try_cols(col_kernel<N, 2, FFT2D_SYM_POISSON, 0>, 2, "cols C=2");
        try_cols(col_kernel<N, 8, FFT2D_SYM_POISSON, 0>, 8, "cols C=8");
    }
    hipFree(f);
    hipFree(u);
    hipFree(u2);
    hipFree(Wref);
    hipFree(Wt);
    fft2d_plan_free(p);
}

}  // namespace

int main() {
    probe<2048>();
    probe<4096>();
    return 0;
}
