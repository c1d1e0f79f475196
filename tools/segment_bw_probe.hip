// Probe: effective bandwidth of column-block access on a row-major (nx, nyh) complex128 array
// as a function of the block width C (segment = C*16 bytes per row), reads and writes.
// Decides the intermediate layout of the hand-written 2-D FFT.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/segprobe tools/segment_bw_probe.hip && /tmp/segprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int C, bool WRITE>
__global__ __launch_bounds__(256) void colblock(double2* __restrict__ a, int nx, int pitch, int ncb,
                                                double2* __restrict__ sink) {
    // XCD-aware: consecutive column blocks on one XCD
    int wg = blockIdx.x;
    int nper = (gridDim.x + 7) / 8;
    int cb = (wg % 8) * nper + wg / 8;
    if (cb >= ncb) return;
    const int lane_c = threadIdx.x % C, lane_r = threadIdx.x / C;
    const int rows_per_it = 256 / C;
    double2 acc{0.0, 0.0};
    for (int r = lane_r; r < nx; r += rows_per_it) {
        double2* p = a + (size_t)r * pitch + cb * C + lane_c;
        if (WRITE) {
            *p = double2{(double)r, (double)cb};
        } else {
            double2 v = *p;
            acc.x += v.x;
            acc.y += v.y;
        }
    }
    if (!WRITE && acc.x == 12345.678) sink[0] = acc;
}

template <int C, bool WRITE>
float run(double2* a, int nx, int pitch, double2* sink) {
    int ncb = pitch / C;
    int grid = ((ncb + 7) / 8) * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((colblock<C, WRITE>), dim3(grid), dim3(256), 0, 0, a, nx, pitch, ncb, sink);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((colblock<C, WRITE>), dim3(grid), dim3(256), 0, 0, a, nx, pitch, ncb, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    for (int n : {2048, 4096}) {
        int nx = n, pitch = n / 2 + 64;   // multiple of 64 columns so every C divides
        size_t bytes = (size_t)nx * pitch * sizeof(double2);
        double2 *a, *sink;
        hipMalloc(&a, bytes);
        hipMalloc(&sink, 64);
        hipMemset(a, 0, bytes);
        printf("n=%d array %.1f MB\n", n, bytes / 1e6);
#define RUN(C)                                                                                   \
    {                                                                                            \
        float r = run<C, false>(a, nx, pitch, sink), w = run<C, true>(a, nx, pitch, sink);       \
        printf("  C=%2d (%4d B segments): read %.1f us %.2f TB/s | write %.1f us %.2f TB/s\n", C, \
               C * 16, r * 1e3, bytes / (r * 1e-3) / 1e12, w * 1e3, bytes / (w * 1e-3) / 1e12);  \
    }
        RUN(1) RUN(2) RUN(4) RUN(8) RUN(16) RUN(64)
        hipFree(a);
        hipFree(sink);
    }
    return 0;
}
