// Probe: do fp64 MFMA (v_mfma_f64_16x16x4_f64) and fp64 VALU FMAs overlap on gfx950?
// Three loops of identical VALU work (NV independent fp64 FMA chains), with 0, 1 or 2
// MFMAs issued per block of NV*? VALU instructions.  If the matrix pipe runs in the
// VALU's shadow the times are equal; if the fp64 MFMA borrows the vector FMA lanes the
// time grows by 64 cycles per MFMA.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_coexec_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NMFMA, int NVALU>
__global__ __launch_bounds__(256) void probe(double* out, int iters, double a, double b) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a * (threadIdx.x + i);
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double ma = a + threadIdx.x, mb = b - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (NMFMA >= 1) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
        if (NMFMA >= 2) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, ma, acc1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NVALU / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fma(v[i], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    s += acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NMFMA, int NVALU>
static float run(double* d_out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NMFMA, NVALU><<<blocks, 256>>>(d_out, iters, 0.999999, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NMFMA, NVALU><<<blocks, 256>>>(d_out, iters, 0.999999, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    int blocks = 256 * 8;  // 8 waves/SIMD resident
    int iters = 20000;
    double* d_out;
    hipMalloc(&d_out, sizeof(double) * blocks * 256);
    printf("blocks=%d iters=%d (times in ms)\n", blocks, iters);
    printf("VALU=40/iter : mfma0 %.3f  mfma1 %.3f  mfma2 %.3f\n", run<0, 40>(d_out, blocks, iters),
           run<1, 40>(d_out, blocks, iters), run<2, 40>(d_out, blocks, iters));
    printf("VALU=16/iter : mfma0 %.3f  mfma1 %.3f  mfma2 %.3f\n", run<0, 16>(d_out, blocks, iters),
           run<1, 16>(d_out, blocks, iters), run<2, 16>(d_out, blocks, iters));
    printf("VALU=0/iter  : mfma1 %.3f  mfma2 %.3f\n", run<1, 0>(d_out, blocks, iters),
           run<2, 0>(d_out, blocks, iters));
    for (int b : {256 * 2, 256 * 4}) {
        printf("blocks=%d VALU=40: mfma0 %.3f mfma1 %.3f\n", b, run<0, 40>(d_out, b, iters),
               run<1, 40>(d_out, b, iters));
    }
    hipFree(d_out);
    return 0;
}
