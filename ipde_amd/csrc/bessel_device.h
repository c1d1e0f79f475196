// K0 / K1 in double precision on the device — the reference-grade evaluation shared by
// the modified-Helmholtz kernels (generic path) and the Ewald spreading kernel.
// Every translation unit that includes this header owns a copy of the coefficient
// arrays in constant memory and must call ipde_bessel_upload() once before launching.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "bessel_coeffs.h"

namespace {

__device__ __constant__ double c_i0s[IPDE_K_I0S_N];
__device__ __constant__ double c_b0[IPDE_K_B0_N];
__device__ __constant__ double c_j1[IPDE_K_J1_N];
__device__ __constant__ double c_c1[IPDE_K_C1_N];
__device__ __constant__ double c_g0[IPDE_K_G0_N];
__device__ __constant__ double c_g1[IPDE_K_G1_N];

inline hipError_t ipde_bessel_upload() {
    hipError_t e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_i0s), ipde_k_i0s, sizeof(ipde_k_i0s))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_b0), ipde_k_b0, sizeof(ipde_k_b0))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_j1), ipde_k_j1, sizeof(ipde_k_j1))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_c1), ipde_k_c1, sizeof(ipde_k_c1))) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_g0), ipde_k_g0, sizeof(ipde_k_g0))) != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_g1), ipde_k_g1, sizeof(ipde_k_g1));
}

template <int N>
__device__ __forceinline__ double horner(const double* c, double y) {
    double p = c[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) p = fma(p, y, c[i]);
    return p;
}

template <int N>
__device__ __forceinline__ double clenshaw(const double* c, double t) {
    double b1 = 0.0, b2 = 0.0;
    const double t2 = 2.0 * t;
#pragma unroll
    for (int i = N - 1; i >= 1; --i) {
        double b0 = fma(t2, b1, c[i]) - b2;
        b2 = b1;
        b1 = b0;
    }
    return fma(t, b1, c[0]) - b2;
}

// K0(x) (WANT & 1) and K1(x)/x (WANT & 2) for y = x^2 > 0:
//   x <= 2:  K0 = -log(x) I0s(y) + B0(y),  K1/x = [1 + y (log(x) J1(y) + C1(y))]/y
//   x  > 2:  K0 = exp(-x)/sqrt(x) G0(t),   K1/x = exp(-x)/sqrt(x) G1(t)/x,  t = 4/x - 1
template <int WANT>
__device__ __forceinline__ void bessel_k01(double y, double& k0, double& k1x) {
    if (y <= IPDE_BESSEL_XS * IPDE_BESSEL_XS) {
        double lx = 0.5 * log(y);
        if (WANT & 1)
            k0 = fma(-lx, horner<IPDE_K_I0S_N>(c_i0s, y), horner<IPDE_K_B0_N>(c_b0, y));
        if (WANT & 2) {
            double in = fma(lx, horner<IPDE_K_J1_N>(c_j1, y), horner<IPDE_K_C1_N>(c_c1, y));
            k1x = fma(y, in, 1.0) / y;
        }
    } else {
        double x = sqrt(y);
        double rx = 1.0 / x;
        double t = fma(2.0 * IPDE_BESSEL_XS, rx, -1.0);
        double ef = exp(-x) * sqrt(rx);
        if (WANT & 1) k0 = ef * clenshaw<IPDE_K_G0_N>(c_g0, t);
        if (WANT & 2) k1x = ef * clenshaw<IPDE_K_G1_N>(c_g1, t) * rx;
    }
}

}  // namespace
