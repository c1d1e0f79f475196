// Where does lu_panel_kernel (csrc/lu_factor.hip) spend a panel?  Builds it with clock stamps
// (IPDE_LU_STAMPS), runs ONE panel step K of a random tiled matrix and prints, per sub-panel, the
// 100 MHz clock at: start, raw loads done, U block done, left-looking update done (= column 0
// starts), each column's start, write-back start, composition start, moves start.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIPDE_LU_STAMPS tools/lu_panel_probe.hip -o /tmp/lu_panel_probe
#include "../ipde_amd/csrc/lu_factor.hip"
#include <cstdlib>

int ipde_devbuf_reserve(ipde_ctx*, DevBuf&, size_t) { return IPDE_ERR_ALLOC; }

#define CK(x)                                                 \
    do {                                                      \
        hipError_t e = (x);                                   \
        if (e != hipSuccess) {                                \
            printf("%s: %s\n", #x, hipGetErrorString(e));     \
            return 1;                                         \
        }                                                     \
    } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const int K = argc > 2 ? atoi(argv[2]) : 0;
    const int nb = n / 64;
    std::vector<double> h((size_t)n * n);
    srand(3);
    for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
    double* T;
    int *perm, *moves;
    unsigned long long* st;
    CK(hipMalloc(&T, h.size() * 8));
    CK(hipMalloc(&perm, n * 4));
    CK(hipMalloc(&moves, 256 * 4));
    CK(hipMalloc(&st, 8 * 16 * 8));
    CK(hipMemcpy(T, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(perm, 0, n * 4));
    CK(hipMemset(st, 0, 8 * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lu_stamps), &st, sizeof(st)));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((lu_panel_kernel<1024, 4, 8>), dim3(1), dim3(1024), 0, 0, T, nb, K, perm, moves);
        CK(hipDeviceSynchronize());
    }
    unsigned long long s[128];
    CK(hipMemcpy(s, st, sizeof(s), hipMemcpyDeviceToHost));
    printf("n %d, panel K = %d (%d rows); microseconds since the sub-panel's start\n", n, K, (nb - K) * 64);
    printf("sub  raw   ublk  left |  columns 0..7 start                                      | wb    comp  moves  total\n");
    for (int sp = 0; sp < 8; ++sp) {
        const unsigned long long* q = s + sp * 16;
        auto us = [&](int k) { return (double)(q[k] - q[0]) * 0.01; };
        printf("%2d %5.1f %5.1f %5.1f |", sp, us(1), sp ? us(2) : 0.0, us(3));
        for (int j = 0; j < 8; ++j) printf(" %5.1f", us(3 + j));
        const double end = sp < 7 ? (double)(s[(sp + 1) * 16] - q[0]) * 0.01 : -1.0;
        printf(" | %5.1f %5.1f %5.1f  %5.1f\n", us(11), us(12), us(13), end);
    }
    return 0;
}
