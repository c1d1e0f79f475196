// Where does a block-pair step of the persistent substitution (csrc/dense.hip) spend its time?
// Builds the kernel with in-kernel clock stamps (IPDE_LU_STAMPS), runs the forward pass on a
// synthetic tiled unit-lower system and prints, per workgroup in dependency order, the 100 MHz
// clock at: start, last predecessor's solution seen, update applied, first block published,
// second chain started, second block published.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIPDE_LU_STAMPS tools/lu_persist_probe.hip -o /tmp/lu_probe
#include "../ipde_amd/csrc/dense.hip"
#include <cstdlib>

// (the library entry points in dense.hip come along with the kernels; their one external)
int ipde_devbuf_reserve(ipde_ctx*, DevBuf&, size_t) { return IPDE_ERR_ALLOC; }

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e = (x);                                              \
        if (e != hipSuccess) {                                           \
            printf("%s: %s\n", #x, hipGetErrorString(e));                \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 6400;
    const int nsys = argc > 2 ? atoi(argv[2]) : 1;
    const int nbp = (n + 127) / 128, nb = 2 * nbp;
    const size_t T = 4096;
    std::vector<double> lu((size_t)nb * nb * T, 0.0);
    srand(1);
    for (int I = 0; I < nb; ++I)
        for (int K = 0; K <= I; ++K)
            for (int c = 0; c < 64; ++c)
                for (int r = 0; r < 64; ++r) {
                    double v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
                    if (I == K && r == c) v = 1.0;
                    if (I == K && r < c) v = 0.0;
                    lu[((size_t)I * nb + K) * T + c * 64 + r] = v;
                }
    std::vector<double> b(n);
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) {
        b[i] = rand() / (double)RAND_MAX;
        perm[i] = i;
    }
    double *d_lu, *d_b, *d_work;
    int* d_perm;
    unsigned* d_abort;
    unsigned long long* d_st;
    const size_t slots = (size_t)nbp * 128;
    const size_t bytes = 16 + nsys * slots * 8;
    CK(hipMalloc(&d_lu, lu.size() * 8));
    CK(hipMalloc(&d_b, n * 8));
    CK(hipMalloc(&d_perm, n * 4));
    CK(hipMalloc(&d_work, bytes));
    CK(hipMalloc(&d_abort, 16));
    CK(hipMalloc(&d_st, (size_t)nsys * nbp * 4 * 8 * 8));
    CK(hipMemcpy(d_lu, lu.data(), lu.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, b.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_abort, 0, 16));
    LuPersist fw{};
    fw.nsys = nsys;
    fw.nbp_max = nbp;
    fw.ticket = (unsigned*)d_work;
    fw.abort_word = d_abort;
    fw.stamps = d_st;
    for (int s = 0; s < nsys; ++s) {
        fw.lu[s] = d_lu;
        fw.perm[s] = d_perm;
        fw.b[s] = d_b;
        fw.n[s] = n;
        fw.slots[s] = (double*)((char*)d_work + 16) + s * slots;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(d_st, 0, (size_t)nsys * nbp * 4 * 8 * 8));
        CK(hipMemsetD32Async((hipDeviceptr_t)d_work, (int)LU_SENTINEL32, bytes / 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(lu_subst_persistent<true>, dim3(nsys * nbp), dim3(DT), 0, 0, fw);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> st((size_t)nsys * nbp * 4 * 8);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned ab;
    CK(hipMemcpy(&ab, d_abort, 4, hipMemcpyDeviceToHost));
    printf("n %d nsys %d: forward pass %.1f us (%.2f us per pair), abort %u\n", n, nsys, ms * 1e3, ms * 1e3 / nbp, ab);
    // system 0 only: ticket = pos * nsys
    const unsigned long long t0 = st[0];
    printf("pos  start  seen(w0) seen(w1) upd(w0) upd(w1)  pub1(w0)  chain2(w1) pub2(w1)   [us]\n");
    double prev = 0;
    for (int pos = 0; pos < nbp; ++pos) {
        const size_t w0 = ((size_t)(pos * nsys) * 4 + 0) * 8, w1 = ((size_t)(pos * nsys) * 4 + 1) * 8;
        auto us = [&](unsigned long long v) { return v ? (double)(v - t0) * 0.01 : -1.0; };
        printf("%3d %6.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f   step %.2f\n", pos, us(st[w0 + 0]), us(st[w0 + 1]),
               us(st[w1 + 1]), us(st[w0 + 2]), us(st[w1 + 2]), us(st[w0 + 3]), us(st[w1 + 4]), us(st[w1 + 5]),
               us(st[w1 + 5]) - prev);
        prev = us(st[w1 + 5]);
    }
    return 0;
}
