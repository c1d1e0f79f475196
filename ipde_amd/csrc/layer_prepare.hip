// Per-apply device-side preparation: target bounding box, coordinate scale,
// packed source records.  No host synchronisation.
#include "layer_pack.h"

// Layout of ctx->src_pack: [ApplyParams (64 B)] [batch-SoA records]
int ipde_layer_prepare(ipde_ctx* ctx, const PackArgs& pa, int64_t ns, const double* tx,
                       const double* ty, int64_t nt, const double** rec,
                       const ApplyParams** prm) {
    const size_t hdr = 64;
    const int64_t ns_alloc = ceil_div64(ns, IPDE_SRC_PAD) * IPDE_SRC_PAD;
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->src_pack,
                                 hdr + (size_t)ns_alloc * IPDE_SRC_NCH * sizeof(double)));
    IPDE_TRY(ipde_devbuf_reserve(ctx, ctx->scratch, 4 * IPDE_BBOX_BLOCKS * sizeof(double)));
    ApplyParams* d_prm = (ApplyParams*)ctx->src_pack.p;
    double* d_rec = (double*)((char*)ctx->src_pack.p + hdr);
    int nbbox = 0;
    if (pa.use_scale) {
        nbbox = (int)std::min<int64_t>(IPDE_BBOX_BLOCKS, ceil_div64(nt, 256));
        hipLaunchKernelGGL(ipde_bbox_kernel, dim3(nbbox), dim3(256), 0, ctx->stream, tx, ty, nt,
                           (double*)ctx->scratch.p);
    }
    hipLaunchKernelGGL(ipde_pack_kernel, dim3(1), dim3(1024), 0, ctx->stream, pa, ns, ns_alloc,
                       (const double*)ctx->scratch.p, nbbox, d_rec, d_prm);
    IPDE_HIP_CHECK(ctx, hipGetLastError());
    *rec = d_rec;
    *prm = d_prm;
    return IPDE_OK;
}
