/*
 * ipde_hip.h — C ABI of libipde_hip.so, the MI355X (gfx950) implementation of
 * dbstein/ipde's hot path: boundary->target layer-potential sums, periodic FFT
 * grid solves / spectral + 4th-order derivatives, and the Chebyshev x Fourier
 * annular solves.
 *
 * Conventions (all entry points)
 *   - return value is a status code (IPDE_OK == 0); no exceptions cross the ABI;
 *     ipde_last_error(ctx) gives a human readable message for the last failure.
 *   - all arrays are IEEE fp64 (complex = interleaved re,im fp64), C-contiguous.
 *   - `loc` says where EVERY array pointer of that call lives: IPDE_HOST (the
 *     library stages through its own device workspace) or IPDE_DEVICE (zero-copy,
 *     work is queued on the context's stream; call ipde_ctx_sync before reading
 *     results from another stream / the host).
 *   - densities are ALREADY multiplied by the quadrature weights: the reference
 *     multiplies `ch*src.weights` on the Python side of its own boundary
 *     (ipde/solvers/internals/poisson.py:31, modified_helmholtz.py:32,
 *     stokes.py:29) and so does our Python shim.
 *   - one context per GPU per process (one process per GPU); calls on one context
 *     are not re-entrant (the reference is single-threaded Python as well).
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * the dbstein/ipde tree).
 */
#ifndef IPDE_HIP_H
#define IPDE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ipde_ctx ipde_ctx;
typedef struct ipde_fft_plan ipde_fft_plan;
typedef struct ipde_annular_scalar ipde_annular_scalar;
typedef struct ipde_annular_stokes ipde_annular_stokes;

enum { IPDE_HOST = 0, IPDE_DEVICE = 1 };

enum {
    IPDE_OK = 0,
    IPDE_ERR_INVALID = 1,   /* bad argument (null pointer, negative size, bad enum) */
    IPDE_ERR_HIP = 2,       /* a HIP runtime call failed                             */
    IPDE_ERR_FFT = 3,       /* a rocFFT call failed                                  */
    IPDE_ERR_NOGPU = 4,     /* no usable gfx950 device                               */
    IPDE_ERR_ALLOC = 5,     /* device allocation failed                              */
    IPDE_ERR_NOCONV = 6     /* GMRES hit maxiter before reaching tol (result is
                               still written; iters/resid report what was reached) */
};

/* flags for the layer-potential applies */
enum {
    IPDE_FLAG_NONE = 0,
    /* skip source/target pairs at zero distance (a point set evaluated onto
       itself); without it such a pair yields inf/nan exactly like the formula */
    IPDE_FLAG_SKIP_COINCIDENT = 1,
    /* force the reference-grade generic kernel (libdevice log / no LDS table) */
    IPDE_FLAG_GENERIC_MATH = 2
};

/* ------------------------------------------------------------------------- */
/* context                                                                   */

const char* ipde_version(void);

/* Owns: HIP stream, device workspaces, LDS math tables, rocFFT plan cache.
   device_id < 0 selects the current device. */
int ipde_ctx_create(int device_id, ipde_ctx** ctx);
int ipde_ctx_destroy(ipde_ctx* ctx);
int ipde_ctx_sync(ipde_ctx* ctx);
/* Use an externally owned hipStream_t (e.g. torch's current stream) for all
   subsequent work; pass NULL to return to the context's own stream, hipStreamLegacy
   ((hipStream_t)1) for the legacy default stream (what the Python host gives the
   process-wide context: torch's default stream is that one, and a solve that
   alternates library kernels with torch operations then stays on one queue). */
int ipde_ctx_set_stream(ipde_ctx* ctx, void* hip_stream);
void* ipde_ctx_get_stream(ipde_ctx* ctx);
/* The same for the legacy default stream, with the handle taken from the HIP header on the C side
   (callers need not know that hipStreamLegacy is (hipStream_t)1).  A context's OWN stream is created
   blocking (hipStreamDefault) on purpose: it orders itself against the legacy default stream, where a
   host framework allocates and fills the buffers it hands to a private context; the price is that a
   launch on the legacy stream waits for every such stream. */
int ipde_ctx_use_legacy_stream(ipde_ctx* ctx);
/* The context's own stream replaced by a non-blocking stream of the lowest priority the device offers (ordering
   against other streams is then the caller's: events): for background work such as the factorisations of a set-up,
   which must not make short kernels of the foreground streams queue behind it (ipde_amd/qfs.py: _OwnAsyncLU). */
int ipde_ctx_use_background_stream(ipde_ctx* ctx);
const char* ipde_last_error(ipde_ctx* ctx);
/* Tuning knobs (kernel geometry variants); never changes results beyond rounding.
   Names: "laplace_variant", "stokes_variant", "dense_pairs", "annular_grouped",
   "fft2d" (1: hand-written 2-D FFT pipeline on power-of-two grids, 0: rocFFT),
   "interp_shifted" (1: ipde_grid_interp through four shifted coarse transforms at any size),
   "timing_split" (1: ipde_laplace_apply_patches_far records one event pair per stage — parents' coefficients,
   blocks' coefficients, patches — instead of one around the three),
   "interp_band" (1, default: ipde_grid_interp / _fields in the band form — oversampled transform along x only
   for the kept columns, exact sums along y per point; 0: full oversampled fine grids and a 2-D window gather),
   "dense_persistent", "annular_fused_fft", "gmres_graphs", "gmres_lookahead" (1: inner iteration
   j + 1 of the annular GMRES enters the stream before the host has read column j; same bits),
   "gmres_fused_scale" (1: the Arnoldi normalisation inside the preconditioner's kernel; same bits),
   "gmres_persistent" (1: the scalar annular GMRES runs its first cycle in ONE launch with the Arnoldi
   bookkeeping on the device; inner products summed in another order: last-bit differences; default 0 —
   measured slower than the launch-per-stage cycle),
   "modhelm_variant" (targets per lane of the modified Helmholtz table kernel). */
int ipde_ctx_set_option(ipde_ctx* ctx, const char* name, int value);
/* Current value of a knob (so that a caller can restore what it found). */
int ipde_ctx_get_option(ipde_ctx* ctx, const char* name, int* value);
/* Duration in ms of the dominant kernel of the LAST layer-potential apply, measured with
   hipEvents on the context's stream (0 if timing is off).  Switching timing on starts a new
   measurement (the history below is emptied). */
int ipde_ctx_enable_timing(ipde_ctx* ctx, int on);
int ipde_ctx_last_kernel_ms(ipde_ctx* ctx, double* ms);
/* Durations (ms) of the dominant kernel of the applies issued since ipde_ctx_enable_timing(ctx, 1),
   oldest first, at most the last 256 and at most `cap`: every apply records its own event pair
   on the context's stream, so a timed loop needs no host synchronisation inside and is resolved
   here afterwards (waits for the last pair).  *n = number written. */
int ipde_ctx_kernel_ms_history(ipde_ctx* ctx, double* ms, int cap, int* n);

/* ------------------------------------------------------------------------- */
/* One host process, several GPUs ("ipde_ctx_create(ndev, dev_ids, &ctx) owns   */
/* streams, rocFFT plans, RCCL comm"; §8(e): targets split over the devices)                  */

typedef struct ipde_multi ipde_multi;
#define IPDE_MULTI_FORCE_COMM 1   /* ipde_multi_create: build the RCCL communicator for one device too */

/* One context (stream, plans, tables) per device of dev_ids and, for ndev > 1 (or with
   IPDE_MULTI_FORCE_COMM), an RCCL communicator over them (ncclCommInitAll; librccl is loaded at run
   time, here).  The devices must be distinct gfx950 devices. */
int ipde_multi_create(int ndev, const int* dev_ids, int flags, ipde_multi** m);
int ipde_multi_destroy(ipde_multi* m);
int ipde_multi_ndev(ipde_multi* m, int* ndev);
int ipde_multi_ctx(ipde_multi* m, int i, ipde_ctx** ctx);       /* the context of device i (borrowed) */
int ipde_multi_has_comm(ipde_multi* m, int* yes);
const char* ipde_multi_last_error(ipde_multi* m);
/* A fixed target set (host arrays) split into ndev contiguous, balanced slices, slice i resident on
   device i: what the reference's solvers evaluate onto (grid_pnai, radial_targ). */
int ipde_multi_set_targets(ipde_multi* m, int64_t nt, const double* tx, const double* ty);
int ipde_multi_target_slice(ipde_multi* m, int i, int64_t* start, int64_t* stop);
/* ipde_laplace_apply / ipde_modhelm_apply / ipde_stokes_apply onto the resident targets, all arrays
   in host memory: the sources go up once and reach the other devices by ncclBroadcast over xGMI, the
   devices sum their slices side by side on their own streams, the slices come back into `out`
   (nt doubles; Stokes: out_u, out_v and, unless NULL, out_p).  No collective on targets or results. */
int ipde_multi_laplace_apply(ipde_multi* m, int64_t ns, const double* sx, const double* sy,
                             const double* w_sigma, const double* nx, const double* ny, const double* w_tau,
                             double* out, int flags);
int ipde_multi_modhelm_apply(ipde_multi* m, double k, int64_t ns, const double* sx, const double* sy,
                             const double* w_sigma, const double* nx, const double* ny, const double* w_tau,
                             double* out, int flags);
int ipde_multi_stokes_apply(ipde_multi* m, int64_t ns, const double* sx, const double* sy,
                            const double* wfx, const double* wfy, const double* nx, const double* ny,
                            const double* wdx, const double* wdy, double* out_u, double* out_v, double* out_p,
                            int flags);

/* ------------------------------------------------------------------------- */
/* layer potentials                                        */

/*
 * Laplace single+double layer:
 *   out_i = sum_j [ -(1/2pi) log|t_i-s_j| * w_sigma_j
 *                   +(1/2pi) (n_j.(t_i-s_j))/|t_i-s_j|^2 * w_tau_j ]
 * Replaces pybie2d Laplace_Layer_Apply(src, trg, charge=, dipstr=, backend='fly')
 * as called by PoissonHelper._define_layer_apply
 * (ipde/solvers/internals/poisson.py:27-36), the Grid_Evaluator closure
 * (ipde/solvers/multi_boundary/poisson.py:57-62) and
 * examples/interior_poisson.py:89.
 * w_sigma NULL => no single layer; w_tau NULL => no double layer (then nx,ny may
 * be NULL).  Sign of the SLP pinned by ipde/grid_evaluators/
 * laplace_grid_evaluator.py:8-12; DLP sign by the interior jump D-I/2
 * (examples/interior_poisson.py:19,84).
 */
int ipde_laplace_apply(ipde_ctx* ctx, int loc,
                       int64_t ns, const double* sx, const double* sy,
                       const double* w_sigma,
                       const double* nx, const double* ny, const double* w_tau,
                       int64_t nt, const double* tx, const double* ty,
                       double* out, int flags);

/*
 * The same sums for targets handed over as 4 x 4 tensor patches (device pointers only):
 * patch p holds the sixteen targets (xs[a], ys[b]), a, b = 0..3, with
 *   pxy[a * np + p] = xs[a], pxy[(4 + a) * np + p] = ys[a]       (8 rows of np doubles)
 *   pout[(4 a + b) * np + p] = position of target (a, b) in `out` (16 rows of np int32)
 * A C-ordered grid list with a band removed — the reference's grid_pnai targets,
 * ipde/solvers/multi_boundary/scalar.py:63-71 — splits into its full 4 x 4 tiles plus a
 * remainder (ipde_amd/target_plan.py); the tiles come here (one add per pair for d^2), the
 * remainder goes through ipde_laplace_apply.  A negative pout entry marks a point of the
 * patch that is not a target (a tile the band cut into): computed, not stored.  Entries of
 * `out` no patch names are not touched.  Same values as ipde_laplace_apply to a rounding of d^2.
 */
int ipde_laplace_apply_patches(ipde_ctx* ctx,
                               int64_t ns, const double* sx, const double* sy,
                               const double* w_sigma,
                               const double* nx, const double* ny, const double* w_tau,
                               int64_t np, const double* pxy, const int32_t* pout,
                               double* out);

/*
 * The same sums again with the FAR sources of every block of patches collapsed into a local
 * (Taylor) expansion about the block's centre — the place of the reference's grid_backend='fmm2d'
 * (fmm2dpy.rfmm2d at eps = 1e-14, ipde/solvers/internals/poisson.py:28-32) and 'flexmm'
 * (ipde/solvers/multi_boundary/poisson.py:50-55) choices, here one level deep and to rounding:
 * 64 consecutive patches must be ONE 8 x 8 block of tiles (ipde_target_plan_build_blocks with
 * block 8 x 8 and pad_blocks = 1).  With c the block's centre and r its half-diagonal, a source
 * z_j beyond 4 r of c enters 27 complex coefficients (log|z - z_j|^2 = log|c - z_j|^2 -
 * 2 Re sum_k ((z - c)/(z_j - c))^k / k, ratio <= 1/4, 26 terms: truncation 3e-18 of sum|w_sigma|);
 * nearer sources are summed pair by pair exactly as in ipde_laplace_apply_patches.  Same
 * arguments and output convention as ipde_laplace_apply_patches; values agree with it to a few
 * roundings of the largest partial sum.
 * Two levels: sixteen consecutive blocks (1024 patches) are taken as one PARENT block — the plan
 * builder orders the blocks in Z order so that they are a 4 x 4 group — whose sources beyond four
 * parent radii enter the parent's coefficients once instead of sixteen blocks' (the same holds for
 * the modified Helmholtz and Stokes forms below).  The geometry of blocks and parents is taken from
 * the bounding boxes of their patches: ANY list of patches gives correct sums, a plan from
 * ipde_target_plan_build_blocks the fast ones.
 */
int ipde_laplace_apply_patches_far(ipde_ctx* ctx,
                                   int64_t ns, const double* sx, const double* sy,
                                   const double* w_sigma,
                                   const double* nx, const double* ny, const double* w_tau,
                                   int64_t np, const double* pxy, const int32_t* pout,
                                   double* out);

/*
 * The far-field form for the radial grid of an annulus: targets (M, N) row-major (tx, ty: DEVICE, M * N
 * doubles each), column j = the M points of one radial line, neighbouring columns neighbouring lines
 * (ipde/embedded_boundary.py:280-358 `radial_x`, `radial_y` raveled).  Single layer; blocks of 64 columns, a
 * block's far sources in its local expansion, near batches pair by pair: the radial sums of correct()
 * (ipde/solvers/internals/scalar.py:113-114); w_sigma and / or (nx, ny, w_tau) as in ipde_laplace_apply (either may be
 * NULL).  out: DEVICE, M * N doubles.  Any (M, N) array gives correct sums
 * (a block's disc is the bounding box of its 64 columns over ALL rows); the form is FAST when neighbouring
 * columns are neighbouring lines.  Modified Helmholtz: sources beyond k d > 45 of a block are dropped
 * (K0(45) = 5e-21: an ABSOLUTE bound — a target with only such sources gets 0, not a relatively accurate value).
 */
int ipde_laplace_apply_columns_far(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                   const double* w_sigma,
                                   const double* nx, const double* ny, const double* w_tau,
                                   int M, int64_t N, const double* tx, const double* ty, double* out);

/*
 * The cut of a target list (HOST arrays x, y of nt points) into those patches, on the host — no
 * GPU work, callable from any thread.  Grid lines are the coordinate values at least
 * line_min_points points share (exact comparisons); tiles of the lattice of lines that hold a
 * point become patches — all of them if the list fills at least partial_min_fill of their 16
 * points on average (missing points: pout = -1), else the full tiles only; patches are ordered in
 * blocks of block_i x block_j tiles (consecutive lanes take consecutive patches); fewer than
 * min_patches patches, or a list that is no grid: np = 0.  The remainder (list positions in
 * increasing order) is everything no patch holds.  ipde_amd/target_plan.py calls this for the
 * reference's grid_pnai lists (ipde/ebdy_collection.py:426-429).
 * build -> sizes -> export into caller-allocated HOST buffers (pxy: 8 np doubles, pout: 16 np
 * int32, rest: nrest int64) -> destroy.
 */
typedef struct ipde_target_plan ipde_target_plan;
int ipde_target_plan_build(int64_t nt, const double* x, const double* y,
                           int block_i, int block_j, double partial_min_fill,
                           int64_t min_patches, int line_min_points,
                           ipde_target_plan** plan);
/* the same; pad_blocks != 0: every block that holds a patch is filled up to block_i * block_j
   patches with copies of its first one that store nothing (pout = -1): patches
   [k B, (k + 1) B), B = block_i block_j, are then exactly one block of tiles; the blocks come in
   Z (Morton) order of their block coordinates: 4^l consecutive blocks are a 2^l x 2^l group */
int ipde_target_plan_build_blocks(int64_t nt, const double* x, const double* y,
                                  int block_i, int block_j, double partial_min_fill,
                                  int64_t min_patches, int line_min_points, int pad_blocks,
                                  ipde_target_plan** plan);
int ipde_target_plan_sizes(const ipde_target_plan* plan, int64_t* np, int64_t* nrest);
int ipde_target_plan_export(const ipde_target_plan* plan, double* pxy, int32_t* pout, int64_t* rest);
int ipde_target_plan_destroy(ipde_target_plan* plan);

/*
 * Modified Helmholtz (k^2 - Lap) single+double layer:
 *   out_i = sum_j [ (1/2pi) K0(k r) w_sigma_j + (k/2pi) K1(k r) (n_j.d)/r w_tau_j ]
 * Replaces pybie2d Modified_Helmholtz_Layer_Apply(src, trg, charge=, k=) as
 * called by ModifiedHelmholtzHelper._define_layer_apply
 * (ipde/solvers/internals/modified_helmholtz.py:28-37); SLP pinned by
 * ipde/grid_evaluators/modified_helmholtz_grid_evaluator.py:8-9.
 */
int ipde_modhelm_apply(ipde_ctx* ctx, int loc, double k,
                       int64_t ns, const double* sx, const double* sy,
                       const double* w_sigma,
                       const double* nx, const double* ny, const double* w_tau,
                       int64_t nt, const double* tx, const double* ty,
                       double* out, int flags);

/*
 * The single-layer sums K0(k r) w_sigma / (2 pi) onto a target list handed over as 4 x 4 patches whose
 * groups of 64 are 8 x 8 blocks of tiles (ipde_target_plan_build_blocks with pad_blocks = 1; layout as
 * in ipde_laplace_apply_patches): every block's sources beyond four block radii enter 27 complex
 * coefficients of a local expansion (Graf's addition theorem: K0(|z - z_j|) = sum_m eps_m K_m(rho_j)
 * I_m(rho) cos(m (phi - phi_j)) about the block's centre, scaled so that nothing over- or underflows),
 * nearer batches of eight sources are summed pair by pair through the table of ipde_modhelm_apply.
 * Blocks whose half-diagonal exceeds 1/k keep every source pair by pair.  The place of the
 * reference's grid_backend='fmm2d' branch (fmm2dpy.hfmm2d with zk = i k,
 * ipde/solvers/internals/modified_helmholtz.py:29-35).  Single layer (w_sigma) and / or double layer
 * (nx, ny, w_tau; either may be NULL): the dipole's coefficients follow from the single layer's by the ladder
 * relations of K_m (the double layer is the source-side directional derivative), one family for both.  DEVICE pointers.
 */
int ipde_modhelm_apply_patches_far(ipde_ctx* ctx, double k, int64_t ns, const double* sx, const double* sy,
                                   const double* w_sigma,
                                   const double* nx, const double* ny, const double* w_tau,
                                   int64_t np, const double* pxy, const int32_t* pout, double* out);

/*
 * The same far-field form for the radial grid of an annulus: targets (M, N) row-major (tx, ty: DEVICE,
 * M * N doubles each), column j = the M points of one radial line, neighbouring columns neighbouring lines
 * (ipde/embedded_boundary.py:280-358 `radial_x`, `radial_y` raveled).  Blocks of 64 columns; a block's far
 * sources in its local expansion, the near batches pair by pair.  The radial sums of correct()
 * (ipde/solvers/internals/scalar.py:113-114).  out: DEVICE, M * N doubles.
 */
int ipde_modhelm_apply_columns_far(ipde_ctx* ctx, double k, int64_t ns, const double* sx, const double* sy,
                                   const double* w_sigma,
                                   const double* nx, const double* ny, const double* w_tau,
                                   int M, int64_t N, const double* tx, const double* ty, double* out);

/*
 * Stokes (mu=1) Stokeslet + stresslet with pressure:
 *   SLP: u = (1/4pi) sum [ -log r f + (d.f) d / r^2 ] ,  p = (1/2pi) sum (d.f)/r^2
 *   DLP: u = (1/pi)  sum (d.n)(d.g) d / r^4 ,
 *        p = (1/pi)  sum [ -(n.g)/r^2 + 2 (d.n)(d.g)/r^4 ]
 * with f=(wfx,wfy) and g=(wdx,wdy) weight-multiplied densities.
 * Replaces StokesHelper._define_layer_apply (pyfmmlib2d SFMM with
 * compute_target_stress=True; ipde/solvers/internals/stokes.py:25-35); pressure
 * formulas restated in ipde/solvers/internals/stokes_save.py:29-81.
 * wfx/wfy NULL => no SLP; wdx/wdy NULL => no DLP; out_p may be NULL.
 */
int ipde_stokes_apply(ipde_ctx* ctx, int loc,
                      int64_t ns, const double* sx, const double* sy,
                      const double* wfx, const double* wfy,
                      const double* nx, const double* ny,
                      const double* wdx, const double* wdy,
                      int64_t nt, const double* tx, const double* ty,
                      double* out_u, double* out_v, double* out_p, int flags);

/*
 * The Stokeslet sums (wfx, wfy; with pressure when out_p != NULL) onto a target list handed over
 * as 4 x 4 patches whose groups of 64 are 8 x 8 blocks of tiles (ipde_target_plan_build_blocks with
 * pad_blocks = 1; layout as in ipde_laplace_apply_patches), every block's sources beyond four block
 * radii through local expansions (three families of 27 complex coefficients: log|d|, d/conj(d) and
 * 1/d about the block's centre, 26 terms at ratio <= 1/4), the nearer batches of eight sources pair
 * by pair.  The place of the reference's FMM call for this sum (pyfmmlib2d SFMM,
 * ipde/solvers/internals/stokes.py:25-35).  The stresslet (nx, ny, wdx, wdy; reference
 * ipde/solvers/internals/stokes_save.py:41-54,77-81) takes the same route: with A = (n_x + i n_y)(g_x + i g_y),
 * U = sum [A / delta + 2 (n.g) / conj(delta) + conj(A) delta / conj(delta)^2] / 4, p / 2 = Re sum A / (2 delta^2),
 * three more coefficient families feeding the SAME three chains per target; either density pair may be NULL.
 * DEVICE pointers; values agree with ipde_stokes_apply to a few roundings of the largest partial sum.
 */
int ipde_stokes_apply_patches_far(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                  const double* wfx, const double* wfy,
                                  const double* nx, const double* ny, const double* wdx, const double* wdy,
                                  int64_t np, const double* pxy, const int32_t* pout,
                                  double* out_u, double* out_v, double* out_p);

/*
 * The same far-field form for the radial grid of an annulus: targets (M, N) row-major (tx, ty: DEVICE, M * N
 * doubles each), column j = the M points of one radial line, neighbouring columns neighbouring lines.
 * Blocks of 64 columns, a block's far sources in its local expansions, near batches pair by pair: the radial
 * sums of the Stokes helpers' correct() (ipde/solvers/internals/vector.py:140-162).  out_*: DEVICE, M * N
 * doubles; out_p may be NULL.
 */
int ipde_stokes_apply_columns_far(ipde_ctx* ctx, int64_t ns, const double* sx, const double* sy,
                                  const double* wfx, const double* wfy,
                                  const double* nx, const double* ny, const double* wdx, const double* wdy,
                                  int M, int64_t N, const double* tx, const double* ty,
                                  double* out_u, double* out_v, double* out_p);

/* ------------------------------------------------------------------------- */
/* periodic spectral grid operators                  */

/*
 * Plan for an (nx, ny) C-ordered real grid with periods Lx = nx*hx, Ly = ny*hy.
 * Wavenumbers follow ipde/ebdy_collection.py:388-395:
 *   kx = fftfreq(nx, hx/2pi) (shape (nx,1)), ky = fftfreq(ny, hy/2pi).
 */
int ipde_fft_plan2d_create(ipde_ctx* ctx, int64_t nx, int64_t ny,
                           double hx, double hy, ipde_fft_plan** plan);
int ipde_fft_plan2d_destroy(ipde_fft_plan* plan);

/* numpy-compatible complex transforms: ipde.utilities.fft2/ifft2
   (ipde/utilities.py:5-12).  direction: -1 forward, +1 inverse (scaled 1/(nx ny)). */
int ipde_fft2_c2c(ipde_fft_plan* plan, int loc, int direction,
                  const double* in_c, double* out_c);
/* fft2 of a real grid returning the full (nx,ny) complex spectrum. */
int ipde_fft2_r2c_full(ipde_fft_plan* plan, int loc, const double* in_r, double* out_c);

/*
 * PoissonSolver._grid_solve (ipde/solvers/multi_boundary/poisson.py:30-38)
 * without the demean step (caller does ebdyc.demean_function):
 *   uhat = fft2(f) * ilap, ilap = 1/(-kx^2-ky^2), ilap[0,0] = 0;  u = ifft2(uhat).real
 * uhat (full (nx,ny) complex) may be NULL.
 */
int ipde_poisson_grid_solve(ipde_fft_plan* plan, int loc,
                            const double* f, double* u, double* uhat);
/* ModifiedHelmholtzSolver._grid_solve
   (ipde/solvers/multi_boundary/modified_helmholtz.py:40-46): ihelm = 1/(k^2+kx^2+ky^2) */
int ipde_modhelm_grid_solve(ipde_fft_plan* plan, int loc, double k,
                            const double* f, double* u, double* uhat);
/* StokesSolver._grid_solve (ipde/solvers/multi_boundary/stokes.py:34-50) */
int ipde_stokes_grid_solve(ipde_fft_plan* plan, int loc,
                           const double* fu, const double* fv,
                           double* u, double* v, double* p);
/*
 * ipde.derivatives.fourier(f, ik) (ipde/derivatives.py:25-28) for ik = 1j*kx
 * (axis 0) or 1j*ky (axis 1): out = ifft2(fft2(f)*ik).real
 */
int ipde_fourier_deriv(ipde_fft_plan* plan, int loc, const double* f, int axis, double* out);

/* Grid -> scattered points (the step after the grid solve in ScalarSolver.__call__, reference
 * ipde/solvers/multi_boundary/scalar.py:80-88: values and gradient of the grid solution on
 * all interface nodes; three finufft type-2 transforms there).
 * ipde_fft_plan2d_keep_spectrum(plan, 1, &ok): subsequent ipde_poisson_grid_solve /
 * ipde_modhelm_grid_solve calls WITHOUT a uhat output keep fft2(f) * symbol on the device
 * (ok = 0: this grid size has no such path — every size up to 2048 x 4096 and the power-of-two
 * sizes up to 4096 x 8192 do —, use uhat and the caller's own evaluation).
 * ipde_grid_interp: out3 (3, np) row-major = u, du/dx, du/dy at the points (x, y), given in
 * box units [0, 2 pi) as ebdyc.interfaces_x_transf / interfaces_y_transf; derivatives in
 * physical units.  Oversampled (2x .. 4x: the fine grid is a power of two) inverse transform +
 * 16 x 16 window gather, ~1e-14. */
int ipde_fft_plan2d_keep_spectrum(ipde_fft_plan* plan, int on, int* supported);
/* Optional: build the interpolation state of this plan now (else at the first call below). */
int ipde_grid_interp_prepare(ipde_fft_plan* plan);
int ipde_grid_interp(ipde_fft_plan* plan, int loc, int64_t np, const double* x, const double* y,
                     double* out3);
/* The same for linear combinations of real grid fields and their first derivatives — the Stokes
 * solver's velocity and stress T = grad u + grad u^T - p I of the grid solution on the
 * interfaces (ipde/solvers/multi_boundary/vector.py:66-82, five type-2 NUFFTs there).
 * fields: nin <= 3 pointers to (nx, ny) real arrays; output k (k < nout <= 8) is the sum over
 * its terms q = term_start[k] .. term_start[k+1]-1 (1 to 3 of them) of
 * term_coef[q] * D^{term_der[q]} fields[term_src[q]], der 0: value, 1: d/dx, 2: d/dy.
 * out (nout, np).  Grid sizes as for ipde_grid_interp (IPDE_ERR_INVALID otherwise). */
int ipde_grid_interp_fields(ipde_fft_plan* plan, int loc, int nin, const double* const* fields,
                            int nout, const int* term_start, const int* term_src,
                            const int* term_der, const double* term_coef, int64_t np,
                            const double* x, const double* y, double* out);

/* general symbol: out = ifft2(fft2(f) * sym).real with sym a full (nx,ny)
   complex array (the reference accepts any broadcastable ik) */
int ipde_fourier_multiply(ipde_fft_plan* plan, int loc, const double* f,
                          const double* sym_c, double* out);
/* ipde.derivatives.fd_x_4 / fd_y_4 (ipde/derivatives.py:3-23): axis 0 = x.
   Rows/cols within 2 of the edge are zero unless periodic_fix. */
int ipde_fd4(ipde_ctx* ctx, int loc, int64_t nx, int64_t ny, double h,
             int axis, int periodic_fix, const double* f, double* out);

/* ------------------------------------------------------------------------- */
/* dense LU substitution (QFS collocation systems)         */
/*
 * x = U^-1 L^-1 P b for packed LU factors (unit lower L below the diagonal, as LAPACK
 * getrf / torch.linalg.lu_factor return them) and the row permutation perm (int32,
 * (P b)[i] = b[perm[i]]).  `lu` is TILED: nb = 2 ceil(n/128), the matrix padded with the
 * identity to (64 nb)^2 and stored as nb x nb contiguous 64x64 tiles, tile (I, K) at
 * ((I nb + K) * 4096) doubles, COLUMN-major inside a tile (element (r, c) at c*64 + r).  Replaces the host `lu_solve` of the reference's
 * third-party qfs package.  All pointers are DEVICE memory; b and x may not alias.
 */
int ipde_dense_lu_solve(ipde_ctx* ctx, int64_t n, const double* lu, const int* perm,
                        const double* b, double* x);
/* The same for nsys <= 8 independent systems advanced in lock-step (the substitution is latency
 * bound: a batch costs the steps of its largest member — the two QFS systems of an interface,
 * or those of all boundaries of a multiply connected domain, in one sequence of launches).
 * n: HOST array of nsys sizes; lu, perm, b, x: HOST arrays of nsys DEVICE pointers. */
int ipde_dense_lu_solve_batch(ipde_ctx* ctx, int nsys, const int64_t* n, const double* const* lu,
                              const int* const* perm, const double* const* b, double* const* x);

/*
 * LU factorisation with partial pivoting (LAPACK dgetrf's pivot rule), in place on the TILED storage
 * ipde_dense_lu_solve reads: what the reference's third-party `qfs` package obtains from host
 * LAPACK for every interface (qfs.two_d_qfs.QFS_Evaluator -> scipy.linalg.lu_factor; call sites
 * ipde/solvers/internals/poisson.py:18-25, stokes.py:21-24, examples/interior_poisson.py:87).
 * tiles: DEVICE, (n_pad/64)^2 tiles of 64 x 64 doubles, tile (I, J) at ((I nb + J) 4096), element
 * (r, c) of a tile at c*64 + r, the matrix padded with the identity to n_pad rows (a multiple of
 * 128, at most 32768).  perm: DEVICE, n_pad ints: row i of P A is row perm[i] of A.
 * Up to 8192 rows a panel is one workgroup's; beyond, the panels of more than 4096 rows are dealt out to
 * up to 64 workgroups (a row per thread, one cross-workgroup exchange per column; a hand-off that times
 * out sets the context's sticky abort word, reported by the next ipde_ctx_sync).
 * Asynchronous on the context's stream.
 */
int ipde_dense_lu_factor(ipde_ctx* ctx, int64_t n_pad, double* tiles, int* perm);

/*
 * y = A x (accumulate = 0) or y += A x (1); A: DEVICE, row-major (m, n); x (n), y (m): DEVICE.  The
 * matrix-vector products around the QFS solves — the one-sided boundary limit S sigma + D tau of
 * qfs.two_d_qfs.QFS_Evaluator.__call__ (call sites ipde/solvers/internals/scalar.py:87-88,
 * vector.py:133-134) and the residual of the refinement step.  Asynchronous on the context's stream.
 */
int ipde_dense_gemv(ipde_ctx* ctx, int64_t m, int64_t n, const double* A, const double* x, double* y,
                    int accumulate);
/* r = b - A x (A: m x n row-major, DEVICE) with products and sum carried in double-double: the residual of an
   iterative-refinement step of the QFS density solves (systems of condition ~1e15 and densities of 10^3 .. 10^4:
   the plain fp64 residual's own rounding is as large as the residual).  HBM bound: the matrix once. */
int ipde_dense_residual(ipde_ctx* ctx, int64_t m, int64_t n, const double* A, const double* x, const double* b,
                        double* r);
/* Noise cut of a periodic two-component QFS source density mu = [x(0..n-1) | y(0..n-1)] (DEVICE; out may be mu):
   the density's spectrum decays, reaches a minimum where the amplified noise of the boundary data takes over
   (collocation singular values ~ e^{-|k| alpha h} / |k|) and rises again; band maxima B_j of max(|z_k|, |z_-k|),
   z = x + i y, running minimum m_j: the first band j below 0.9 Nyquist whose whole tail up to 0.9 Nyquist lies above
   rise * m_{j-1}, with m_{j-1} < floor_rel * max B, marks the turnaround and every mode above the band of the
   minimum is removed (no such band: nothing is).  max_keep: modes above max_keep x Nyquist are removed in any case
   (>= 1: none).  kcut (DEVICE int, may be NULL) receives the last mode kept.  Restates what the reference's absent
   third-party `qfs` package leaves to its caller; used by ipde_amd/qfs.py Stokes_QFS (reference call sites
   ipde/solvers/internals/vector.py:124-125).  pack, FFT, cut (one workgroup), inverse FFT, unpack: five launches. */
int ipde_density_noise_cut(ipde_ctx* ctx, int64_t n, const double* mu, double* out, double rise, double floor_rel,
                           double max_keep, int* kcut);

/* ------------------------------------------------------------------------- */
/* Closest-point coordinates of points near a closed curve  */
/*
 * For every point p = (px[i], py[i]) within ~1.5 width of the curve X(t): the local
 * coordinates (r, t) with p = X(t) + r n(t), n the outward unit normal (r < 0 inside a
 * counter-clockwise curve), by Newton's method from the start values t0[i] (parameter of
 * the nearest sample).  Replaces near_finder.gridpoints_near_curve_update as used by
 * ipde/embedded_boundary.py:185-214.
 *   curve_tab : DEVICE, 3*nf complex128 (x + i y): X, X', X'' at t_j = 2 pi j / nf (the
 *               curve trigonometrically upsampled, >= 16 samples per shortest wavelength)
 *   bary_w    : HOST, 12 barycentric weights of 12 equispaced nodes
 *   px, py, t0, r_out, t_out : DEVICE, npts doubles; t_out in [0, 2 pi)
 */
int ipde_curve_local_coordinates(ipde_ctx* ctx, int64_t nf, const double* curve_tab,
                                 const double* bary_w, int64_t npts, const double* px,
                                 const double* py, const double* t0, double width, double tol,
                                 int maxiter, double* r_out, double* t_out);

/*
 * Inside mask of a whole nx x ny grid (row-major) from the near band alone: the nband cells
 * (ix[i], iy[i]) take r[i] < 0; every other cell takes the state of the last band cell before it
 * in its row (0 before the first).  The band index pairs must lie inside the grid (checked by the
 * caller: they come from the grid's own index arrays).  Replaces the full-grid half of the
 * reference's point classification (ipde/ebdy_collection.py:330-372 over
 * ipde/embedded_boundary.py:185-214).  ix, iy, r, inside: DEVICE; inside: nx*ny bytes, 0/1.
 */
int ipde_grid_inside_scan(ipde_ctx* ctx, int64_t nx, int64_t ny, int64_t nband, const int64_t* ix,
                          const int64_t* iy, const double* r, uint8_t* inside);

/*
 * Radial -> grid interpolation, gather half (replaces the per-mode type-2 NUFFT of
 * ipde/embedded_boundary.py:419-443): out[i] = sum_m T_m(xi[i]) c_m(t[i]), the Chebyshev
 * coefficient rows c_m given oversampled on nf equispaced t (cf: DEVICE, M x nf doubles,
 * row-major, nf = 16 x the boundary's node count) and read with 16-point barycentric
 * Lagrange interpolation.  bary_w: HOST, 16 weights; xi, t, out: DEVICE, npts doubles.
 */
int ipde_chebfourier_gather(ipde_ctx* ctx, int64_t M, int64_t nf, const double* cf,
                            const double* bary_w, int64_t npts, const double* xi, const double* t,
                            double* out);

/*
 * Radial -> grid interpolation, whole: what ipde/embedded_boundary.py:419-443 (`interpolate_radial_to_points`:
 * Chebyshev analysis along r, FFT along t, one type-2 NUFFT per Chebyshev mode) does for ONE field
 * of one boundary, for `nfld` fields that share their targets (the (u, v, p) of a Stokes solve,
 * ipde/solvers/multi_boundary/stokes.py:104-110), in one call and without a host round trip:
 *     out[f][idx ? idx[i] : i] = sum_m T_m(xi[i]) c^f_m(t[i]),   i < npts.
 * fr: nfld x M x N doubles at `loc` — values on (M Chebyshev-Gauss nodes, lowest first) x
 * (N equispaced t); bary_w: HOST, 16 barycentric weights; xi, t: DEVICE, npts doubles;
 * idx: DEVICE, npts int64 scatter positions, or NULL; out: HOST array of nfld DEVICE pointers.
 * Asynchronous on the context's stream.  nfld <= 8, M <= 512, N >= 16.
 */
int ipde_radial_to_grid(ipde_ctx* ctx, int loc, int64_t nfld, int64_t M, int64_t N, const double* fr,
                        const double* bary_w, int64_t npts, const double* xi, const double* t,
                        const int64_t* idx, double* const* out);

/*
 * Stokes helper algebra around the annular solve (ipde/solvers/internals/vector.py:65-144), device
 * arrays, asynchronous on the context's stream.  geom: DEVICE, 6 x n doubles — boundary normal x, y,
 * boundary tangent x, y, interface normal x, y.
 *   ipde_stokes_rotate           (u, v) -> (r, t) components (to_rt = 1; ipde/embedded_boundary.py:242-244)
 *                                or back (0; :246-248) of two (M, n) fields; a, b at `loc`, outputs DEVICE
 *   ipde_stokes_interface_jumps  from the annular solution (rr, tr, pr: (M, n)): its (u, v) components
 *                                (ur, vr), its traction on the interface (get_interface_traction_uvp:
 *                                radial derivative = D00 product, tangential derivative = batched FFT with
 *                                the wavenumbers rk, rs / irs = radial speed and its inverse (M, n), est =
 *                                interface estimator row (M)) and the jumps against the grid solution
 *                                bdata = (u, v, T_xx, T_xy, T_yy) on the interface (5 x n):
 *                                taus = sign [traction - T n], taud = sign [u, v], each 2 n doubles
 *                                (sign: +1 interior, -1 exterior boundary)
 */
/*   ipde_scalar_interface_jumps  the scalar counterpart (ipde/solvers/internals/scalar.py:76-90):
 *                                slp = sign [est . ur - (u_x n_x + u_y n_y)], dlp = sign u on the interface;
 *                                ur (M, n), est (M), nrm (2 x n: interface normal), bdata (3 x n: u, u_x, u_y)
 *                                of the grid solution); all DEVICE */
int ipde_scalar_interface_jumps(ipde_ctx* ctx, int M, int n, const double* ur, const double* est,
                                const double* nrm, const double* bdata, double sign, double* slp, double* dlp);
int ipde_stokes_rotate(ipde_ctx* ctx, int loc, int M, int n, const double* a, const double* b,
                       const double* geom, int to_rt, double* o1, double* o2);
int ipde_stokes_interface_jumps(ipde_ctx* ctx, int M, int n, const double* rr, const double* tr,
                                const double* pr, const double* geom, const double* rs, const double* irs,
                                const double* D00, const double* est, const double* rk, const double* bdata,
                                double sign, double* ur, double* vr, double* taus, double* taud);

/*
 * Grid <-> list moves of the multi-boundary solvers (ipde/embedded_function.py:105-113,135-138 and
 * ipde/solvers/multi_boundary/scalar.py:72-117 do them with numpy masks).  All arrays DEVICE; idx:
 * int64 positions in the flat grid, each at most once.
 *   ipde_grid_scatter   out = 0 (ngrid doubles); out[idx[i]] = src[i] * (scale ? scale[idx[i]] : 1)
 *                       — the forcing's physical values times the grid step function
 *   ipde_grid_add_at    out[idx[i]] += src[i]   — the dense sums' values onto grid_pna
 *   ipde_grid_gather    out[i] = in[idx[i]]     — the physical values of the answer
 */
int ipde_grid_scatter(ipde_ctx* ctx, int64_t ngrid, int64_t nidx, const int64_t* idx, const double* src,
                      const double* scale, double* out);
int ipde_grid_add_at(ipde_ctx* ctx, int64_t nidx, const int64_t* idx, const double* src, double* out);
int ipde_grid_gather(ipde_ctx* ctx, int64_t nidx, const int64_t* idx, const double* in, double* out);

/* ------------------------------------------------------------------------- */
/* Ewald-type grid evaluator, first half                       */
/*
 * ScalarGridBackend.ewald_local_freespace / ewald_local_periodic
 * (ipde/grid_evaluators/scalar_grid_evaluator.py:131-178,189-229): for every source
 * accumulate  q chi(r) G(r)  into `u_loc` and  q rho(r),
 * rho = 2 chi' G' + (chi'' + chi'/r) G  (= L[(1-chi) G]),  into `op` on the
 * (2 sw + 3)^2 grid points around it (r <= sw h).  kind 0: G = -log(r)/(2 pi)
 * (laplace_grid_evaluator.py:8-12); kind 1: G = K0(k r)/(2 pi)
 * (modified_helmholtz_grid_evaluator.py:8-9).  The second half of the evaluator,
 * the grid convolution of `op`, is ipde_fourier_multiply with the truncated-kernel
 * spectrum (scalar_grid_evaluator.py:283-307).
 * mol_tab (HOST): chi, d chi/dr, d2 chi/dr2 as [3][ni][deg+1] monomial coefficients in
 * the local variable t in [-1,1] of interval i of x = 1 - 2 r/(sw h) in [-1,1].
 */
typedef struct ipde_ewald ipde_ewald;
int ipde_ewald_create(ipde_ctx* ctx, int kind, double k, double h, int sw,
                      const double* mol_tab, int ni, int deg, ipde_ewald** out);
int ipde_ewald_destroy(ipde_ewald* e);
/* Grid point (ix, iy) = (x0 + ix h, y0 + iy h) is stored at
 * [(ix + offx) * nby + (iy + offy)]; periodic != 0 wraps the indices modulo (nbx, nby),
 * otherwise a stencil that leaves the array is an IPDE_ERR_INVALID.  `loc` describes
 * sx, sy, q; u_loc and op are DEVICE arrays (nbx, nby) that are accumulated into (the
 * caller zeroes them).  A grid point coinciding with a source is skipped. */
int ipde_ewald_spread(ipde_ewald* e, int loc, int64_t ns, const double* sx, const double* sy,
                      const double* q, double x0, double y0, int64_t nbx, int64_t nby,
                      int64_t offx, int64_t offy, int periodic, double* u_loc, double* op);

/* Stokeslet sum with pressure through the same split (kind-0 handle): the near part is
 * accumulated complete into loc3 = (u, v, p) planes, six Laplace densities
 * rho * {f_x, f_y, y_x f_x, y_x f_y, y_y f_x, y_y f_y} (y relative to (cx, cy)) into the
 * op6 planes; the caller finishes with
 *   u_i = loc_i + T*[op_i + d_j op_{2+2i+j}]/2 - (x_i - c_i) B/2,  p = loc_p - B,
 *   B = T*[d_x op_0 + d_y op_1]
 * (the Laplace reduction of the stokeslet; reference Layer_Apply of
 * ipde/solvers/internals/stokes.py:25-35 evaluated on a grid).  loc3 / op6: DEVICE
 * arrays of 3 / 6 contiguous (nbx, nby) planes, accumulated into. */
int ipde_ewald_spread_stokes(ipde_ewald* e, int loc, int64_t ns, const double* sx,
                             const double* sy, const double* fx, const double* fy, double x0,
                             double y0, double cx, double cy, int64_t nbx, int64_t nby,
                             int64_t offx, int64_t offy, double* loc3, double* op6);

/*
 * Set-up of the split evaluator: the truncated spectral Green's function of radius L on the
 * wavenumber quadrant kx (nx) x ky (ny), row-major into out (reference
 * ipde/grid_evaluators/laplace_grid_evaluator.py:21-33, modified_helmholtz_grid_evaluator.py:14-17;
 * helmholtz != 0: parameter kap with K0 = K0(L kap), K1 = K1(L kap) from the host).  J0 / J1 from
 * j01_tab: 2 x ni x (deg + 1) Chebyshev coefficients on pieces [i w, (i + 1) w] of the argument
 * (arguments beyond the table use its last piece: the caller sizes the table).  All arrays DEVICE.
 */
int ipde_trunc_sgf_quadrant(ipde_ctx* ctx, int64_t nx, int64_t ny, const double* kx, const double* ky,
                            double L, int helmholtz, double kap, double K0, double K1,
                            const double* j01_tab, int64_t ni, int deg, double piece_width, double* out);

/* batched 1-D complex FFT along the last axis of a (batch, n) array:
   ipde.utilities.fft / ifft (ipde/utilities.py:5-12). direction -1 / +1 (scaled). */
int ipde_fft1_c2c(ipde_ctx* ctx, int loc, int64_t batch, int64_t n, int direction,
                  const double* in_c, double* out_c);
/* create the plan of that transform ahead of time (rocFFT compiles its kernels at plan
   creation); thread safe, so a host warm-up thread may call it during set-up */
int ipde_fft1_prepare(ipde_ctx* ctx, int64_t batch, int64_t n);

/* ------------------------------------------------------------------------- */
/* annular solvers                                        */

/*
 * AnnularModifiedHelmholtzSolver / AnnularPoissonSolver
 * (ipde/annular/modified_helmholtz.py:90-203, ipde/annular/poisson.py:3-21) on the
 * `annular_full` geometry (ns == n; ipde/solvers/internals/scalar.py:2-3).
 * The small Chebyshev matrices and the per-mode inverse blocks are set-up data
 * computed by the Python host exactly as the reference computes them
 * (ChebyshevOperators, _construct) and handed over once:
 *   R01 (M-1,M)  R12 (M-2,M-1)  R02 (M-2,M)  D01 (M-1,M)  D12 (M-2,M-1)
 *   ibc, obc (M,)   = ia*ibc_dirichlet+ib*ibc_neumann, oa*obc_dirichlet+ob*obc_neumann
 *   kinv (n,M,M) real = Stacked_KINVS.real (modified_helmholtz.py:150-153)
 * all HOST pointers (set-up is host-side in the reference too).
 */
int ipde_annular_scalar_create(ipde_ctx* ctx, int M, int n, double helmholtz_k,
                               const double* R01, const double* R12, const double* R02,
                               const double* D01, const double* D12,
                               const double* ibc, const double* obc,
                               const double* kinv,
                               ipde_annular_scalar** h);
int ipde_annular_scalar_destroy(ipde_annular_scalar* h);
/* RealAnnularGeometry fields used by _apply (ipde/annular/annular_full.py:87-108):
   psi1, inv_psi1 (M-1,n), inv_psi2 (M-2,n); HOST or DEVICE per `loc`. */
int ipde_annular_scalar_set_geometry(ipde_annular_scalar* h, int loc,
                                     const double* psi1, const double* ipsi1,
                                     const double* ipsi2);
/* one operator application in Fourier space: out = _apply(uh)
   (modified_helmholtz.py:172-186); uh,out complex (M,n) row-major */
int ipde_annular_scalar_apply(ipde_annular_scalar* h, int loc,
                              const double* uh_c, double* out_c);
/* out = _optim_preconditioner(fh) (modified_helmholtz.py:157-159, :68-88) */
int ipde_annular_scalar_precondition(ipde_annular_scalar* h, int loc,
                                     const double* fh_c, double* out_c);
/*
 * solve (modified_helmholtz.py:187-203): f (M,n) real, ig, og (n,) real ->
 * out (M,n) real.  Right-preconditioned restarted GMRES on the device; stops when
 * ||r||/||b|| <= tol.  negate_f != 0 gives AnnularPoissonSolver.solve (f -> -f).
 */
int ipde_annular_scalar_solve(ipde_annular_scalar* h, int loc,
                              const double* f, const double* ig, const double* og,
                              int negate_f, double tol, int maxiter, int restart,
                              double* out, int* iters, double* resid);

/*
 * AnnularStokesSolver (ipde/annular/stokes.py:73-541) on the `annular` geometry
 * (ns == n-1, Nyquist mode dropped by mfft/mifft, ipde/utilities.py:78-99).
 * Set-up data (HOST): R01,R12,R02,D01,D12 as above, ibcd, obcd (M,), VI1row0 (M-1,)
 * = CO.VI1[0], kinv (ns,3M-1,3M-1) complex = Stacked_KINVS, mu.
 */
int ipde_annular_stokes_create(ipde_ctx* ctx, int M, int n, double mu,
                               const double* R01, const double* R12, const double* R02,
                               const double* D01, const double* D12,
                               const double* ibcd, const double* obcd,
                               const double* VI1row0,
                               const double* kinv_c,
                               ipde_annular_stokes** h);
int ipde_annular_stokes_destroy(ipde_annular_stokes* h);
/* RAG fields (ipde/annular/annular.py:87-108): psi0 (M,n), psi1, ipsi1 (M-1,n),
   ipsi2, DR_psi2, ipsi_DR_ipsi_DT_psi2, ipsi_DT_ipsi_DR_psi2 (M-2,n) */
int ipde_annular_stokes_set_geometry(ipde_annular_stokes* h, int loc,
                                     const double* psi0, const double* psi1,
                                     const double* ipsi1, const double* ipsi2,
                                     const double* DR_psi2,
                                     const double* ipsi_DR_ipsi_DT_psi2,
                                     const double* ipsi_DT_ipsi_DR_psi2);
/* out = _apply_optim_real(uuh) (stokes.py:321-385); vectors complex length
   NB = 2*M*ns + (M-1)*ns */
int ipde_annular_stokes_apply(ipde_annular_stokes* h, int loc,
                              const double* uuh_c, double* out_c);
/* out = _preconditioner(ffh) (stokes.py:200-210, :51-71) */
int ipde_annular_stokes_precondition(ipde_annular_stokes* h, int loc,
                                     const double* ffh_c, double* out_c);
/* solve (stokes.py:519-541): fr, ft (M,n); irg,itg,org,otg (n,) ->
   ur, ut (M,n), p (M,n) (= P10 . p_{M-1}); P10 (M,M-1) handed in (HOST). */
int ipde_annular_stokes_solve(ipde_annular_stokes* h, int loc,
                              const double* fr, const double* ft,
                              const double* irg, const double* itg,
                              const double* org, const double* otg,
                              const double* P10_host,
                              double tol, int maxiter, int restart,
                              double* ur, double* ut, double* p,
                              int* iters, double* resid);

#ifdef __cplusplus
}
#endif
#endif /* IPDE_HIP_H */
