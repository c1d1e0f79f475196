// Internal interface of the hand-written 2-D real FFT pipeline (fft2d.hip), used by
// spectral.hip for power-of-two grids.
#pragma once
#include "ipde_common.h"

enum { FFT2D_SYM_POISSON = 0, FFT2D_SYM_MODHELM = 1, FFT2D_SYM_DX = 2, FFT2D_SYM_DY = 3, FFT2D_SYM_NONE = 4 };

struct Fft2dPlan {
    bool ready = false;
    int64_t nx = 0, ny = 0, pitch = 0;   // pitch = ny/2: complex entries per PACKED half-spectrum row
    double hx = 0, hy = 0;
    void* tw_x = nullptr;    // exp(-2 pi i m / nx),      m < nx
    void* tw_h = nullptr;    // exp(-2 pi i m / (ny/2)),  m < ny/2
    void* tw_ny = nullptr;   // exp(-2 pi i m / ny),      m < ny/2
    void* W[3] = {nullptr, nullptr, nullptr};   // packed half spectra, (nx, pitch) complex, row-major;
                                                // Im W[x][0] holds the (real) Nyquist entry ky = ny/2
};

bool fft2d_supported(int64_t nx, int64_t ny);
int fft2d_plan_init(ipde_ctx* ctx, Fft2dPlan& p, int64_t nx, int64_t ny, double hx, double hy);
void fft2d_plan_free(Fft2dPlan& p);
// rows: real (nx, ny) -> half spectrum W[slot]; and back (out = c2r / 2)
int fft2d_rows_forward(ipde_ctx* ctx, const Fft2dPlan& p, const double* f, int slot);
int fft2d_rows_inverse(ipde_ctx* ctx, const Fft2dPlan& p, int slot, double* out);
// columns of W[slot] in place; mode 0: forward * symbol * scale, inverse; 1: forward; 2: inverse
// spec_slot >= 0 (mode 0): also store the post-symbol spectrum in W[spec_slot]; ncols > 0: only
// the leading ncols columns (the rest are zeros, e.g. a zero-padded spectrum)
int fft2d_cols(ipde_ctx* ctx, const Fft2dPlan& p, int slot, int sym, int mode, double k2h, double scale,
               int spec_slot = -1, int64_t ncols = 0);
int fft2d_scalar_solve(ipde_ctx* ctx, const Fft2dPlan& p, int sym, double k2h, const double* f, double* u,
                       bool keep_spectrum = false);
int fft2d_stokes_solve(ipde_ctx* ctx, const Fft2dPlan& p, const double* fu, const double* fv, double* u,
                       double* v, double* pr);

// grid -> scattered points through an oversampled inverse transform (nufft.hip)
struct GridInterp;
bool grid_interp_supported(int64_t nx, int64_t ny);
int grid_interp_create(ipde_ctx* ctx, int64_t nx, int64_t ny, double hx, double hy, GridInterp** out,
                       bool general = false);
bool grid_interp_general_supported(int64_t nx, int64_t ny);
int grid_interp_eval_general(GridInterp* gi, const void* spec, int loc, int64_t np, const double* px,
                             const double* py, double dkx, double dky, double* out);
int grid_interp_fields_general(GridInterp* gi, int nin, const void* const* specs, int nout,
                               const int* term_start, const int* term_src, const int* term_der,
                               const double* term_coef, int loc_points, int64_t np, const double* px,
                               const double* py, double dkx, double dky, double* out);
void grid_interp_destroy(GridInterp* gi);
// coarse: the grid's own plan; its W[1] holds the kept spectrum, W[2] is scratch
int grid_interp_eval(GridInterp* gi, const Fft2dPlan& coarse, int loc, int64_t np, const double* px,
                     const double* py, double dkx, double dky, double* out);
int grid_interp_fields(GridInterp* gi, const Fft2dPlan& coarse, int nin, const double* const* d_fields,
                       int nout, const int* term_start, const int* term_src, const int* term_der,
                       const double* term_coef, int loc_points, int64_t np, const double* px,
                       const double* py, double dkx, double dky, double* out);
// test hook: force the shifted-copies variant (four coarse transforms) on the next create
void grid_interp_force_shifted(bool on);
