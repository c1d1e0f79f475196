"""Own minimal stand-ins for the part of the pybie2d surface that ipde's scripts and
solvers touch.  pybie2d is not part of the reference tree; these
classes follow the ATTRIBUTE CONTRACT the reference reads
(`.x .y .c .N .t .dt .speed .curvature .normal_x .normal_y .tangent_x .tangent_y
.weights .get_stacked_boundary()`; reference ipde/embedded_boundary.py:280-358,
ipde/solvers/internals/scalar.py:31-35, ipde/ebdy_collection.py:22-31) with standard
spectral differentiation of a closed curve sampled at equispaced parameter values.
Host numpy: one-time geometry set-up.
"""
import numpy as np


def star(N, x=0.0, y=0.0, r=1.0, a=0.5, f=3, rot=0.0):
    """Complex points of the star curve  (x+iy) + r (1 + a cos(f (t - rot))) e^{it}
    (the call shape of pybie2d.misc.curve_descriptions.star; reference
    examples/interior_poisson.py:41 uses star(nb, a=0.2, f=5))."""
    t = np.linspace(0.0, 2 * np.pi, N, endpoint=False)
    return (x + 1j * y) + (r + r * a * np.cos(f * (t - rot))) * np.exp(1j * t)


def squish(N, x=0.0, y=0.0, r=1.0, b=1.0, rot=0.0):
    """Ellipse-like closed curve (semi-axes r, r*b)."""
    t = np.linspace(0.0, 2 * np.pi, N, endpoint=False)
    return (x + 1j * y) + np.exp(1j * rot) * (r * np.cos(t) + 1j * r * b * np.sin(t))


class PointSet(object):
    def __init__(self, x=None, y=None, c=None):
        if c is not None:
            c = np.asarray(c)
            x, y = c.real, c.imag
        self.x = np.ascontiguousarray(x, dtype=float).ravel()
        self.y = np.ascontiguousarray(y, dtype=float).ravel()
        self.c = self.x + 1j * self.y
        self.N = self.x.shape[0]

    def get_stacked_boundary(self, T=True):
        s = np.vstack([self.x, self.y])
        return s if T else s.T


class Global_Smooth_Boundary(PointSet):
    """Smooth closed curve, counter-clockwise, equispaced in its parameter t."""

    def __init__(self, x=None, y=None, c=None):
        super().__init__(x, y, c)
        N = self.N
        self.t, self.dt = np.linspace(0.0, 2 * np.pi, N, endpoint=False, retstep=True)
        self.k = np.fft.fftfreq(N, 1.0 / N)
        self.ik = 1j * self.k
        ch = np.fft.fft(self.c)
        self.cp = np.fft.ifft(ch * self.ik)
        self.cpp = np.fft.ifft(ch * self.ik ** 2)
        self.speed = np.abs(self.cp)
        self.tangent_c = self.cp / self.speed
        self.tangent_x, self.tangent_y = self.tangent_c.real.copy(), self.tangent_c.imag.copy()
        self.normal_c = -1j * self.tangent_c          # outward for a ccw curve
        self.normal_x, self.normal_y = self.normal_c.real.copy(), self.normal_c.imag.copy()
        self.curvature = (self.cp.real * self.cpp.imag - self.cp.imag * self.cpp.real) / self.speed ** 3
        self.weights = self.speed * self.dt
        self.max_h = np.max(self.weights)
        self.area = 0.5 * np.sum((self.x * self.normal_x + self.y * self.normal_y) * self.weights)

    # self-interaction forms as METHODS: the call shape of the pybie2d generation the
    # reference's single-boundary code was written against
    # (ipde/solvers/single_boundary/interior/modified_helmholtz.py:36,
    # examples/interior_modified_helmholtz.py:73,76).  That generation scaled the
    # modified-Helmholtz double layer by k^2 (the example pairs it with a jump of k^2/2;
    # SURVEY §9.13): `Modified_Helmholtz_DLP_Self_Form` keeps that scaling so the old
    # scripts stay consistent; everything else in this package is unscaled.
    def Laplace_SLP_Self_Form(self):
        return Laplace_Layer_Singular_Form(self, ifcharge=True)

    def Laplace_DLP_Self_Form(self):
        return Laplace_Layer_Singular_Form(self, ifdipole=True)

    def Modified_Helmholtz_SLP_Self_Form(self, k=1.0):
        return Modified_Helmholtz_Layer_Singular_Form(self, k=k, ifcharge=True)

    def Modified_Helmholtz_DLP_Self_Form(self, k=1.0):
        return k * k * Modified_Helmholtz_Layer_Singular_Form(self, k=k, ifdipole=True)

    def generate_resampled_boundary(self, new_N):
        return Global_Smooth_Boundary(c=fourier_resample(self.c, new_N))


GSB = Global_Smooth_Boundary


def fourier_resample(f, new_N):
    """Trigonometric resampling of a periodic sequence to new_N points."""
    N = f.shape[0]
    fh = np.fft.fft(f)
    out = np.zeros(new_N, dtype=complex)
    if new_N >= N:
        h = N // 2
        out[:h] = fh[:h]                          # k = 0 .. h-1
        if N % 2 == 0:
            out[h] = 0.5 * fh[h]                  # split the Nyquist mode
            out[new_N - h] += 0.5 * fh[h]
            if h > 1:
                out[new_N - h + 1:] = fh[h + 1:]  # k = -(h-1) .. -1
        else:
            out[h] = fh[h]
            out[new_N - h:] = fh[h + 1:]
    else:
        h = (new_N - 1) // 2                      # keep |k| <= h
        out[:h + 1] = fh[:h + 1]
        if h > 0:
            out[new_N - h:] = fh[N - h:]
    res = np.fft.ifft(out) * (new_N / N)
    return res if np.iscomplexobj(f) else res.real


def arc_length_parameterize(x, y, tol=1e-13):
    """Resample a closed curve at equal arclength (the role of
    personal_utilities.arc_length_reparametrization in the reference's scripts,
    examples/multi_stokes.py:40-43).  The arclength s(t) is the spectral antiderivative of
    the speed; s(t_j) = j L / N is solved by Newton and the curve's Fourier series is
    evaluated at the new parameters.  Returns (x, y)."""
    c = np.asarray(x, dtype=float) + 1j * np.asarray(y, dtype=float)
    N = c.shape[0]
    k = np.fft.fftfreq(N, 1.0 / N)
    ch = np.fft.fft(c) / N
    dk = 1j * k
    if N % 2 == 0:
        dk[N // 2] = 0.0
    speed = np.abs(np.fft.ifft(dk * ch * N))
    sh = np.fft.fft(speed) / N
    L = 2 * np.pi * sh[0].real
    ah = np.zeros_like(sh)
    nz = k != 0
    ah[nz] = sh[nz] / (1j * k[nz])
    if N % 2 == 0:
        ah[N // 2] = 0.0
    target = np.arange(N) * (L / N)
    t = np.arange(N) * (2 * np.pi / N)
    # trigonometric sums at the N current parameters: dense N x N exponentials, on the GPU
    # when there is one (N = 12 400 costs seconds per Newton step in numpy)
    try:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    except Exception:
        dev = None
    if dev is not None:
        kd = torch.as_tensor(k, device=dev)
        coef = torch.as_tensor(np.stack([ah, sh, ch], axis=1), device=dev)       # (N, 3)

        def sums(tt):
            ang = torch.as_tensor(tt, device=dev)[:, None] * kd[None, :]
            E = torch.complex(torch.cos(ang), torch.sin(ang))
            r = (E @ coef).cpu().numpy()
            return r[:, 0], r[:, 1], r[:, 2]
    else:
        def sums(tt):
            E = np.exp(1j * np.outer(tt, k))
            return E @ ah, E @ sh, E @ ch
    s0 = float(np.sum(ah).real)                  # periodic part of s at t = 0
    for _ in range(50):
        a, b, _c = sums(t)
        dtn = (sh[0].real * t + a.real - s0 - target) / b.real
        t = t - dtn
        nit = _ + 1
        if np.abs(dtn).max() < tol:
            break
    arc_length_parameterize.last_iterations = nit
    cn = sums(t)[2]
    return cn.real, cn.imag


class Grid(object):
    def __init__(self, x_bounds, Nx, y_bounds, Ny, mask=None, x_endpoints=(True, True),
                 y_endpoints=(True, True)):
        self.x_bounds, self.y_bounds = list(x_bounds), list(y_bounds)
        self.Nx, self.Ny = int(Nx), int(Ny)
        self.xv = self._axis(x_bounds, self.Nx, x_endpoints)
        self.yv = self._axis(y_bounds, self.Ny, y_endpoints)
        self.xh = self.xv[1] - self.xv[0]
        self.yh = self.yv[1] - self.yv[0]
        self.xg, self.yg = np.meshgrid(self.xv, self.yv, indexing='ij')
        self.shape = (self.Nx, self.Ny)
        self.N = self.Nx * self.Ny
        self.mask = mask
        self.x_endpoints, self.y_endpoints = list(x_endpoints), list(y_endpoints)

    @staticmethod
    def _axis(b, n, ends):
        if ends[0] and ends[1]:
            return np.linspace(b[0], b[1], n, endpoint=True)
        if ends[0]:
            return np.linspace(b[0], b[1], n, endpoint=False)
        v = np.linspace(b[0], b[1], n + 1, endpoint=ends[1])
        return v[1:]


class BoundaryCollection(object):
    """Concatenation of source curves (reference multi_boundary/scalar.py:34-39)."""

    def __init__(self):
        self.boundaries = []
        self.sides = []

    def add(self, bdy, side):
        self.boundaries.append(bdy)
        self.sides.append(side)

    def amass_information(self):
        cat = lambda name: np.concatenate([np.asarray(getattr(b, name)) for b in self.boundaries])
        self.x, self.y = cat('x'), cat('y')
        self.c = self.x + 1j * self.y
        self.weights = cat('weights')
        self.normal_x, self.normal_y = cat('normal_x'), cat('normal_y')
        self.N = self.x.shape[0]
        self.Ns = [b.N for b in self.boundaries]

    def get_stacked_boundary(self, T=True):
        s = np.vstack([self.x, self.y])
        return s if T else s.T


# ---------------------------------------------------------------------------
# dense layer matrices (host; set-up of QFS and of the example-level boundary solve)
def Laplace_Layer_Form(source, target=None, ifcharge=False, ifdipole=False):
    """Dense matrix of the Laplace SLP and/or DLP from `source` to `target`
    (off-surface; weights included).  Conventions of ipde_amd.layer_potentials."""
    if target is None:
        target = source
    dx = target.x[:, None] - source.x[None, :]
    dy = target.y[:, None] - source.y[None, :]
    d2 = dx * dx + dy * dy
    out = np.zeros_like(d2)
    if ifcharge:
        out += (-0.25 / np.pi) * np.log(d2) * source.weights[None, :]
    if ifdipole:
        out += (0.5 / np.pi) * (dx * source.normal_x[None, :] + dy * source.normal_y[None, :]) / d2 \
            * source.weights[None, :]
    return out


_kress_cache = {}


def _kress_log_weights(N):
    """R_j with  int_0^{2pi} log(4 sin^2((t_i-s)/2)) f(s) ds ~ sum_j R_{|i-j|} f(s_j)
    for trigonometric f (Kress' quadrature for the periodic log singularity):
    R_j = -(4 pi/N) [ sum_{m=1}^{M} cos(m t_j)/m  (+ cos(N t_j/2)/N for even N) ],
    M = N/2 - 1 (even N) or (N-1)/2 (odd N: no Nyquist mode).  The cosine sum is the real
    part of a length-N FFT of the coefficients 1/m; kept per N."""
    R = _kress_cache.get(N)
    if R is None:
        j = np.arange(N)
        M = N // 2 - 1 if N % 2 == 0 else (N - 1) // 2
        a = np.zeros(N)
        a[1:M + 1] = 1.0 / np.arange(1, M + 1)
        extra = np.cos(np.pi * j) / N if N % 2 == 0 else 0.0
        R = -(4 * np.pi / N) * (np.fft.fft(a).real + extra)
        _kress_cache[N] = R
    return R


def Laplace_Layer_Singular_Form(bdy, ifcharge=False, ifdipole=False):
    """On-surface Nystrom matrices of the Laplace SLP (Kress log split) and DLP (smooth
    kernel, diagonal -curvature/(4 pi) * weight) on a Global_Smooth_Boundary.  The DLP
    here is the principal value; interior limit = D - I/2 (reference
    examples/interior_poisson.py:19)."""
    N = bdy.N
    out = np.zeros((N, N))
    dx = bdy.x[:, None] - bdy.x[None, :]
    dy = bdy.y[:, None] - bdy.y[None, :]
    d2 = dx * dx + dy * dy
    if ifcharge:
        dt = bdy.t[:, None] - bdy.t[None, :]
        s2 = 4 * np.sin(dt / 2) ** 2
        np.fill_diagonal(d2, 1.0)
        np.fill_diagonal(s2, 1.0)
        smooth = -0.25 / np.pi * np.log(d2 / s2)
        np.fill_diagonal(smooth, -0.5 / np.pi * np.log(bdy.speed))
        R = _kress_log_weights(N)
        idx = np.abs(np.arange(N)[:, None] - np.arange(N)[None, :])
        out += (-0.25 / np.pi) * R[idx] * bdy.speed[None, :] + smooth * bdy.weights[None, :]
        np.fill_diagonal(d2, 0.0)
    if ifdipole:
        with np.errstate(divide='ignore', invalid='ignore'):
            D = (0.5 / np.pi) * (dx * bdy.normal_x[None, :] + dy * bdy.normal_y[None, :]) / d2
        np.fill_diagonal(D, -bdy.curvature / (4 * np.pi))
        out += D * bdy.weights[None, :]
    return out


# ---------------------------------------------------------------------------
# modified Helmholtz (k^2 - Lap): kernels of ipde_amd.layer_potentials
def _row_blocks(fn, nrows, out):
    """out[a:b] = fn(a, b) over row blocks in a thread pool (numpy / scipy.special ufuncs
    release the GIL; the 8192^2 Bessel evaluations of one modified-Helmholtz form take
    ~15 s on one core)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(16, os.cpu_count() or 1))
    step = max(64, -(-nrows // (4 * nthreads)))
    blocks = [(a, min(nrows, a + step)) for a in range(0, nrows, step)]

    def work(ab):
        out[ab[0]:ab[1]] = fn(*ab)
    if nthreads == 1 or len(blocks) == 1:
        for ab in blocks:
            work(ab)
    else:
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(work, blocks))
    return out


def Modified_Helmholtz_Layer_Form(source, target=None, k=1.0, ifcharge=False, ifdipole=False):
    """Dense off-surface matrix: (1/2pi) K0(k r) w  and/or  (k/2pi) K1(k r) (n.d)/r w."""
    from scipy.special import k0, k1
    if target is None:
        target = source

    def rows(a, b):
        dx = target.x[a:b, None] - source.x[None, :]
        dy = target.y[a:b, None] - source.y[None, :]
        r = np.hypot(dx, dy)
        blk = np.zeros_like(r)
        if ifcharge:
            blk += (0.5 / np.pi) * k0(k * r) * source.weights[None, :]
        if ifdipole:
            nd = dx * source.normal_x[None, :] + dy * source.normal_y[None, :]
            blk += (0.5 * k / np.pi) * k1(k * r) * nd / r * source.weights[None, :]
        return blk
    return _row_blocks(rows, target.x.shape[0], np.empty((target.x.shape[0], source.x.shape[0])))


def Modified_Helmholtz_Layer_Singular_Form(bdy, k=1.0, ifcharge=False, ifdipole=False):
    """On-surface Nystrom matrices with a LOCALISED Kress logarithmic split:
       K0(k r)/(2 pi)                = -(1/4pi) I0(k r) psi(r) L + remainder,
       (k/2pi) K1(k r) (n.d)/r       =  (k/4pi) I1(k r) psi(r) (n.d)/r L + remainder,
    L = log(4 sin^2((t-s)/2)).  The analytic log-coefficients I0, I1 grow like e^{k r};
    used globally they make the split cancel catastrophically once k*diameter >~ 20
    (error ~ eps I0(k diam)).  psi is a C-infinity cut-off (Slepian step) equal to 1 for
    k r <= 2 and 0 beyond k r = 6 (at least 24 nodes wide), so the remainder is still
    smooth, the coefficient stays below I0(6) ~ 67, and beyond the cut-off the plain
    trapezoid rule acts on the smooth kernel.  Diagonal limits:
    -(log(k speed/2) + gamma)/(2 pi) for the SLP remainder, -curvature/(4 pi) for the DLP
    remainder.  The DLP is the principal value (interior limit D - I/2).
    Row blocks are built in a thread pool; the cut-off and I0/I1 are evaluated only
    where psi is neither 0 nor 1 / not 0."""
    from scipy.special import k0, k1, i0, i1
    from .heavisides import SlepianMollifier
    N = bdy.N
    j = np.arange(N)
    lt = 4 * np.sin(0.5 * bdy.dt * j) ** 2            # L and R depend on |i - j| only
    lt[0] = 1.0
    lt = np.log(lt)
    Rt = _kress_log_weights(N)
    r1 = 2.0 / k
    r2 = max(6.0 / k, r1 + 24 * bdy.max_h)
    mol = SlepianMollifier(30)

    def rows(a, b):
        ii = np.arange(a, b)
        dx = bdy.x[a:b, None] - bdy.x[None, :]
        dy = bdy.y[a:b, None] - bdy.y[None, :]
        r = np.hypot(dx, dy)
        diag = (ii - a, ii)
        r[diag] = 1.0
        sep = np.abs(ii[:, None] - j[None, :])
        L, R = lt[sep], Rt[sep]
        psi = (r <= r1).astype(float)
        band = (r > r1) & (r < r2)
        psi[band] = 1.0 - mol.step(2.0 * (r[band] - r1) / (r2 - r1) - 1.0)
        psi[diag] = 1.0
        near = psi > 0.0
        blk = np.zeros((b - a, N))
        if ifcharge:
            S1 = np.zeros_like(r)
            S1[near] = -(0.25 / np.pi) * i0(k * r[near]) * psi[near]
            S2 = (0.5 / np.pi) * k0(k * r) - S1 * L
            S1[diag] = -(0.25 / np.pi)
            S2[diag] = -(0.5 / np.pi) * (np.log(0.5 * k * bdy.speed[a:b]) + np.euler_gamma)
            blk += (S1 * R + S2 * bdy.dt) * bdy.speed[None, :]
        if ifdipole:
            nd = dx * bdy.normal_x[None, :] + dy * bdy.normal_y[None, :]
            D1 = np.zeros_like(r)
            D1[near] = (0.25 * k / np.pi) * i1(k * r[near]) * psi[near] * nd[near] / r[near]
            D2 = (0.5 * k / np.pi) * k1(k * r) * nd / r - D1 * L
            D1[diag] = 0.0
            D2[diag] = -bdy.curvature[a:b] / (4 * np.pi)
            blk += (D1 * R + D2 * bdy.dt) * bdy.speed[None, :]
        return blk
    return _row_blocks(rows, N, np.empty((N, N)))


# ---------------------------------------------------------------------------
# Stokes (mu = 1): kernels of ipde_amd.layer_potentials.stokes_apply.  Vector densities are
# stacked [x-components; y-components]; the matrices are 2x2 blocks [[xx, xy], [yx, yy]].
def _stack_blocks(Bxx, Bxy, Byy):
    return np.block([[Bxx, Bxy], [Bxy, Byy]])


def Stokes_Layer_Form(source, target=None, ifforce=False, ifdipole=False):
    """Dense off-surface matrix (2 Nt x 2 Ns; weights included) of
       stokeslet  (1/4pi) [ -log r  I + d d^T / r^2 ]      and / or
       stresslet  (1/pi) (d.n) d d^T / r^4,     d = target - source, n = source normal
    (pybie2d.kernels.high_level.stokes call shape; reference examples/multi_stokes.py:131-133)."""
    if target is None:
        target = source
    dx = target.x[:, None] - source.x[None, :]
    dy = target.y[:, None] - source.y[None, :]
    d2 = dx * dx + dy * dy
    id2 = 1.0 / d2
    w = source.weights[None, :]
    Bxx = np.zeros_like(d2)
    Bxy = np.zeros_like(d2)
    Byy = np.zeros_like(d2)
    if ifforce:
        c = 0.25 / np.pi
        lg = -0.5 * np.log(d2)
        Bxx += c * (lg + dx * dx * id2) * w
        Bxy += c * (dx * dy * id2) * w
        Byy += c * (lg + dy * dy * id2) * w
    if ifdipole:
        q = (dx * source.normal_x[None, :] + dy * source.normal_y[None, :]) * id2 * id2 * w / np.pi
        Bxx += q * dx * dx
        Bxy += q * dx * dy
        Byy += q * dy * dy
    return _stack_blocks(Bxx, Bxy, Byy)


def Stokes_Layer_Singular_Form(bdy, ifforce=False, ifdipole=False):
    """On-surface Nystrom matrices (2N x 2N).  Stokeslet: the -log r part is half the
    Laplace single layer (Kress split), d d^T / r^2 is smooth with diagonal limit t t^T.
    Stresslet: smooth, diagonal limit -curvature/(2 pi) t t^T; principal value, the
    interior limit is D - I/2 (reference examples/multi_stokes.py:134-138)."""
    N = bdy.N
    dx = bdy.x[:, None] - bdy.x[None, :]
    dy = bdy.y[:, None] - bdy.y[None, :]
    d2 = dx * dx + dy * dy
    np.fill_diagonal(d2, 1.0)
    id2 = 1.0 / d2
    w = bdy.weights[None, :]
    txx = bdy.tangent_x * bdy.tangent_x
    txy = bdy.tangent_x * bdy.tangent_y
    tyy = bdy.tangent_y * bdy.tangent_y
    Bxx = np.zeros((N, N))
    Bxy = np.zeros((N, N))
    Byy = np.zeros((N, N))
    if ifforce:
        c = 0.25 / np.pi
        half_slp = 0.5 * Laplace_Layer_Singular_Form(bdy, ifcharge=True)
        Rxx, Rxy, Ryy = dx * dx * id2, dx * dy * id2, dy * dy * id2
        np.fill_diagonal(Rxx, txx)
        np.fill_diagonal(Rxy, txy)
        np.fill_diagonal(Ryy, tyy)
        Bxx += half_slp + c * Rxx * w
        Bxy += c * Rxy * w
        Byy += half_slp + c * Ryy * w
    if ifdipole:
        q = (dx * bdy.normal_x[None, :] + dy * bdy.normal_y[None, :]) * id2 * id2 / np.pi
        Dxx, Dxy, Dyy = q * dx * dx, q * dx * dy, q * dy * dy
        lim = -bdy.curvature / (2 * np.pi)
        np.fill_diagonal(Dxx, lim * txx)
        np.fill_diagonal(Dxy, lim * txy)
        np.fill_diagonal(Dyy, lim * tyy)
        Bxx += Dxx * w
        Bxy += Dxy * w
        Byy += Dyy * w
    return _stack_blocks(Bxx, Bxy, Byy)


def Stokes_Pressure_Fix(source, target):
    """n_trg (x) n_src w_src / |source curve| — the rank-one completion the reference's
    scripts add to the interior double-layer block (examples/multi_stokes.py:122-128)."""
    nt = np.concatenate([target.normal_x, target.normal_y])
    ns = np.concatenate([source.normal_x * source.weights, source.normal_y * source.weights])
    return np.outer(nt, ns) / np.sum(source.weights)
