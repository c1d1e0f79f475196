"""EmbeddedBoundary — the attribute contract of ipde/embedded_boundary.py (reference
:55-557) that the solvers and the example scripts read, rebuilt on our own geometry,
QFS and interpolation modules.  A smooth closed boundary, its interface curve at
distance M*h along the normal, the boundary-fitted (Chebyshev-Gauss x periodic)
radial grid between them, the classification of background-grid points and the
cut-off functions coupling the two grids.  Host numpy (one-time set-up); the
radial -> grid interpolation evaluates on the device (ipde_amd.interp).
"""
import numpy as np

from .device import prewarm
from .heavisides import SlepianMollifier
from .near import local_coordinates, band_mask
from .pybie2d_compat import Global_Smooth_Boundary as GSB, PointSet
from .qfs import QFS_Boundary
from .utilities import affine_transformation, get_chebyshev_nodes


def setit(n, dictionary, default):
    return dictionary[n] if n in dictionary else default


def LoadEmbeddedBoundary(d):
    """EmbeddedBoundary from the dictionary its `save` method made (reference
    ipde/embedded_boundary.py:38-43).  The heaviside function is not stored (`save` drops it, as the
    reference's does): the loaded boundary gets the default one unless d['kwargs'] is given one."""
    bdy = GSB(x=np.array(d['bx'], dtype=float), y=np.array(d['by'], dtype=float))
    return EmbeddedBoundary(bdy, d['interior'], d['M'], d['h'], **d['kwargs'])


class EmbeddedBoundary(object):
    def __init__(self, bdy, interior, M, h, *legacy, **kwargs):
        # library loads and the length-N 1-D FFT kernels (annular solver, radial
        # interpolation) compile while the host does the geometry set-up
        prewarm(fft1=((M, bdy.N), (16 * M, bdy.N)))
        # old call form (reference examples/interior_modified_helmholtz.py:41):
        # EmbeddedBoundary(bdy, interior, M, h, pad_zone, heaviside)
        if len(legacy) > 2:
            raise TypeError("EmbeddedBoundary(bdy, interior, M, h[, pad_zone[, heaviside]], **kwargs)")
        for name, val in zip(('pad_zone', 'heaviside'), legacy):
            kwargs.setdefault(name, val)
        self._solo = None
        """bdy: Global_Smooth_Boundary; interior: bool; M: radial modes; h: radial grid
        spacing (radial_width = M*h).  kwargs as the reference (:106-112): pad_zone,
        heaviside, qfs_tolerance, coordinate_tolerance, ... (unknown ones are kept)."""
        self.bdy = bdy
        self.interior = interior
        self.M = M
        self.h = h
        self.pad_zone = setit('pad_zone', kwargs, 0)
        self.coordinate_tolerance = setit('coordinate_tolerance', kwargs, 1e-14)
        self.qfs_tolerance = setit('qfs_tolerance', kwargs, 1e-12)
        self.qfs_fsuf = setit('qfs_fsuf', kwargs, None)
        self.qfs_FF = setit('qfs_FF', kwargs, 0.0)
        self.heaviside = kwargs['heaviside'] if 'heaviside' in kwargs \
            else SlepianMollifier(2 * self.M).step
        self.radial_width = self.M * self.h
        self.heaviside_width = self.radial_width - self.pad_zone * self.h
        self.kwargs = kwargs
        self._generate_radial_grid()
        self._generate_qfs_boundaries()

    # -- radial grid (reference :280-358) ---------------------------------------
    def _generate_radial_grid(self):
        bdy = self.bdy
        X, Y, N = bdy.x, bdy.y, bdy.N
        NX, NY = bdy.normal_x, bdy.normal_y
        sign = -1 if self.interior else 1
        self.interface = GSB(x=X + sign * self.radial_width * NX, y=Y + sign * self.radial_width * NY)
        lb = -self.radial_width if self.interior else 0.0
        ub = 0.0 if self.interior else self.radial_width
        rc, rv, rat = get_chebyshev_nodes(lb, ub, self.M)
        self.radial_rv = rv
        self.radial_tv = bdy.t
        self.radial_r, self.radial_t = np.meshgrid(rv, bdy.t, indexing='ij')
        self.radial_x = X + self.radial_r * NX
        self.radial_y = Y + self.radial_r * NY
        self.radial_shape = (self.M, N)
        self.radial_k = np.fft.fftfreq(N, bdy.dt / (2 * np.pi))
        self.radial_speed = bdy.speed * (1.0 + bdy.curvature * self.radial_r)
        # the boundary-fitted coordinates need 1 + curvature * r > 0 across the annulus; near 0
        # they fold (the reference does not check: the solvers then return garbage silently)
        jac = float(np.min(1.0 + bdy.curvature * (lb if self.interior else ub)))
        self.min_radial_jacobian = jac
        if jac <= 0.25:
            import warnings
            warnings.warn("EmbeddedBoundary: the annulus (width %.3g = M*h) is %s for this curve: "
                          "min(1 + curvature*r) = %.2f; use more boundary nodes or a smaller M"
                          % (self.radial_width, "folded" if jac <= 0 else "nearly folded", jac))
        self.inverse_radial_speed = 1.0 / self.radial_speed
        V0 = np.polynomial.chebyshev.chebvander(rc, self.M - 1)
        VI0 = np.linalg.inv(V0)
        DC01 = np.polynomial.chebyshev.chebder(np.eye(self.M)) / rat
        DC00 = np.vstack([DC01, np.zeros(self.M)])
        self.D00 = V0 @ DC00 @ VI0
        e_hi = (np.polynomial.chebyshev.chebvander(1, self.M - 1) @ VI0)[0]
        e_lo = (np.polynomial.chebyshev.chebvander(-1, self.M - 1) @ VI0)[0]
        self.chebyshev_interp_f_to_bdy = e_hi if self.interior else e_lo
        self.chebyshev_interp_f_to_interface = e_lo if self.interior else e_hi
        self.chebyshev_interp_df_dn_to_bdy = self.chebyshev_interp_f_to_bdy @ self.D00
        self.bdy_centroid_x, self.bdy_centroid_y = np.mean(X), np.mean(Y)
        self.approximate_radius = np.mean(np.hypot(X - self.bdy_centroid_x, Y - self.bdy_centroid_y))
        self.radial_targ = PointSet(self.radial_x.ravel(), self.radial_y.ravel())
        # Fejer-1 weights on the Chebyshev-Gauss nodes for radial integrals
        self.radial_quadrature_weights = bdy.dt * _fejer1(self.M)[:, None] * self.radial_width / 2.0 \
            * self.radial_speed
        self._minx = X.min() if self.interior else self.interface.x.min()
        self._maxx = X.max() if self.interior else self.interface.x.max()
        self._miny = Y.min() if self.interior else self.interface.y.min()
        self._maxy = Y.max() if self.interior else self.interface.y.max()

    def _generate_qfs_boundaries(self):
        """(reference :534-551)"""
        eps, fsuf, FF = self.qfs_tolerance, self.qfs_fsuf, self.qfs_FF
        self.bdy_qfs = QFS_Boundary(self.bdy, eps=eps, forced_source_upsampling_factor=fsuf, FF=FF)
        self.interface_qfs = QFS_Boundary(self.interface, eps=eps,
                                          forced_source_upsampling_factor=fsuf, FF=FF)
        q = self.interface_qfs
        self.interface_grid_source = q.interior_source_bdy if self.interior else q.exterior_source_bdy
        self.interface_radial_source = q.exterior_source_bdy if self.interior else q.interior_source_bdy
        q = self.bdy_qfs
        self.bdy_outward_source = q.exterior_source_bdy if self.interior else q.interior_source_bdy
        self.bdy_inward_source = q.interior_source_bdy if self.interior else q.exterior_source_bdy

    # -- grid registration (reference :185-270) ----------------------------------
    def check_if_r_in_annulus(self, r):
        if self.interior:
            w1, w2 = r <= 0, r >= -self.radial_width
        else:
            w1, w2 = r >= 0, r <= self.radial_width
        return np.logical_and(w1, w2), np.logical_not(w2), np.logical_not(w1)

    def save(self):
        """Dictionary sufficient for recreating the object with LoadEmbeddedBoundary (reference
        ipde/embedded_boundary.py:160-176; the heaviside callable is left out, as there)."""
        return {
            'bx': np.array(self.bdy.x), 'by': np.array(self.bdy.y),
            'interior': self.interior, 'M': self.M, 'h': self.h,
            'kwargs': {k: v for k, v in self.kwargs.items() if k != 'heaviside'},
        }

    def register_grid(self, grid, verbose=False):
        """Find the grid points in the annulus, their (r, t) coordinates and the
        cut-off functions.  Returns (r, t, found) on the whole grid for the collection's
        inside/outside classification."""
        self.grid = grid
        # candidates: the band of grid points within reach of the curve (local_coordinates
        # keeps those within 1.5 width + 2 max_h of it)
        reach = 1.5 * self.radial_width + 2 * self.bdy.max_h + max(grid.xh, grid.yh)
        IX, IY = np.nonzero(band_mask(self.bdy, grid, reach))
        r, t, found = local_coordinates(self.bdy, grid.xv[IX], grid.yv[IY], self.radial_width,
                                        tol=self.coordinate_tolerance)
        self._near = (IX[found], IY[found], r[found], t[found])
        ia, _, _ = self.check_if_r_in_annulus(r[found])
        self.grid_ia_xind = IX[found][ia]
        self.grid_ia_yind = IY[found][ia]
        self.grid_ia_x = grid.xv[self.grid_ia_xind]
        self.grid_ia_y = grid.yv[self.grid_ia_yind]
        self.grid_ia_r = r[found][ia]
        self.grid_ia_t = t[found][ia]
        lb = -self.radial_width if self.interior else 0.0
        ub = 0.0 if self.interior else self.radial_width
        self.grid_ia_xi = affine_transformation(self.grid_ia_r, lb, ub, -1.0, 1.0)
        self.interface_x_transf = affine_transformation(self.interface.x, grid.x_bounds[0],
                                                        grid.x_bounds[1], 0.0, 2 * np.pi)
        self.interface_y_transf = affine_transformation(self.interface.y, grid.y_bounds[0],
                                                        grid.y_bounds[1], 0.0, 2 * np.pi)
        # regularised Heaviside functions coupling the grids (reference :239-247)
        lbh = -self.heaviside_width if self.interior else self.heaviside_width
        grts = affine_transformation(self.grid_ia_r, lbh, 0, -1, 1)
        self.grid_to_radial_step = 1.0 - self.heaviside(grts)
        arts = affine_transformation(self.radial_rv, lbh, 0, -1, 1)
        self.radial_cutoff = self.heaviside(arts)
        return self._near

    # -- the old single-boundary API (reference examples/interior_modified_helmholtz.py:44-56:
    # `ebdy.register_grid(grid)` on a bare boundary, then ebdy.phys / ebdy.ext / ...) through a
    # one-boundary collection built on demand
    def solo_collection(self):
        if self._solo is None or self._solo.grid is not self.grid:
            from .ebdy_collection import EmbeddedBoundaryCollection
            c = EmbeddedBoundaryCollection([self])
            c.register_grid(self.grid)
            self._solo = c
        return self._solo

    phys = property(lambda self: self.solo_collection().phys)
    ext = property(lambda self: self.solo_collection().ext)
    grid_step = property(lambda self: self.solo_collection().grid_step)
    grid_in_annulus = property(lambda self: self.solo_collection().in_annulus)
    grid_not_in_annulus = property(lambda self: self.solo_collection().phys_not_in_annulus)

    def interpolate_grid_to_interface(self, f, order=np.inf, cutoff=None):
        return self.solo_collection().interpolate_grid_to_interface(f, order=order, cutoff=cutoff)

    def interpolate_radial_to_grid(self, fr, f):
        return self.interpolate_radial_to_grid1(fr, f)

    def register_ia_inds(self, phys_inds):
        self.ia_inds = phys_inds[self.grid_ia_xind, self.grid_ia_yind]

    # -- radial <-> boundary / grid ------------------------------------------------
    def interpolate_radial_to_boundary(self, f):
        return self.chebyshev_interp_f_to_bdy.dot(f)

    def interpolate_radial_to_interface(self, f):
        return self.chebyshev_interp_f_to_interface.dot(f)

    def interpolate_radial_to_boundary_normal_derivative(self, f):
        return self.chebyshev_interp_df_dn_to_bdy.dot(f)

    def interpolate_radial_to_points(self, fr, xi, t):
        """fr (M, N) on the radial grid -> values at local coordinates (xi in [-1,1], t)
        (reference :419-434; here an exact Chebyshev x Fourier evaluation on the device)."""
        from .interp import chebyshev_fourier_eval
        return chebyshev_fourier_eval(fr, xi, t)

    def interpolate_radial_to_grid1(self, fr, f):
        vals = self.interpolate_radial_to_points(fr, self.grid_ia_xi, self.grid_ia_t)
        f[self.grid_ia_xind, self.grid_ia_yind] = vals.cpu().numpy()

    # -- derivatives on the radial grid (reference :463-483) -------------------------
    def _radial_grid_t_derivative(self, f):
        return np.fft.ifft(np.fft.fft(f) * 1j * self.radial_k).real

    def _radial_grid_tau_derivative(self, f):
        return self._radial_grid_t_derivative(f) * self.inverse_radial_speed

    def _radial_grid_r_derivative(self, f):
        return self.D00.dot(f)

    def radial_grid_derivatives(self, f):
        bdy = self.bdy
        ft = self._radial_grid_tau_derivative(f)
        fr = self._radial_grid_r_derivative(f)
        return fr * bdy.normal_x + ft * bdy.tangent_x, fr * bdy.normal_y + ft * bdy.tangent_y

    def radial_grid_laplacian(self, f):
        """(reference :493-498: the gradient applied twice)"""
        fx, fy = self.radial_grid_derivatives(f)
        fxx, _ = self.radial_grid_derivatives(fx)
        _, fyy = self.radial_grid_derivatives(fy)
        return fxx + fyy

    def gradient(self, fx, fy, fr):
        """Gradient on the radial grid; its values replace fx, fy on the grid points under
        this annulus (reference :499-508)."""
        fxr, fyr = self.radial_grid_derivatives(fr)
        self.interpolate_radial_to_grid1(fxr, fx)
        self.interpolate_radial_to_grid1(fyr, fy)
        return fxr, fyr

    def laplacian(self, lapf, fr):
        """(reference :509-517)"""
        lapfr = self.radial_grid_laplacian(fr)
        self.interpolate_radial_to_grid1(lapfr, lapf)
        return lapfr

    def convert_uv_to_rt(self, fu, fv):
        bdy = self.bdy
        return fu * bdy.normal_x + fv * bdy.normal_y, fu * bdy.tangent_x + fv * bdy.tangent_y

    def convert_rt_to_uv(self, fr, ft):
        bdy = self.bdy
        return fr * bdy.normal_x + ft * bdy.tangent_x, fr * bdy.normal_y + ft * bdy.tangent_y

    def radial_integral(self, fr):
        return np.sum(fr * self.radial_cutoff[:, None] * self.radial_quadrature_weights)


def _fejer1(n):
    """Fejer's first rule weights on the n Chebyshev-Gauss nodes of [-1, 1]."""
    k = np.arange(n)
    theta = (2 * k + 1) * np.pi / (2 * n)
    w = np.ones(n)
    for j in range(1, n // 2 + 1):
        w -= 2.0 * np.cos(2 * j * theta) / (4 * j * j - 1)
    return w * 2.0 / n
