"""EmbeddedFunction / BoundaryFunction containers — the data layout of
ipde/embedded_function.py (reference :16-342): a flat fp64 vector
[physical grid values in C-order of the `phys` mask | radial (M, N) blocks, one per
boundary], with a weak reference to the collection.  Host-side containers; the hot
path reads them through get_components() / load_data() (reference :105-113,135-138).
"""
import weakref

import numpy as np


def LoadEmbeddedFunction(d, ebdyc=None):
    """(EmbeddedFunction, its collection) from a `save` dictionary and a collection, or from a
    `full_save` dictionary alone (reference ipde/embedded_function.py:6-14)."""
    if ebdyc is None:
        if 'ebdyc_dict' not in d:
            raise Exception('Need ebdyc provided unless save was generated with full_save method.')
        from .ebdy_collection import LoadEmbeddedBoundaryCollection
        ebdyc = LoadEmbeddedBoundaryCollection(d['ebdyc_dict'])
    ef = EmbeddedFunction(ebdyc)
    ef.load_linear_data(d['linear_data'])
    return ef, ebdyc


class EmbeddedFunction(np.ndarray):
    def __new__(cls, ebdyc, dtype=float, array=None, function=None, grid_value=None,
                radial_value_list=None, linear_data=None, zero=False):
        gn = ebdyc.grid_phys.N
        rn = int(np.sum([np.prod(ebdy.radial_shape) for ebdy in ebdyc]))
        if array is None:
            array = np.zeros(gn + rn, dtype=dtype)
        out = np.asarray(array).view(cls)
        out.ebdyc = weakref.ref(ebdyc)
        out._generate()
        if function is not None:
            out.define_via_function(function)
        if grid_value is not None and radial_value_list is not None:
            out.load_data(grid_value, radial_value_list)
        if linear_data is not None:
            out.load_linear_data(linear_data)
        if zero:
            out[:] = 0.0
        return out

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.ebdyc = getattr(obj, 'ebdyc', None)
        # (set by a solver's sharded result: which entries are complete on this rank)
        own = getattr(obj, 'owned', None)
        if own is not None and getattr(own, 'shape', None) == self.shape:
            self.owned = own
        if self.ebdyc is not None and self.ebdyc() is not None and self.ndim == 1:
            try:
                self._generate()
            except Exception:
                pass

    def _ebdyc_test(self):
        ebdyc = self.ebdyc() if self.ebdyc is not None else None
        if ebdyc is None:
            raise Exception('Underlying ebdyc has been deleted.')
        return ebdyc

    def _generate(self):
        ebdyc = self._ebdyc_test()
        self.n_grid = ebdyc.grid_phys.N
        self.gslice = slice(0, self.n_grid)
        self.radial_slices, self.radial_shapes = [], []
        start = self.n_grid
        for ebdy in ebdyc:
            rsh = ebdy.radial_shape
            end = start + int(np.prod(rsh))
            self.radial_slices.append(slice(start, end))
            self.radial_shapes.append(rsh)
            start = end
        self.radial_length = len(self.radial_shapes)

    # -- accessors ------------------------------------------------------------
    def get_gdata(self):
        return np.ndarray.__getitem__(self, self.gslice).view(np.ndarray)

    def __getitem__(self, arg):
        if isinstance(arg, (int, np.integer)):
            sl, rsh = self.radial_slices[arg], self.radial_shapes[arg]
            return np.ndarray.__getitem__(self, sl).view(np.ndarray).reshape(rsh)
        if isinstance(arg, str) and arg == 'grid':
            return self.get_gdata()
        return np.ndarray.__getitem__(self, arg).view(np.ndarray)

    def __setitem__(self, arg, value):
        if isinstance(arg, (int, np.integer)):
            np.ndarray.__setitem__(self, self.radial_slices[arg], np.asarray(value).ravel())
        elif isinstance(arg, str) and arg == 'grid':
            np.ndarray.__setitem__(self, self.gslice, value)
        else:
            np.ndarray.__setitem__(self, arg, value)

    def __iter__(self):
        for i in range(self.radial_length):
            yield self[i]

    def load_data(self, grid_value, radial_value_list):
        gd = self.get_gdata()
        grid_value = np.asarray(grid_value)
        if grid_value.shape == gd.shape:
            gd[:] = grid_value
        else:
            gd[:] = grid_value[self._ebdyc_test().phys]
        for arg, rv in enumerate(radial_value_list):
            self[int(arg)] = rv

    def load_linear_data(self, data):
        np.ndarray.__setitem__(self, slice(None), data)

    def define_via_function(self, f):
        ebdyc = self._ebdyc_test()
        self.get_gdata()[:] = f(ebdyc.grid_phys.x, ebdyc.grid_phys.y)
        for arg, ebdy in enumerate(ebdyc):
            self[int(arg)] = f(ebdy.radial_x, ebdy.radial_y)

    def get_radial_value_list(self):
        return [self[int(i)] for i in range(self.radial_length)]

    def get_grid_value(self, masked=False):
        ebdyc = self._ebdyc_test()
        g = np.zeros(ebdyc.grid.shape, dtype=self.dtype)
        g[ebdyc.phys] = self.get_gdata()
        return np.ma.array(g, mask=ebdyc.ext) if masked else g

    def get_smoothed_grid_value(self):
        return self.get_grid_value() * self._ebdyc_test().grid_step

    def get_components(self):
        """(grid values, grid values * grid_step, [radial arrays])  (reference :135-138)"""
        g = self.get_grid_value()
        return g, g * self._ebdyc_test().grid_step, self.get_radial_value_list()

    def integrate(self):
        """volume integral over the physical domain (reference :214-218)"""
        return self._ebdyc_test().volume_integral(self)

    def gradient(self, derivative_type='spectral'):
        """(reference :219-220)"""
        return self._ebdyc_test().gradient(self, derivative_type)

    def get_ebdyc(self):
        return self._ebdyc_test()

    def get_rdata(self):
        """all radial values, flattened (reference :82-83)"""
        return np.ndarray.__getitem__(self, slice(self._ebdyc_test().grid_phys.N, None)).view(np.ndarray)

    def asarray(self):
        return np.array(self)

    def min(self):
        return float(np.min(self.view(np.ndarray)))

    def max(self):
        return float(np.max(self.view(np.ndarray)))

    def define_via_functions(self, f_grid, f_radial_list):
        """grid values from f_grid(x, y), radial values of boundary i from f_radial_list[i]"""
        ebdyc = self._ebdyc_test()
        self.get_gdata()[:] = f_grid(ebdyc.grid_phys.x, ebdyc.grid_phys.y)
        for arg, (ebdy, fr) in enumerate(zip(ebdyc, f_radial_list)):
            self[int(arg)] = fr(ebdy.radial_x, ebdy.radial_y)

    def load_full_grid(self, grid_value):
        """Load from values given on the WHOLE grid; the radial values are interpolated from
        them, which is only meaningful for a function smooth across the whole box
        (reference :114-121)."""
        ebdyc = self._ebdyc_test()
        self.load_data(grid_value, ebdyc.interpolate_grid_to_radial(grid_value))

    def extract_pnar(self):
        """[values on phys-not-in-annulus grid points | all radial values] (reference :223-229)"""
        ebdyc = self._ebdyc_test()
        gv = self.get_grid_value()
        return np.concatenate([gv[ebdyc.phys_not_in_annulus], self.get_rdata()])

    def copy(self):
        return EmbeddedFunction(self._ebdyc_test(), array=np.array(self.view(np.ndarray), copy=True))

    def zero(self):
        self[:] = 0.0

    def save(self):
        return {'linear_data': np.array(self.view(np.ndarray))}

    def full_save(self):
        """`save` plus the collection's own dictionary: LoadEmbeddedFunction then needs nothing
        else (reference ipde/embedded_function.py:56-61)."""
        return {'ebdyc_dict': self._ebdyc_test().save(), 'linear_data': np.array(self.view(np.ndarray))}


class BoundaryFunction(np.ndarray):
    """Flat vector of boundary-node values, one block per boundary (reference :231-342)."""

    def __new__(cls, ebdyc, dtype=float, array=None, function=None, data=None):
        n = int(np.sum([ebdy.bdy.N for ebdy in ebdyc]))
        if array is None:
            array = np.zeros(n, dtype=dtype)
        out = np.asarray(array).view(cls)
        out.ebdyc = weakref.ref(ebdyc)
        out._generate()
        if function is not None:
            out.define_via_function(function)
        if data is not None:
            out.load_data(data)
        return out

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.ebdyc = getattr(obj, 'ebdyc', None)
        if self.ebdyc is not None and self.ebdyc() is not None and self.ndim == 1:
            try:
                self._generate()
            except Exception:
                pass

    def _generate(self):
        ebdyc = self.ebdyc()
        self.slices = []
        start = 0
        for ebdy in ebdyc:
            self.slices.append(slice(start, start + ebdy.bdy.N))
            start += ebdy.bdy.N

    @property
    def bdy_value_list(self):
        return [np.ndarray.__getitem__(self, sl).view(np.ndarray) for sl in self.slices]

    def __getitem__(self, arg):
        if isinstance(arg, (int, np.integer)):
            return np.ndarray.__getitem__(self, self.slices[arg]).view(np.ndarray)
        return np.ndarray.__getitem__(self, arg).view(np.ndarray)

    def __setitem__(self, arg, value):
        if isinstance(arg, (int, np.integer)):
            np.ndarray.__setitem__(self, self.slices[arg], value)
        else:
            np.ndarray.__setitem__(self, arg, value)

    def load_data(self, value_list):
        for sl, v in zip(self.slices, value_list):
            np.ndarray.__setitem__(self, sl, v)

    def define_via_functions(self, f_list):
        """(reference :302-307)"""
        for sl, ebdy, f in zip(self.slices, self.ebdyc(), f_list):
            np.ndarray.__setitem__(self, sl, f(ebdy.bdy.x, ebdy.bdy.y))

    def asarray(self):
        return np.array(self)

    def min(self):
        return float(np.min(self.view(np.ndarray)))

    def max(self):
        return float(np.max(self.view(np.ndarray)))

    def zero(self):
        np.ndarray.__setitem__(self, slice(None, None), 0.0)

    def define_via_function(self, f):
        for sl, ebdy in zip(self.slices, self.ebdyc()):
            np.ndarray.__setitem__(self, sl, f(ebdy.bdy.x, ebdy.bdy.y))
