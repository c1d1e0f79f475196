"""Drop-in import surface: lets a script written against the reference stack import the
names it expects —

    import pybie2d                                  (.misc.curve_descriptions.star, ...)
    from ipde.embedded_boundary import EmbeddedBoundary
    from ipde.solvers.multi_boundary.poisson import PoissonSolver
    from qfs.two_d_qfs import QFS_Evaluator
    from personal_utilities.arc_length_reparametrization import arc_length_parameterize

— and get this package's classes (the reference's scripts:
examples/interior_poisson.py:1-20, interior_modified_helmholtz.py:1-16, multi_stokes.py:1-21).

    import ipde_amd.compat; ipde_amd.compat.install()      # before the script's imports
    python -m ipde_amd.compat path/to/script.py [args]     # or run a script under it

`ipde.X` resolves to the module object `ipde_amd.X` itself (an alias, not a second copy:
classes stay identical).  pybie2d / qfs / personal_utilities are small module trees whose
leaves are the objects of `ipde_amd.pybie2d_compat` / `ipde_amd.qfs`.  A real installation
of any of these packages wins: install() only fills names that do not import."""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types

_PREFIX = "ipde"
_TARGET = "ipde_amd"


# module names the reference's scripts import that its tree no longer has under that name
# (examples/poisson_for_paper.py:9 imports ipde.embedded_boundary_standalone)
_RENAMED = {".embedded_boundary_standalone": ".embedded_boundary"}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        mod = importlib.import_module(self.target)
        self._spec = mod.__spec__
        return mod

    def exec_module(self, module):
        # the import machinery has just pointed the shared module's __spec__ at the alias
        # name: give the module its own spec back (repr, reload, pickling by module name)
        module.__spec__ = self._spec


class _IpdeAliasFinder(importlib.abc.MetaPathFinder):
    """`ipde[.x.y]` -> the already-imported module `ipde_amd[.x.y]`"""

    def find_spec(self, name, path=None, target=None):
        if name != _PREFIX and not name.startswith(_PREFIX + "."):
            return None
        real = _TARGET + _RENAMED.get(name[len(_PREFIX):], name[len(_PREFIX):])
        try:
            mod = importlib.import_module(real)
        except ImportError:
            return None
        return importlib.machinery.ModuleSpec(name, _AliasLoader(real), is_package=hasattr(mod, "__path__"))


def _tree(root, leaves):
    """register a module tree `root.a.b = obj` in sys.modules from {'a.b.name': obj}"""
    made = {}

    def mod(name):
        m = made.get(name) or sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            m.__path__ = []          # a package: `import root.a.b` works
            m.__doc__ = "ipde_amd.compat stand-in for %s" % name
            sys.modules[name] = m
            if "." in name:
                parent, child = name.rsplit(".", 1)
                setattr(mod(parent), child, m)
        made[name] = m
        return m
    mod(root)
    for dotted, obj in leaves.items():
        parent, leaf = dotted.rsplit(".", 1) if "." in dotted else ("", dotted)
        setattr(mod(root + ("." + parent if parent else "")), leaf, obj)
    return made[root]


def _importable(name):
    try:
        importlib.import_module(name)
        return True
    except Exception:
        return False


_installed = False


def install():
    """Idempotent.  Returns the list of top-level names that were filled in."""
    global _installed
    if _installed:
        return []
    filled = []
    from . import pybie2d_compat as P
    from . import qfs as Q
    from . import layer_potentials as L
    if not _importable("ipde"):
        sys.meta_path.insert(0, _IpdeAliasFinder())   # ahead of PathFinder, which would load second copies via ipde_amd.__path__
        filled.append("ipde")
    if not _importable("pybie2d"):
        _tree("pybie2d", {
            "misc.curve_descriptions.star": P.star,
            "misc.curve_descriptions.squished_circle": P.squish,
            "boundaries.global_smooth_boundary.global_smooth_boundary.Global_Smooth_Boundary":
                P.Global_Smooth_Boundary,
            "boundaries.collection.BoundaryCollection": P.BoundaryCollection,
            "grid.Grid": P.Grid,
            "point_set.PointSet": P.PointSet,
            "kernels.high_level.laplace.Laplace_Layer_Form": P.Laplace_Layer_Form,
            "kernels.high_level.laplace.Laplace_Layer_Singular_Form": P.Laplace_Layer_Singular_Form,
            "kernels.high_level.laplace.Laplace_Layer_Apply": L.Laplace_Layer_Apply,
            "kernels.high_level.modified_helmholtz.Modified_Helmholtz_Layer_Form":
                P.Modified_Helmholtz_Layer_Form,
            "kernels.high_level.modified_helmholtz.Modified_Helmholtz_Layer_Singular_Form":
                P.Modified_Helmholtz_Layer_Singular_Form,
            "kernels.high_level.modified_helmholtz.Modified_Helmholtz_Layer_Apply":
                L.Modified_Helmholtz_Layer_Apply,
            "kernels.high_level.stokes.Stokes_Layer_Form": P.Stokes_Layer_Form,
            "kernels.high_level.stokes.Stokes_Layer_Singular_Form": P.Stokes_Layer_Singular_Form,
            "kernels.high_level.stokes.Stokes_Layer_Apply": L.Stokes_Layer_Apply,
        })
        pb = sys.modules["pybie2d"]
        # the shortcuts pybie2d itself exposes at package level
        pb.Grid, pb.PointSet = P.Grid, P.PointSet
        filled.append("pybie2d")
    if not _importable("qfs"):
        _tree("qfs", {
            "two_d_qfs.QFS_Evaluator": Q.QFS_Evaluator,
            "two_d_qfs.QFS_Boundary": Q.QFS_Boundary,
            "laplace_qfs.Laplace_QFS": Q.Laplace_QFS,
            "modified_helmholtz_qfs.Modified_Helmholtz_QFS": Q.Modified_Helmholtz_QFS,
            "stokes_qfs.Stokes_QFS": Q.Stokes_QFS,
        })
        filled.append("qfs")
    if not _importable("personal_utilities"):
        _tree("personal_utilities", {
            "arc_length_reparametrization.arc_length_parameterize": P.arc_length_parameterize,
        })
        filled.append("personal_utilities")
    _installed = True
    return filled


def main(argv=None):
    import runpy
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise SystemExit("usage: python -m ipde_amd.compat script.py [args...]")
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
