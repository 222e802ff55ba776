"""
glimslib_amd -- MI355X-native forward solver for GlimSLib's mechanically-coupled reaction-diffusion tumour-growth
model.  Keeps the reference's user surface (``TumorGrowth`` / ``TumorGrowthBrain``: ``setup_global_parameters``,
``setup_model_parameters``, ``run``), replaces FEniCS assemble + PETSc SNES/LU by hand-written HIP kernels
(libglimship.so).  There is no CPU compute path: importing the solver classes without the built library raises.
"""
__version__ = "0.1.0"

from . import fenics_local  # noqa: F401
from .mesh import Mesh, RectangleMesh, BoxMesh, UnitSquareMesh, UnitCubeMesh  # noqa: F401


def __getattr__(name):
    if name in ("TumorGrowth", "TumorGrowthBrain", "FenicsSimulation"):
        from . import simulation
        return getattr(simulation, name)
    raise AttributeError(name)
