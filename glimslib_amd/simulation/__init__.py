# The reference's scripts import these from the package root
# (test_case_comparison_3D_atlas.py:16) although its own __init__ is empty; here they exist.
from .simulation_base import FenicsSimulation  # noqa: F401
from .simulation_tumor_growth import TumorGrowth, SolverDidNotConverge  # noqa: F401
from .simulation_tumor_growth_brain import TumorGrowthBrain  # noqa: F401
