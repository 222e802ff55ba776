from .helper_classes import (SubDomains, FunctionSpace, SubSpaces, BoundaryConditions, Parameters, Results,  # noqa: F401
                             TimeSeriesData, TimeSeriesDataTimePoint, TimeSeriesMultiData, DiscontinuousScalar,
                             Boundary)
from .postprocess import PostProcess, PostProcessTumorGrowth, PostProcessTumorGrowthBrain, Comparison  # noqa: F401
