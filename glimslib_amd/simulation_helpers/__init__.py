from .helper_classes import (SubDomains, FunctionSpace, SubSpaces, BoundaryConditions, Parameters, Results,  # noqa: F401
                             TimeSeriesData, TimeSeriesDataTimePoint, TimeSeriesMultiData, DiscontinuousScalar,
                             Boundary)
