"""Counterpart of glimslib/simulation/config.py:1-3 (output location of simulation runs)."""
import os
import tempfile

output_dir_simulation_tmp = os.path.join(tempfile.gettempdir(), "glimslib_amd_output", "simulation_tmp")
USE_ADJOINT = False   # glimslib/config.py: the dolfin-adjoint mode is not reproduced
