"""
Base class of the time-dependent simulations; same protocol and method names as the reference's
``FenicsSimulation`` (glimslib/simulation/simulation_base.py:36-325), with the FEniCS objects replaced by
numpy arrays on the host and libglimship (HIP, gfx950) behind ``self.solver``.

A model class provides ``_define_model_params``, ``_setup_functionspace``, ``_setup_problem(u_previous)`` -- which
must leave ``self.solver`` (an object with ``.solve()``) and ``self.solution`` -- and ``run_for_adjoint``.
``run()`` reproduces the reference loop (simulation_base.py:236-317): guard ``t <= sim_time - 1e-5``, record every
``keep_nth`` step, on solver failure "warn, stop, return the last solution".
"""
from __future__ import annotations

import logging
import os
from abc import ABC, abstractmethod

import numpy as np

from .. import fenics_local as fenics
from ..simulation_helpers.helper_classes import (SubDomains, FunctionSpace, BoundaryConditions, Parameters, Results,
                                                 Plotting)
from . import config


class FenicsSimulation(ABC):
    """See module docstring.  The name is kept so that reference scripts switch by changing one import."""

    def __init__(self, mesh, time_dependent=True, device=0, solver_options=None):
        self.logger = logging.getLogger(__name__)
        self.mesh = mesh
        self.geometric_dimension = self.mesh.geometry().dim()
        self.time_dependent = time_dependent
        self.projection_parameters = {'solver_type': 'cg', 'preconditioner_type': 'amg'}
        self.functionspace = FunctionSpace(self.mesh, projection_parameters=self.projection_parameters)
        self.device = device
        self.solver_options = dict(solver_options or {})
        self._backend = None            # libglimship handle, created on first _setup_problem
        self._define_model_params()

    # -- hooks -------------------------------------------------------------------------------------------
    @abstractmethod
    def _define_model_params(self):
        self.required_params = []
        self.optional_params = []

    @abstractmethod
    def _setup_functionspace(self):
        ...

    @abstractmethod
    def _setup_problem(self, u_previous):
        ...

    @abstractmethod
    def run_for_adjoint(self, parameters):
        ...

    # -- setup -------------------------------------------------------------------------------------------
    def setup_global_parameters(self, label_function=None, subdomains=None, domain_names=None, boundaries=None,
                                dirichlet_bcs=None, von_neumann_bcs=None):
        """simulation_base.py:160-198 -- same arguments; meshes, label functions and boundary selectors are arrays /
        callables / SubDomain instances of ``glimslib_amd.fenics_local``."""
        self.logger.info("-- Setting up global parameters")
        self.geometric_dimension = self.mesh.geometry().dim()
        self.subdomains = SubDomains(self.mesh)
        self.subdomains.setup_subdomains(label_function=label_function, subdomains=subdomains, replace=False)
        self.subdomains.setup_boundaries(tissue_map=domain_names, boundary_fct_dict=boundaries)
        self.subdomains.setup_measures()
        self._setup_functionspace()
        self.bcs = BoundaryConditions(self.functionspace, self.subdomains)
        self.bcs.setup_dirichlet_boundary_conditions(dirichlet_bcs)
        self.bcs.setup_von_neumann_boundary_conditions(von_neumann_bcs)
        self._close_backend()           # new labels / mesh usage -> new device discretisation

    def setup_model_parameters(self, iv_expression, **kwargs):
        """simulation_base.py:200-217"""
        self._define_model_params()
        self.params = Parameters(self.functionspace, self.subdomains, time_dependent=self.time_dependent)
        self.params.set_initial_value_expressions(iv_expression)
        self.params.define_required_params(self.required_params)
        self.params.define_optional_params(self.optional_params)
        self.params.init_parameters(kwargs)

    def _update_expressions(self, time):
        """simulation_base.py:219-226"""
        self.params.time_update_parameters(time)
        self.bcs.time_update_bcs(time, kind='dirichlet')
        self.bcs.time_update_bcs(time, kind='von-neumann')
        # extension: the reference leaves `source_term` / `body_force` (plain attributes, stg:91-96) at their initial
        # `.t`; here they follow the simulation time like every other expression
        for name in ('source_term', 'rd_source_term', 'body_force'):
            obj = getattr(self, name, None)
            if obj is not None and hasattr(obj, 't'):
                obj.t = time

    # -- backend lifetime ----------------------------------------------------------------------------------
    def _update_mesh_displacements(self, displacement):
        """
        simulation_base.py:228-234 (``fenics.ALE.move``): adds the nodal displacement to the mesh coordinates.
        As in the reference this changes the current mesh and repeated calls add up; the device discretisation of a
        running simulation is not rebuilt (the reference only uses this for plotting deformed configurations).
        """
        u = displacement.values() if hasattr(displacement, 'values') else np.asarray(displacement, dtype=np.float64)
        self.mesh.points += u.reshape(self.mesh.points.shape)
        self.mesh._facets = None

    def _close_backend(self):
        if self._backend is not None:
            self._backend.close()
            self._backend = None

    def close(self):
        self._close_backend()

    def __del__(self):
        try:
            self._close_backend()
        except Exception:
            pass

    # -- run -----------------------------------------------------------------------------------------------
    def run(self, keep_nth=1, save_method='xdmf', clear_all=False, plot=True,
            output_dir=config.output_dir_simulation_tmp, results_on_device=None):
        """
        simulation_base.py:236-317.  ``save_method``: None, 'vtk' (<field>/<field>_<step>.pvd + .vtu, the reference's
        layout) or 'xdmf' (solution.xdmf + solution.bin instead of solution.h5).
        Returns ``self.solution`` (mixed Function {0: displacement, 1: concentration}).

        ``results_on_device`` (extension): keep the recorded steps in HBM and materialise them lazily -- the
        concentration is downloaded, and the displacement of a recorded step is solved, only when that step is
        accessed.  None = automatic (on for meshes of >= 100 000 nodes when nothing is written to disk per step).
        """
        if self.geometric_dimension == 3:
            plot = False
        self.logger.info("-- Computing solutions: ")
        self.results = Results(self.functionspace, self.subdomains,
                               output_dir=output_dir if save_method is not None else None)
        self.results.save_solution_start(method=save_method, clear_all=clear_all)
        self.plotting = Plotting(self.results, output_dir=os.path.join(output_dir, 'plots'))
        u_previous = self.params.create_initial_value_function()
        self._setup_problem(u_previous)

        if not self.time_dependent:
            raise NotImplementedError("only the time-dependent tumour-growth models are implemented on this backend")

        current_sim_time = 0.0
        self._update_expressions(current_sim_time)
        time_step = 0
        recording_step = 0
        self.results.add_to_results(0, 0, recording_step, u_previous)
        self.results.save_solution(recording_step, current_sim_time, function=u_previous, method=save_method)
        continue_simulation = True
        dt = float(self.params.sim_time_step)
        sim_time = float(self.params.sim_time)
        keep_nth = max(1, int(keep_nth))
        single_step = self.solver.time_dependent_inputs
        if results_on_device is None:
            results_on_device = self.mesh.num_vertices() >= 100000 and save_method is None
        lazy = bool(results_on_device) and getattr(self.solver, 'supports_snapshots', lambda: False)()
        while (current_sim_time <= sim_time - 1e-5) and continue_simulation:
            # steps until the next recorded step (or until the loop guard fails); they run back to back on the
            # device unless some input carries a time attribute `.t`, which forces the reference's per-step updates
            k, t = 0, current_sim_time
            while t <= sim_time - 1e-5:
                t += dt
                k += 1
                if single_step or (time_step + k) % keep_nth == 0:
                    break
            if single_step:
                self._update_expressions(current_sim_time + dt)
                self.solver.update_time_dependent_inputs()
            self.logger.info("    - solving for time = %.2f / %.2f" % (current_sim_time + k * dt, sim_time))
            try:
                self.solver.solve(k)
                done = k
            except Exception as exc:            # reference: bare except around solver.solve() (:301-305)
                self.logger.warning("    - Solver did not converge -- will shutdown simulation (%s)" % exc)
                continue_simulation = False
                # the device counts converged steps; the attempt that failed advances the clock too, as in the reference
                # (t += dt precedes solver.solve(), :297-302)
                done = getattr(self.solver, 'steps_done_in_last_call', 0) + 1
            current_sim_time += done * dt
            time_step += done
            if (time_step % keep_nth == 0) and continue_simulation:
                recording_step += 1
                if lazy:
                    self.results.add_to_results(current_sim_time, time_step, recording_step, self.solver.snapshot())
                else:
                    self.solver.sync_solution(with_mechanics=True)
                    self.results.add_to_results(current_sim_time, time_step, recording_step, self.solution)
                self.results.save_solution(recording_step, current_sim_time, method=save_method)

        self.solver.sync_solution(with_mechanics=continue_simulation)
        u_previous.assign(self.solution)
        self.results.save_solution_end(method=save_method)
        if save_method is not None:
            self.results.save_solution_hdf5()
        return self.solution

    run_forward = run   # BASELINE.json's name for the forward entry point

    def reload_from_hdf5(self, path_to_hdf5, output_dir=config.output_dir_simulation_tmp):
        """simulation_base.py:319-325 (time series are stored as .npz by this build)"""
        self.results = Results(self.functionspace, self.subdomains, output_dir=output_dir)
        self.results.data.load_from_hdf5(path_to_hdf5)
        self.plotting = Plotting(self.results, output_dir=os.path.join(output_dir, 'plots'))
