"""
Mechanically-coupled reaction-diffusion model of tumour growth -- counterpart of
glimslib/simulation/simulation_tumor_growth.py:13-173 on the HIP backend.

Equations (reference :110-120), P1 elements, backward Euler, cell-wise constant coefficients:

    F_m  = int sigma(u):eps(v) dx - int sigma(v):(c gamma I) dx - int f.v dx - oint g_u.v ds          (:110-113)
    F_rd = int (c - c_prev) w dx + dt int D grad c.grad w dx - dt int rho c (1 - c) w dx
           - dt int s w dx - dt oint g_c D w ds                                                         (:115-120)

The reference solves F = F_m + F_rd monolithically with SNES + LU each step (:124-130).  F_rd does not contain u,
and F_m is linear in u, so the backend advances c with Newton-PCG on the device and solves the (constant) elastic
system for u only at recorded steps; same fixed point, checked against the monolithic oracle in tests/.
"""
from __future__ import annotations

import logging
import os

import numpy as np

from .. import fenics_local as fenics
from ..fenics_local import Constant, Expression, Function, interpolate_nodal
from ..simulation_helpers import math_linear_elasticity as mle, math_reaction_diffusion as mrd  # noqa: F401
from ..simulation_helpers.helper_classes import DiscontinuousScalar
from .simulation_base import FenicsSimulation
from . import config
from .. import _backend


class SolverDidNotConverge(RuntimeError):
    pass


class HipTimeStepSolver:
    """
    What ``self.solver`` is on this backend: ``solve()`` advances the concentration by one implicit step
    (``solve(k)`` by k steps without returning to Python), raising on non-convergence like DOLFIN's solver does.
    """

    def __init__(self, sim, handle, mechanics):
        self.sim = sim
        self.handle = handle
        self.mechanics = mechanics
        self.steps_done_in_last_call = 0
        self.time_dependent_inputs = sim._has_time_dependent_inputs()
        self._mech_current = False

    def solve(self, n_steps=1):
        before = self.handle.stats()['steps']
        status = self.handle.step(int(n_steps))
        self.steps_done_in_last_call = int(self.handle.stats()['steps'] - before)
        self._mech_current = False
        if status != _backend.GLIMS_OK:
            raise SolverDidNotConverge("libglimship status %d (%s)" % (
                status, {1: "iteration cap reached", 2: "non-finite residual"}.get(status, "?")))

    def update_time_dependent_inputs(self):
        self.sim._upload_loads_and_bcs(self.handle)

    def supports_snapshots(self):
        return hasattr(self.handle, 'snapshot_save')

    def snapshot(self):
        """Record the current step on the device; returns a lazily materialised Function."""
        sid = self.handle.snapshot_save()
        return DeviceSnapshotFunction(self.sim.mesh, self.sim.functionspace.subspaces.names, self, sid)

    def sync_solution(self, with_mechanics=True):
        if self.mechanics and with_mechanics and not self._mech_current:
            st = self.handle.solve_mechanics()
            if st != _backend.GLIMS_OK:
                self.sim.logger.warning("    - displacement solve did not converge (status %d)" % st)
            self._mech_current = True
        c, u = self.handle.get_state(want_u=self.mechanics)
        sol = self.sim.solution
        sol.components[1] = c
        if u is not None:
            sol.components[0] = u.reshape(-1, self.sim.geometric_dimension)


class _LazyComponents(dict):
    """{0: displacement, 1: concentration} of one recorded step, fetched / solved on first access."""

    def __init__(self, solver, sid, dim, n):
        super().__init__()
        self._solver, self._sid, self._dim, self._n = solver, sid, dim, n

    def _materialise(self, k):
        if dict.__contains__(self, k):
            return
        if k == 1:
            dict.__setitem__(self, 1, self._solver.handle.snapshot_load(self._sid))
        elif k == 0:
            if self._solver.mechanics:
                u, st = self._solver.handle.snapshot_mechanics(self._sid)
                self._solver._mech_current = False          # the device's displacement buffer now belongs to this step
                if st != _backend.GLIMS_OK:
                    self._solver.sim.logger.warning("    - displacement solve did not converge (status %d)" % st)
                dict.__setitem__(self, 0, u.reshape(self._n, self._dim))
            else:
                dict.__setitem__(self, 0, np.zeros((self._n, self._dim)))
        else:
            raise KeyError(k)

    def __getitem__(self, k):
        self._materialise(k)
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k in (0, 1)

    def keys(self):
        return [0, 1]

    def __iter__(self):
        return iter([0, 1])

    def __len__(self):
        return 2

    def items(self):
        return [(k, self[k]) for k in (0, 1)]

    def values(self):
        return [self[k] for k in (0, 1)]


class DeviceSnapshotFunction(Function):
    """
    A recorded solution that lives on the device (``glims_snapshot_save``): the concentration is downloaded and the
    displacement is SOLVED only when first accessed -- legitimate because the displacement of a step depends on that
    step's concentration alone (one-way coupling, simulation_tumor_growth.py:110-120).  Immutable: ``copy()`` returns
    self, which is what ``Results.add_to_results`` (a deep copy per recorded step in the reference) stores.
    """

    def __init__(self, mesh, names, solver, sid):
        self.mesh = mesh
        self.names = names
        self._name = 'solution_function'
        self.label = 'solution_function'
        self.components = _LazyComponents(solver, sid, mesh.dim, mesh.num_vertices())
        self.snapshot_id = sid

    def copy(self, deepcopy=True):
        return self

    def assign(self, other):
        raise TypeError("device snapshots are immutable")


class TumorGrowth(FenicsSimulation):
    """See module docstring; constructor and parameter names as in the reference (:63, :74-76)."""

    def __init__(self, mesh, time_dependent=True, **kwargs):
        super().__init__(mesh, time_dependent=time_dependent, **kwargs)
        self.units = {'motility': 'm^2/s', 'Emodulus': 'N/m^2', 'none': '', 'growth_rate': '1/s'}

    def _setup_functionspace(self):
        """Mixed space [P1^dim, P1], names {0: 'displacement', 1: 'concentration'} (:67-72)."""
        element = {0: self.geometric_dimension, 1: 1}
        subspace_names = {0: 'displacement', 1: 'concentration'}
        self.functionspace.init_function_space(element, subspace_names)

    def _define_model_params(self):
        self.required_params = ['diffusion', 'coupling', 'proliferation', 'E', 'poisson']
        self.optional_params = []

    # -- coefficient tables -----------------------------------------------------------------------------------
    def _labels(self):
        return np.asarray(self.subdomains.subdomains.array(), dtype=np.int32)

    @staticmethod
    def _table(param, n_labels):
        if isinstance(param, DiscontinuousScalar):
            return param.table(n_labels)
        if isinstance(param, Constant):
            param = float(param)
        return np.full(n_labels, float(param))

    def _material_tables(self, n_labels):
        p = self.params
        return dict(D=self._table(p.diffusion, n_labels), rho=self._table(p.proliferation, n_labels),
                    gamma=self._table(p.coupling, n_labels), E=self._table(p.E, n_labels),
                    nu=self._table(p.poisson, n_labels))

    # -- loads / BCs --------------------------------------------------------------------------------------------
    def _rd_source(self):
        """The source term s of F_rd ('source_term', :95-96, :119)."""
        return getattr(self, 'source_term', None)

    def _time_objects(self):
        objs = [self._rd_source(), getattr(self, 'source_term', None), getattr(self, 'body_force', None)]
        for attr in ('dirichlet_bcs_dict', 'von_neumann_bcs_dict'):
            objs += [bc.get('bc_value') for bc in getattr(self.bcs, attr, {}).values()]
        return objs

    def _has_time_dependent_inputs(self):
        return any(hasattr(o, 't') for o in self._time_objects() if o is not None)

    def _lumped(self):
        vol = self.mesh.cell_volumes()
        d = self.geometric_dimension
        out = np.zeros(self.mesh.num_vertices())
        np.add.at(out, self.mesh.cells.reshape(-1), np.repeat(vol / (d + 1), d + 1))
        return out

    def _mass_apply(self, s):
        vol = self.mesh.cell_volumes()
        d = self.geometric_dimension
        sl = s[self.mesh.cells]
        loc = (vol / ((d + 1) * (d + 2)))[:, None] * (sl + sl.sum(axis=1, keepdims=True))
        out = np.zeros(self.mesh.num_vertices())
        np.add.at(out, self.mesh.cells.reshape(-1), loc.reshape(-1))
        return out

    def _upload_loads_and_bcs(self, h):
        d = self.geometric_dimension
        dt = float(self.params.sim_time_step)
        n = self.mesh.num_vertices()
        labels = self._labels()
        # reaction-diffusion load: dt * ( int s w dx + oint g D w ds )                       (:119-120)
        rd = np.zeros(n)
        src = self._rd_source()
        if src is not None:
            if isinstance(src, (int, float)) or (isinstance(src, Constant)):
                if float(src) != 0.0:
                    rd += float(src) * self._lumped()
            else:
                rd += self._mass_apply(interpolate_nodal(src, self.mesh, 1))
        if getattr(self.bcs, 'von_neumann_bcs', None):
            Dcell = self._table(self.params.diffusion, int(labels.max()) + 1)[labels]
            rd += self.bcs.implement_von_neumann_bc(Dcell, subspace_id=1)
        h.set_rd_load(dt * rd if np.any(rd != 0.0) else None)
        # mechanical load: int f.v dx + oint g.v ds                                           (:112-113)
        ml = np.zeros(n * d)
        bf = getattr(self, 'body_force', None)
        if bf is not None:
            f = interpolate_nodal(bf, self.mesh, d)
            if np.any(f != 0.0):
                if isinstance(bf, Constant) or not callable(bf):
                    ml += (self._lumped()[:, None] * f).reshape(-1)
                else:
                    ml += np.stack([self._mass_apply(f[:, a]) for a in range(d)], axis=1).reshape(-1)
        if getattr(self.bcs, 'von_neumann_bcs', None):
            ml += self.bcs.implement_von_neumann_bc(None, subspace_id=0)
        h.set_mech_load(ml if np.any(ml != 0.0) else None)
        dofs, vals = self.bcs.dirichlet_dofs(0)
        h.set_dirichlet_u(dofs, vals)
        nodes, cvals = self.bcs.dirichlet_dofs(1)
        h.set_dirichlet_c(nodes, cvals)
        return nodes, cvals

    # -- problem --------------------------------------------------------------------------------------------------
    def _setup_problem(self, u_previous):
        labels = self._labels()
        if labels.min() < 0 or labels.max() > 255:
            raise ValueError("tissue ids must lie in [0, 255]")
        n_labels = int(labels.max()) + 1
        if not hasattr(self, 'body_force'):
            self.body_force = Constant(np.zeros(self.geometric_dimension))
        if not hasattr(self, 'source_term'):
            self.source_term = Constant(0.0)
        mechanics = bool(self.solver_options.get('mechanics', True))
        if self._backend is None:
            from ..parallel import dist_info, DistributedHandle
            dist, rank, world = dist_info()
            if world > 1:
                # SPMD like the reference under mpirun: this rank owns a Morton range of the nodes on its own GPU
                device = int(os.environ.get("GLIMS_FORCE_DEVICE", os.environ.get("LOCAL_RANK", self.device)))
                self._backend = DistributedHandle(self.mesh.points, self.mesh.cells, labels, dist, rank, world, device)
            else:
                self._backend = _backend.Handle(self.mesh.points, self.mesh.cells, labels, device=self.device)
        h = self._backend
        t = self._material_tables(n_labels)
        h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
        opts = {k: v for k, v in self.solver_options.items() if k != 'mechanics'}
        h.set_options(dt=float(self.params.sim_time_step), **opts)
        self._upload_loads_and_bcs(h)
        h.setup(with_mechanics=mechanics)
        # the initial state is the initial-value function itself (simulation_base.py:253): Dirichlet data constrain the
        # unknown of each step, not u_previous
        h.set_state(u_previous.components[1], u_previous.components[0].reshape(-1) if mechanics else None)
        h.reset_stats()
        if hasattr(h, 'snapshot_clear'):
            h.snapshot_clear()
        self.solution = self.functionspace.new_function(name='solution_function')
        self.solution.label = 'solution_function'
        self.solution.assign(u_previous)
        self.solver = HipTimeStepSolver(self, h, mechanics)

    # -- parameter sweeps (the forward half of the reference's adjoint entry points) -------------------------------
    def run_for_adjoint(self, parameters, output_dir=config.output_dir_simulation_tmp):
        """:142-155 -- update (diffusion, proliferation, coupling) and re-run on the same mesh / space."""
        self.params.diffusion, self.params.proliferation, self.params.coupling = parameters
        self.run(keep_nth=1, save_method=None, clear_all=False, plot=False, output_dir=output_dir)
        return self.solution

    def run_for_adjoint2(self, parameters, output_dir=config.output_dir_simulation_tmp):
        """:157-170"""
        self.params.diffusion, self.params.proliferation = parameters
        self.run(keep_nth=1, save_method=None, clear_all=False, plot=False, output_dir=output_dir)
        return self.solution

    def init_postprocess(self, output_dir=config.output_dir_simulation_tmp):
        """:172-173 -- derived fields (stress, pressure, Jacobians ...) of the recorded solutions"""
        from ..simulation_helpers.postprocess import PostProcessTumorGrowth
        if self._backend is None:
            raise RuntimeError("init_postprocess needs a finished run() (the L2 projections run on its device backend)")
        labels = self._labels()
        t = self._material_tables(int(labels.max()) + 1)
        self.postprocess = PostProcessTumorGrowth(self.results, self.params, output_dir=output_dir,
                                                  backend=self._backend, tables=t, labels=labels)
        return self.postprocess

    def solver_statistics(self):
        return self._backend.stats() if self._backend is not None else {}
