"""
Derived fields of a finished simulation -- counterpart of PostProcess / PostProcessTumorGrowth /
PostProcessTumorGrowthBrain / Comparison in glimslib/simulation_helpers/helper_classes.py:1521-2036 (numeric
methods only; the matplotlib plotting methods are out of scope).

Every field of the reference is ``fenics.project(<UFL expression>, P1 space)``, i.e. the solution of
``M p = int expr * phi_i dx``.  Here the right-hand side is integrated with numpy (cell-wise constants exactly,
polynomial / nonlinear integrands with a degree-7 conical-product Gauss rule) and the mass solve runs on the device
(``glims_project``).  Tensor fields are returned as Functions with values of shape [N, d, d].
"""
from __future__ import annotations

import logging
import os

import numpy as np
from scipy.special import roots_jacobi

from ..fenics_local import Function, Constant
from . import math_linear_elasticity as mle, math_reaction_diffusion as mrd
from .helper_classes import DiscontinuousScalar


def simplex_quadrature(d, n=4):
    """
    Conical-product (Stroud) Gauss-Jacobi rule on the simplex: barycentric points [Q, d+1] and weights summing to 1,
    exact for polynomials of degree 2n-1.  Collapsed coordinates t_k in [0,1] with weight (1-t_k)^(d-1-k):
        lambda_1 = t_0,  lambda_2 = t_1 (1-t_0),  lambda_3 = t_2 (1-t_0)(1-t_1),  lambda_0 = prod (1-t_k).
    """
    nodes, weights = [], []
    for k in range(d):
        x, w = roots_jacobi(n, d - 1 - k, 0.0)
        nodes.append((x + 1.0) / 2.0)
        weights.append(w / 2.0 ** (d - k))
    grids = np.meshgrid(*nodes, indexing='ij')
    W = np.ones_like(grids[0])
    for k in range(d):
        shape = [1] * d
        shape[k] = n
        W = W * weights[k].reshape(shape)
    T = [g.ravel() for g in grids]
    lam = np.zeros((T[0].size, d + 1))
    rem = np.ones(T[0].size)
    for k in range(d):
        lam[:, k + 1] = T[k] * rem
        rem = rem * (1.0 - T[k])
    lam[:, 0] = rem
    w = W.ravel()
    return lam, w / w.sum()


class PostProcess:
    """helper_classes.py:1521-1732 (numeric part)."""

    def __init__(self, results, params, output_dir=None, plot_params=None, backend=None, tables=None, labels=None):
        self.logger = logging.getLogger(__name__)
        self._results = results
        self._params = params
        self._functionspace = results._functionspace
        self._subdomains = getattr(results, '_subdomains', None)
        self._mesh = self._functionspace._mesh
        self._backend = backend
        self._tables = tables
        self._labels = labels
        self.output_dir = output_dir
        self.plot_params = dict(plot_params or {})
        mesh = self._mesh
        self._vol = mesh.cell_volumes()
        X = mesh.points[mesh.cells]
        J = X[:, 1:, :] - X[:, :1, :]
        Jinv = np.linalg.inv(J)
        g = np.empty((len(J), mesh.dim + 1, mesh.dim))
        g[:, 1:, :] = np.transpose(Jinv, (0, 2, 1))
        g[:, 0, :] = -g[:, 1:, :].sum(axis=1)
        self._grad = g                                            # grad(lambda_a) per cell
        self._quad = simplex_quadrature(mesh.dim, 4)

    def set_output_dir(self, output_dir):
        self.output_dir = output_dir
        os.makedirs(output_dir, exist_ok=True)

    def get_output_dir(self):
        return self.output_dir

    # -- inputs -------------------------------------------------------------------------------------------------
    def get_solution_displacement(self, recording_step=None):
        return self._results.get_solution_function(subspace_name='displacement', recording_step=recording_step)

    def get_solution_concentration(self, recording_step=None):
        return self._results.get_solution_function(subspace_name='concentration', recording_step=recording_step)

    def _cell_values(self, name):
        """Per-cell value of a material: 'E', 'nu', 'rho', 'gamma', 'D' (tables from the simulation)."""
        if self._tables is not None and self._labels is not None:
            return np.asarray(self._tables[name], dtype=np.float64)[self._labels]
        pname = {'E': 'E', 'nu': 'poisson', 'rho': 'proliferation', 'gamma': 'coupling', 'D': 'diffusion'}[name]
        p = getattr(self._params, pname)
        if isinstance(p, DiscontinuousScalar):
            return p.cell_values()
        return np.full(self._mesh.num_cells(), float(p))

    def _cell_grad_u(self, recording_step=None):
        u = self.get_solution_displacement(recording_step).values()            # [N, d]
        return np.einsum('mad,mab->mbd', self._grad, u[self._mesh.cells])      # [M, b, d] = d u_b / d x_d

    # -- projections ------------------------------------------------------------------------------------------------
    def _solve_mass(self, rhs):
        if self._backend is None:
            raise RuntimeError("PostProcess needs the simulation's device backend for L2 projections "
                               "(create it through sim.init_postprocess)")
        return self._backend.project(rhs)

    def project_cell_field(self, q, name="f"):
        """L2 projection of a cell-wise constant field q [M, ...] onto P1."""
        q = np.asarray(q, dtype=np.float64)
        mesh = self._mesh
        d = mesh.dim
        tail = q.shape[1:]
        k = int(np.prod(tail)) if tail else 1
        loc = (q.reshape(len(q), k) * (self._vol / (d + 1))[:, None])
        rhs = np.zeros((mesh.num_vertices(), k))
        for a in range(d + 1):
            np.add.at(rhs, mesh.cells[:, a], loc)
        out = self._solve_mass(rhs if k > 1 else rhs[:, 0])
        return Function(mesh, {None: out.reshape((mesh.num_vertices(),) + tail)}, name=name)

    def project_pointwise(self, fn, nodal_fields, cell_fields=(), name="f"):
        """
        L2 projection of fn(*P1 fields evaluated at quadrature points, *cell fields) -> scalar.
        nodal_fields: arrays [N, ...] interpolated linearly inside each cell; cell_fields: arrays [M, ...].
        """
        mesh = self._mesh
        lam, w = self._quad
        cells = mesh.cells
        rhs = np.zeros(mesh.num_vertices())
        for q in range(len(w)):
            vals = [np.tensordot(lam[q], np.moveaxis(np.asarray(f)[cells], 1, 0), axes=(0, 0)) for f in nodal_fields]
            fq = fn(*vals, *cell_fields)                                       # [M]
            for a in range(mesh.dim + 1):
                np.add.at(rhs, cells[:, a], w[q] * lam[q, a] * self._vol * fq)
        return Function(mesh, {None: self._solve_mass(rhs)}, name=name)

    # -- fields ---------------------------------------------------------------------------------------------------
    def get_strain_tensor(self, recording_step=None):
        """:1566-1573"""
        return self.project_cell_field(mle.compute_strain(self._cell_grad_u(recording_step)), "strain_tensor")

    def get_stress_tensor(self, recording_step=None):
        """:1736-1744 -- 2 mu eps + lambda tr(eps) I with cell-wise mu, lambda"""
        E, nu = self._cell_values('E'), self._cell_values('nu')
        sig = mle.compute_stress(self._cell_grad_u(recording_step), mle.compute_mu(E, nu), mle.compute_lambda(E, nu))
        return self.project_cell_field(sig, "stress_tensor")

    def get_pressure(self, recording_step=None):
        """:1587-1593 -- 1/3 tr(stress) of the projected (P1) stress; projecting a P1 field again is the identity"""
        s = self.get_stress_tensor(recording_step).values()
        return Function(self._mesh, {None: mle.compute_pressure_from_stress_tensor(s)}, name="pressure")

    def get_van_mises_stress(self, recording_step=None):
        """:1595-1601"""
        s = self.get_stress_tensor(recording_step).values()
        d = self._mesh.dim
        return self.project_pointwise(lambda sq: mle.compute_van_mises_stress(sq, d), [s], name="van_mises_stress")

    def get_displacement_norm(self, recording_step=None):
        """:1612-1618"""
        u = self.get_solution_displacement(recording_step).values()
        return self.project_pointwise(lambda uq: np.sqrt((uq * uq).sum(axis=-1)), [u], name="displacement_norm")

    def get_logistic_growth(self, recording_step=None):
        """:1746-1752"""
        c = self.get_solution_concentration(recording_step).values()
        rho = self._cell_values('rho')
        return self.project_pointwise(lambda cq, r: mrd.compute_growth_logistic(cq, r, 1.0), [c], [rho], "log_growth")

    def get_mech_expansion(self, recording_step=None):
        """:1754-1761 -- c * coupling * I"""
        c = self.get_solution_concentration(recording_step).values()
        d = self._mesh.dim
        gam = self._cell_values('gamma')
        if np.all(gam == gam[0]):
            return Function(self._mesh, {None: mle.compute_growth_induced_strain(c, gam[0], d)}, name="mech_expansion")
        diag = self.project_pointwise(lambda cq, g: cq * g, [c], [gam]).values()
        return Function(self._mesh, {None: diag[:, None, None] * np.eye(d)}, name="mech_expansion")

    def get_total_jacobian(self, recording_step=None):
        """:1763-1769 -- det(I + grad u), constant per cell"""
        return self.project_cell_field(mle.compute_total_jacobian(self._cell_grad_u(recording_step)), "total_jacobian")

    def get_growth_induced_jacobian(self, recording_step=None):
        """:1771-1777 -- det(I + P1 growth strain)"""
        sg = self.get_mech_expansion(recording_step).values()
        d = self._mesh.dim
        return self.project_pointwise(lambda s: mle.compute_growth_induced_jacobian(s, d), [sg],
                                      name="growth_induced_jacobian")

    def get_concentration_deformed_configuration(self, recording_step=None):
        """:1779-1786 -- c * J_growth / J_total (math_linear_elasticity.py:66-70)"""
        c = self.get_solution_concentration(recording_step).values()
        d = self._mesh.dim
        gam = self._cell_values('gamma')
        jt = mle.compute_total_jacobian(self._cell_grad_u(recording_step))
        return self.project_pointwise(lambda cq, g, j: cq * (1.0 + g * cq) ** d / j, [c], [gam, jt],
                                      name="concentration_deformed_config")

    def compute_force(self, recording_step=None, subdomain_id=None):
        """:1603-1610 -- oint sigma.n ds over the exterior facets (optionally only those with one interface id)"""
        s = self.get_stress_tensor(recording_step).values()
        mesh = self._mesh
        f = mesh.facets()
        mask = f['exterior'].copy()
        if subdomain_id is not None:
            mask &= self._subdomains.subdomain_boundaries.array() == subdomain_id
        verts = f['vertices'][mask]
        X = mesh.points[verts]
        d = mesh.dim
        if d == 2:
            t = X[:, 1] - X[:, 0]
            nrm = np.stack([t[:, 1], -t[:, 0]], axis=1)                          # |n| = facet length
        else:
            nrm = 0.5 * np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0])           # |n| = facet area
        inside = mesh.points[mesh.cells[f['cell0'][mask]]].mean(axis=1) - X.mean(axis=1)
        nrm *= np.where((nrm * inside).sum(axis=1) > 0, -1.0, 1.0)[:, None]      # outward
        smean = s[verts].mean(axis=1)                                            # exact facet integral of a P1 field
        return list(np.einsum('fab,fb->a', smean, nrm))

    def save_all(self, save_method='vtk', clear_all=False, selection=slice(None), output_dir=None):
        """:1922-1943 -- writes the derived fields of the selected recording steps as .vtu files"""
        from ..utils.vtu_io import write_vtu
        out = output_dir or self.output_dir
        os.makedirs(out, exist_ok=True)
        written = []
        for step in self._results.get_recording_steps()[selection]:
            fields = {'concentration': self.get_solution_concentration(step).values(),
                      'displacement': self.get_solution_displacement(step).values(),
                      'pressure': self.get_pressure(step).values(),
                      'van_mises_stress': self.get_van_mises_stress(step).values(),
                      'total_jacobian': self.get_total_jacobian(step).values(),
                      'growth_induced_jacobian': self.get_growth_induced_jacobian(step).values(),
                      'concentration_deformed_config': self.get_concentration_deformed_configuration(step).values(),
                      'log_growth': self.get_logistic_growth(step).values()}
            path = os.path.join(out, "postprocess_%05d.vtu" % step)
            write_vtu(path, self._mesh.points, self._mesh.cells, fields,
                      {'label_map': self._labels} if self._labels is not None else None)
            written.append(path)
        return written


class PostProcessTumorGrowth(PostProcess):
    """helper_classes.py:1734-1972"""


class PostProcessTumorGrowthBrain(PostProcessTumorGrowth):
    """helper_classes.py:1945-1972: ``map_params`` turns the per-tissue scalars into cell-wise coefficients; here the
    simulation hands over its per-label tables, so there is nothing left to map."""

    def map_params(self):
        return None


class Comparison:
    """helper_classes.py:1975-2036 -- errornorms between two simulations' recorded solutions (same mesh)."""

    def __init__(self, sim1, sim2):
        self.sim1, self.sim2 = sim1, sim2
        self.steps = sorted(set(sim1.results.get_recording_steps()) & set(sim2.results.get_recording_steps()))

    def _mass_norm(self, e):
        h = self.sim1._backend
        e = np.asarray(e, dtype=np.float64)
        if e.ndim == 1:
            return float(np.sqrt(max(0.0, e @ h.apply(2, e)[0])))
        return float(np.sqrt(max(0.0, sum(e[:, a] @ h.apply(2, e[:, a])[0] for a in range(e.shape[1])))))

    def compute_errornorm_by_subspace(self, recording_step):
        """L2 errornorm per subspace = sqrt(e^T M e) for P1 data."""
        out = {}
        for name in ('displacement', 'concentration'):
            a = self.sim1.results.get_solution_function(subspace_name=name, recording_step=recording_step).values()
            b = self.sim2.results.get_solution_function(subspace_name=name, recording_step=recording_step).values()
            out[name] = self._mass_norm(a - b)
        return out

    def compute_max_difference(self, recording_step):
        out = {}
        for name in ('displacement', 'concentration'):
            a = self.sim1.results.get_solution_function(subspace_name=name, recording_step=recording_step).values()
            b = self.sim2.results.get_solution_function(subspace_name=name, recording_step=recording_step).values()
            out[name] = float(np.abs(a - b).max())
        return out

    def compare(self, selection=slice(None)):
        """Table (list of dicts; a pandas DataFrame if pandas is importable) of errornorms per recording step."""
        rows = []
        for step in self.steps[selection]:
            e, m = self.compute_errornorm_by_subspace(step), self.compute_max_difference(step)
            rows.append({'recording_step': step, 'errornorm_displacement': e['displacement'],
                         'errornorm_concentration': e['concentration'], 'max_diff_displacement': m['displacement'],
                         'max_diff_concentration': m['concentration']})
        try:
            import pandas as pd
            return pd.DataFrame(rows)
        except Exception:   # pragma: no cover
            return rows
