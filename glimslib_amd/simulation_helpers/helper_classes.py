"""
Host-side helper classes of the forward path, same names / semantics as the reference's
glimslib/simulation_helpers/helper_classes.py, on numpy arrays instead of DOLFIN objects.

Only the pieces that ``FenicsSimulation.run()`` and its setup touch are here (SURVEY.md section 8a rows
a7-a13): SubDomains, FunctionSpace/SubSpaces, BoundaryConditions, Parameters, TimeSeries*, Results.
Plotting / PostProcess / Comparison are out of scope (post-hoc analysis).

Known deviations from the reference text, all deliberate and tested:
  * dict-order dependence.  The reference iterates ``tissue_id_name_map`` in dict order
    (helper_classes.py:469-473, 567-573).  It ran on Python 3.5 (FEniCS 2017.2 images), where small-int keys
    iterate in ascending order whatever the literal's order, so a map written {1:'CSF', 3:'WM', 2:'GM', 4:...}
    (test_case_comparison_3D_atlas.py:46-49) behaved as if sorted.  On Python >= 3.7 the same code would silently
    swap WM/GM parameters and miss interface ids.  This build sorts by tissue id = the behaviour
    the reference's results were produced with.
  * unknown BC keys ('boundary_name', 'boundary_id') are skipped like in the reference (helper_classes.py:718-721)
    but logged at WARNING with the accepted spellings.
  * ``dirichlet_bcs=None`` / ``{}`` mean "no Dirichlet BCs" instead of TypeError / missing attribute (q5).
"""
from __future__ import annotations

import copy
import itertools
import logging
import os
import shutil

import numpy as np

from .. import fenics_local as fenics
from ..fenics_local import Function, Constant, Expression, SubDomain, CellFunction, interpolate_nodal


class DiscontinuousScalar:
    """
    Cell-wise constant coefficient: value = coeffs[cell_function[cell]] (helper_classes.py:47-58).
    ``coeffs`` maps tissue id -> float.
    """

    def __init__(self, cell_function, scalars, **kwargs):
        self.cell_function = np.asarray(cell_function.array() if hasattr(cell_function, 'array') else cell_function)
        self.coeffs = dict(scalars) if isinstance(scalars, dict) else {i: s for i, s in enumerate(scalars)}

    def cell_values(self):
        table = self.table(int(self.cell_function.max()) + 1)
        return table[self.cell_function]

    def table(self, n_labels):
        t = np.zeros(n_labels)
        for k, v in self.coeffs.items():
            if 0 <= k < n_labels:
                t[k] = float(v.value) if isinstance(v, Constant) else float(v)
        return t


class Boundary(SubDomain):
    """helper_classes.py:61-63"""

    def inside(self, x, on_boundary):
        return on_boundary

    def inside_vectorized(self, X, on_boundary):
        return np.asarray(on_boundary, dtype=bool)


class _FacetFunction:
    def __init__(self, values):
        self._a = np.asarray(values, dtype=np.int64)

    def array(self):
        return self._a


class Measure:
    """Carrier of (kind, subdomain_data, id) -- the numeric backend only needs to know *where* to integrate."""

    def __init__(self, kind, data=None, subdomain_id=None):
        self.kind = kind
        self._data = data
        self.subdomain_id_ = subdomain_id

    def subdomain_data(self):
        return self._data

    def __call__(self, subdomain_id):
        return Measure(self.kind, self._data, subdomain_id)


# ---------------------------------------------------------------------------------------------------
class SubSpaces:
    """
    helper_classes.py:66-232.  Per-subspace bookkeeping: names plus optional per-subspace attributes (elements,
    function spaces, initial-value expressions, Dirichlet / Neumann BC lists), each stored as
    ``{subspace_id: item}`` under ``_<attribute>``; lists are enumerated, an existing attribute is only replaced
    when asked to, wrong lengths / types are logged and ignored -- all as in the reference.
    """

    def __init__(self, names=None):
        self.logger = logging.getLogger(__name__)
        self.names = dict(names or {})
        self.n = len(self.names)
        self._attribute_prefix = '_'

    def get_subspace_names(self):
        return self.names.values()

    def get_subspace_name(self, subspace_id):
        return self.names.get(subspace_id)

    def get_subspace_ids(self):
        return self.names.keys()

    def get_subspace_id(self, subspace_name):
        inv = {v: k for k, v in self.names.items()}
        if subspace_name in inv:
            return inv[subspace_name]
        self.logger.warning("Functionspace does not have '%s' subspace." % subspace_name)
        return None

    # -- generic attribute store ---------------------------------------------------------------------------
    def _set_subspace_attribute(self, name, content, replace=False):
        internal = self._attribute_prefix + name
        if not isinstance(content, (list, dict)):
            self.logger.error('Expect either list or dictionary')
            return
        if len(content) != self.n:
            self.logger.error('Expect content with %i items, but this has %i items.' % (self.n, len(content)))
            return
        as_dict = dict(enumerate(content)) if isinstance(content, list) else content
        if hasattr(self, internal) and not replace:
            self.logger.warning("Attribute '%s' already exists ... do nothing." % internal)
            return
        setattr(self, internal, as_dict)

    def _get_subspace_attribute(self, name, subspace_id=None, subspace_name=None):
        if subspace_id is None and subspace_name is None:
            self.logger.error("No subspace or subspace name specified")
        elif subspace_id is None:
            subspace_id = self.get_subspace_id(subspace_name)
        store = getattr(self, self._attribute_prefix + name, None)
        if store is None:
            self.logger.warning("Attribute '%s' does not exist." % name)
            return None
        if subspace_id in store:
            return store[subspace_id]
        self.logger.warning("Attribute '%s' has no information for subspace '%s'" % (name, subspace_id))
        return None

    def get_element(self, subspace_id=None, subspace_name=None):
        return self._get_subspace_attribute('elements', subspace_id, subspace_name)

    def get_inital_value_expression(self, subspace_id=None, subspace_name=None):
        return self._get_subspace_attribute('inital_value_expressions', subspace_id, subspace_name)

    def get_functionspace(self, subspace_id=None, subspace_name=None):
        return self._get_subspace_attribute('functionspaces', subspace_id, subspace_name)

    def get_dirichlet_bcs(self, subspace_id=None, subspace_name=None):
        return self._get_subspace_attribute('bcs_dirichlet', subspace_id, subspace_name)

    def get_von_neumann_bcs(self, subspace_id=None, subspace_name=None):
        return self._get_subspace_attribute('bcs_von_neumann', subspace_id, subspace_name)

    def set_elements(self, content, replace=False):
        self._set_subspace_attribute('elements', content, replace=replace)

    def set_inital_value_expressions(self, content, replace=False):
        self._set_subspace_attribute('inital_value_expressions', content, replace=replace)

    def set_functionspaces(self, content, replace=False):
        self._set_subspace_attribute('functionspaces', content, replace=replace)

    def _rearrange_dict_by_subspace(self, dict_in):
        """{bc name: {..., 'subspace_id': i}} -> {i: [bc dict + 'name']}, an (empty) list for every subspace."""
        by_subspace = {sid: [] for sid in self.names}
        for bc_name, item in dict_in.items():
            sid = item.get('subspace_id')
            if sid in by_subspace:
                item['name'] = bc_name
                by_subspace[sid].append(item)
        return by_subspace

    def set_dirichlet_bcs(self, content, replace=False):
        self._set_subspace_attribute('bcs_dirichlet', self._rearrange_dict_by_subspace(content), replace=replace)

    def set_von_neumann_bcs(self, content, replace=False):
        self._set_subspace_attribute('bcs_von_neumann', self._rearrange_dict_by_subspace(content), replace=replace)

    def project_over_subspace(self, function_expr, subspace_id=None, subspace_name=None, **kwargs):
        if subspace_id is None and subspace_name is None:
            self.logger.error("No subspace or subspace name specified")
            return None
        if subspace_id is None:
            subspace_id = self.get_subspace_id(subspace_name)
        space = self.get_functionspace(subspace_id=subspace_id)
        if space is None:
            return None
        try:
            return space.project_over_space(function_expr)
        except Exception as e:   # noqa: BLE001 -- the reference logs and returns None
            self.logger.warning("Cannot project functions over subspace %s: %r" % (subspace_id, e))
            return None


class _CollapsedSpace:
    """Stand-alone space of one subspace of a mixed FunctionSpace (the reference builds
    ``fenics.FunctionSpace(mesh, sub_element)`` per subspace, helper_classes.py:258-269)."""

    def __init__(self, mesh, element, value_size, name):
        self._mesh, self.element, self._vs, self.name = mesh, element, value_size, name

    def ufl_element(self):
        return self.element

    def project_over_space(self, function_expr):
        return Function(self._mesh, {None: interpolate_nodal(function_expr, self._mesh, self._vs)}, name=self.name,
                        space=self)


class FunctionSpace:
    """
    helper_classes.py:234-383.  ``element`` is either the reference's own description -- a
    ``fenics.MixedElement([VectorElement, FiniteElement])`` / a single element (fenics_local supplies P1 descriptors)
    -- or the short form this package uses internally, ``{subspace_id: value_size}`` / an int.  The mixed
    tumour-growth space is {0: dim, 1: 1}.
    """

    def __init__(self, mesh, projection_parameters=None):
        self.logger = logging.getLogger(__name__)
        self._mesh = mesh
        self.dim_geo = mesh.geometric_dimension()
        self._projection_parameters = projection_parameters or {}
        self.has_subspaces = False

    def init_function_space(self, element, name):
        self.element = element
        dim = self.dim_geo
        if isinstance(name, dict):
            self.has_subspaces = True
            self.subspaces = SubSpaces(name)
            if isinstance(element, dict):
                self.value_sizes = {k: int(v) for k, v in element.items()}
                sub_elements = dict(element)
            else:
                subs = element.sub_elements()
                self.value_sizes = {i: int(e.value_size(dim)) for i, e in enumerate(subs)}
                sub_elements = dict(enumerate(subs))
            self.subspaces.set_elements(sub_elements)
            self.subspaces.set_functionspaces({i: _CollapsedSpace(self._mesh, sub_elements[i], vs, name.get(i))
                                               for i, vs in self.value_sizes.items()})
        else:
            self.has_subspaces = False
            self.name = name
            vs = int(element) if isinstance(element, (int, np.integer)) else int(element.value_size(dim))
            self.value_sizes = {None: vs}
        self.function_space = self

    # -- accessors of the reference ---------------------------------------------------------------------------
    def get_element(self, subspace_id=None, subspace_name=None):
        if self.has_subspaces and not (subspace_id is None and subspace_name is None):
            return self.subspaces.get_element(subspace_id=subspace_id, subspace_name=subspace_name)
        return self.element

    def get_functionspace(self, subspace_id=None, subspace_name=None):
        if self.has_subspaces and not (subspace_id is None and subspace_name is None):
            return self.subspaces.get_functionspace(subspace_id=subspace_id, subspace_name=subspace_name)
        return self.function_space

    def get_functionspace_orig_subspace(self, subspace_id=None, subspace_name=None):
        if subspace_id is None and subspace_name is None:
            self.logger.error("No subspace or subspace name specified")
            return None
        if subspace_id is None:
            subspace_id = self.get_subspace_id(subspace_name)
        return (self, subspace_id)

    def get_subspace_id(self, subspace_name):
        return self.subspaces.get_subspace_id(subspace_name)

    def value_size(self, subspace_id=None):
        return self.value_sizes[subspace_id]

    def new_function(self, name="f"):
        n = self._mesh.num_vertices()
        comps = {k: (np.zeros(n) if vs == 1 else np.zeros((n, vs))) for k, vs in self.value_sizes.items()}
        return Function(self._mesh, comps, names=getattr(self, 'subspaces', SubSpaces()).names, name=name, space=self)

    def project_over_space(self, function_expr, subspace_id=None, subspace_name=None, **kwargs):
        """
        helper_classes.py:332-360.  Nodal interpolation (the reference L2-projects with CG+AMG at KSP rtol 1e-6;
        for P1 data the two agree to that tolerance -- SURVEY.md section 7.3 (i)).
        """
        if self.has_subspaces and isinstance(function_expr, dict):
            f = self.new_function()
            for key, expr in function_expr.items():
                sid = self.subspaces.get_subspace_id(key) if isinstance(key, str) else key
                f.components[sid] = interpolate_nodal(expr, self._mesh, self.value_sizes[sid])
            return f
        if self.has_subspaces and subspace_id is None and subspace_name is not None:
            subspace_id = self.subspaces.get_subspace_id(subspace_name)
        if isinstance(function_expr, Function):
            if subspace_id is not None and subspace_id in function_expr.components:
                return function_expr.sub(subspace_id)
            return function_expr.copy()
        vs = self.value_sizes[subspace_id] if subspace_id in self.value_sizes else 1
        space = self.subspaces.get_functionspace(subspace_id) if self.has_subspaces and subspace_id is not None else self
        return Function(self._mesh, {None: interpolate_nodal(function_expr, self._mesh, vs)}, space=space)

    def split_function(self, function, subspace_id=None, subspace_name=None):
        """helper_classes.py:362-383"""
        if self.has_subspaces:
            if subspace_id is None and subspace_name is None:
                return function
            if subspace_id is None:
                subspace_id = self.subspaces.get_subspace_id(subspace_name)
            sub = function.sub(subspace_id)
            sub._space = self.subspaces.get_functionspace(subspace_id)
            return sub
        return function


# ---------------------------------------------------------------------------------------------------
class SubDomains:
    """helper_classes.py:385-615"""

    def __init__(self, mesh):
        self.logger = logging.getLogger(__name__)
        self._mesh = mesh
        self.dim_geo = mesh.geometric_dimension()

    # -- cell labels ------------------------------------------------------------------------------------
    def setup_subdomains(self, label_function=None, subdomains=None, replace=False):
        """
        helper_classes.py:402-444: (a) given cell function, (b) from a label function through
        ``int(label(cell.midpoint()))``, (c) all zero.
        """
        if hasattr(self, 'subdomains') and not replace:
            self.logger.warning("'subdomains' already exists. ... do nothing.")
            return
        if hasattr(self, 'subdomains'):
            self.logger.warning("... replacing existing 'subdomains'.")
        if subdomains is not None:
            arr = subdomains.array() if hasattr(subdomains, 'array') else subdomains
            arr = np.asarray(arr, dtype=np.int64)
            if arr.shape != (self._mesh.num_cells(),):
                raise ValueError("subdomains must hold one integer per cell")
            self.subdomains = CellFunction(self._mesh, arr)
        elif label_function is not None:
            self._setup_subdomains_from_labelmapfunction(label_function)
        else:
            self.subdomains = CellFunction(self._mesh)
            self.subdomains.set_all(0)

    def _setup_subdomains_from_labelmapfunction(self, label_function):
        """
        ``int(label_function(cell.midpoint()))`` per cell (helper_classes.py:441-442).  A P1 / DG1 function equals the
        mean of the cell's vertex values at the midpoint: nodal array [N] or Function (P1), array [M, d+1] or
        DG1Function (the reference's scripts and unit tests pass ``fenics.project(expr, DG1)``).  A raw Expression or
        callable is evaluated AT the midpoint itself, as the reference's call does -- for a step function such as
        '(x[0]>=0) ? 1 : 2' the two rules differ by the cell layer that touches the step.
        """
        self.label_function = label_function
        cells = self._mesh.cells
        if isinstance(label_function, Function):
            vals = label_function.values()[cells]
        elif isinstance(label_function, fenics.DG1Function):
            vals = label_function.cell_vertex_values
        elif isinstance(label_function, (Expression, Constant)) or callable(label_function):
            mid = np.asarray(label_function(self._mesh.cell_midpoints()), dtype=np.float64).reshape(-1)
            if mid.shape != (len(cells),):
                raise ValueError("label function has the wrong shape")
            mid = np.where(np.abs(mid - np.round(mid)) < 1e-9, np.round(mid), mid)
            self.subdomains = CellFunction(self._mesh, mid.astype(np.int64))
            return
        else:
            a = np.asarray(label_function, dtype=np.float64)
            vals = a[cells] if a.shape == (self._mesh.num_vertices(),) else a
        if vals.shape != cells.shape:
            raise ValueError("label function has the wrong shape")
        mid = vals.mean(axis=1)
        # guard the truncation against round-off in the mean of equal integers (2.0 must stay 2)
        mid = np.where(np.abs(mid - np.round(mid)) < 1e-9, np.round(mid), mid)
        self.subdomains = CellFunction(self._mesh, mid.astype(np.int64))

    # -- facets -----------------------------------------------------------------------------------------
    def setup_boundaries(self, tissue_map=None, boundary_fct_dict=None):
        if tissue_map is not None:
            self._setup_boundaries_from_subdomains(tissue_map)
        if boundary_fct_dict is not None:
            self._setup_boundaries_from_functions(boundary_fct_dict)

    def _setup_boundaries_from_subdomains(self, tissue_id_name_map):
        """
        helper_classes.py:457-501: interface id of every facet between two different tissues; ids enumerate
        ``itertools.combinations`` of the tissue ids (ascending, see module docstring), 'no_boundary' = max + 1 for
        every other facet.
        """
        if not hasattr(self, 'subdomains'):
            self.logger.warning("Need subdomains to define boundaries. No subdomains defined.")
            return
        self.tissue_id_name_map = dict(sorted(tissue_id_name_map.items()))
        ids = list(self.tissue_id_name_map.keys())
        names = list(self.tissue_id_name_map.values())
        boundary_types = list(itertools.combinations(ids, 2))
        boundary_names_string = ['_'.join(p) for p in itertools.combinations(names, 2)]
        boundary_type_dict = dict(zip(boundary_types, boundary_names_string))
        boundary_id_dict = dict(zip(boundary_names_string, range(len(boundary_type_dict))))
        value_no_boundary = (max(boundary_id_dict.values()) + 1) if boundary_id_dict else 0
        f = self._mesh.facets()
        lab = self.subdomains.array()
        l0 = lab[f['cell0']]
        l1 = np.where(f['cell1'] >= 0, lab[np.maximum(f['cell1'], 0)], l0)
        lo, hi = np.minimum(l0, l1), np.maximum(l0, l1)
        facet_ids = np.full(len(l0), value_no_boundary, dtype=np.int64)
        for (a, b), name in boundary_type_dict.items():
            facet_ids[(lo == a) & (hi == b) & (lo != hi)] = boundary_id_dict[name]
        boundary_id_dict['no_boundary'] = value_no_boundary
        self.subdomain_boundaries = _FacetFunction(facet_ids)
        self.subdomain_boundaries_id_dict = boundary_id_dict
        self.logger.info("     ... found boundaries %s" % (np.unique(facet_ids)))

    def _setup_boundaries_from_functions(self, boundary_dict):
        """
        helper_classes.py:503-528: ids 1..k in dict order; ``SubDomain.mark`` = facet marked when ``inside`` holds at
        all its vertices and its midpoint, with on_boundary = "facet is exterior".  Later entries overwrite earlier.
        Besides SubDomain instances a boolean facet mask or a vectorised callable (X, on_boundary) -> bool is accepted.
        """
        f = self._mesh.facets()
        verts, ext = f['vertices'], f['exterior']
        nf, k = verts.shape
        pts = self._mesh.points
        values = np.zeros(nf, dtype=np.int64)
        boundary_id_dict = {}
        for boundary_id, (name, bdef) in enumerate(boundary_dict.items(), start=1):
            if isinstance(bdef, np.ndarray) and bdef.dtype == bool and bdef.shape == (nf,):
                mask = bdef
            else:
                fn = bdef.inside_vectorized if isinstance(bdef, SubDomain) else bdef
                # successive tests only on the facets that passed the previous ones: after the first vertex a typical
                # `on_boundary and ...` subdomain has ~1 % of the facets left (1.0 s -> 0.2 s at 12 M facets)
                idx = np.arange(nf)
                for j in range(k):
                    idx = idx[np.asarray(fn(pts[verts[idx, j]], ext[idx]), dtype=bool)]
                idx = idx[np.asarray(fn(pts[verts[idx]].mean(axis=1), ext[idx]), dtype=bool)]
                mask = np.zeros(nf, dtype=bool)
                mask[idx] = True
            values[mask] = boundary_id
            boundary_id_dict[name] = boundary_id
        self.named_boundaries_id_dict = boundary_id_dict
        self.named_boundaries_function_dict = boundary_dict
        self.named_boundaries = _FacetFunction(values)

    def setup_measures(self):
        """helper_classes.py:539-562"""
        self.dx = Measure('dx', getattr(self, 'subdomains', None))
        self.ds = Measure('ds', getattr(self, 'subdomain_boundaries', None))
        self.dsn = Measure('ds', getattr(self, 'named_boundaries', None))

    # -- per-tissue parameters ----------------------------------------------------------------------------
    def create_discontinuous_scalar_from_parameter_map(self, param_dict, name, replace=False):
        """helper_classes.py:564-603; the value of a tissue id absent from the map is 0 (the leading Constant(0))."""
        if not hasattr(self, 'tissue_id_name_map'):
            self.logger.warning("No subdomains have been defined, cannot assign parameter values")
            return None
        if hasattr(self, name) and not replace:
            self.logger.warning("Parameter '%s' already exists. ... do nothing." % name)
            return None
        coeffs = {tid: float(param_dict[tname]) for tid, tname in self.tissue_id_name_map.items()}
        disc = DiscontinuousScalar(self.subdomains, coeffs)
        setattr(self, name, disc)
        return disc

    def get_subdomain_id(self, subdomain_name):
        """helper_classes.py:609-615"""
        if not hasattr(self, 'tissue_name_id_map'):
            self.tissue_name_id_map = {v: k for k, v in getattr(self, 'tissue_id_name_map', {}).items()}
        if subdomain_name in self.tissue_name_id_map:
            return self.tissue_name_id_map[subdomain_name]
        self.logger.error("Subdomain '%s' does not exist" % subdomain_name)
        return None


# ---------------------------------------------------------------------------------------------------
class DirichletBC:
    """Resolved Dirichlet condition: the nodes of the selected facets, the subspace, the value object."""

    def __init__(self, subspace_id, value, nodes):
        self.subspace_id = subspace_id
        self.value = value
        self.nodes = np.asarray(nodes, dtype=np.int64)


class BoundaryConditions:
    """helper_classes.py:618-908"""

    def __init__(self, functionspace, subdomains):
        self.logger = logging.getLogger(__name__)
        self._functionspace = functionspace
        self._subdomains = subdomains
        self.dirichlet_bcs = []

    # -- Dirichlet --------------------------------------------------------------------------------------
    def setup_dirichlet_boundary_conditions(self, dirichlet_bcs=None):
        self.dirichlet_bcs = []
        if not dirichlet_bcs:
            return
        self.dirichlet_bcs_dict = dirichlet_bcs
        for bc_name, bc_dict in dirichlet_bcs.items():
            bc = self._construct_dirichlet_bc(bc_dict, bc_name)
            if bc is not None:
                self.dirichlet_bcs.append(bc)

    def _facet_nodes(self, mask):
        f = self._functionspace._mesh.facets()
        return np.unique(f['vertices'][mask])

    def _construct_dirichlet_bc(self, dirichlet_bc, bc_name=""):
        """helper_classes.py:673-723 -- accepted keys: 'boundary' | 'subdomain_boundary' | 'named_boundary'."""
        subspace_id = dirichlet_bc.get('subspace_id') if self._functionspace.has_subspaces else None
        if self._functionspace.has_subspaces and subspace_id is None:
            self.logger.error("Dirichlet BC dictionary does not contain id of function sub space 'subspace_id'")
            return None
        if 'bc_value' not in dirichlet_bc:
            self.logger.error("Dirichlet BC dictionary does not contain BC value 'bc_value'")
            return None
        value = dirichlet_bc['bc_value']
        mesh = self._functionspace._mesh
        f = mesh.facets()
        if 'boundary' in dirichlet_bc:
            b = dirichlet_bc['boundary']
            fn = b.inside_vectorized if isinstance(b, SubDomain) else b
            verts, ext = f['vertices'], f['exterior']
            mask = np.ones(len(ext), dtype=bool)
            for j in range(verts.shape[1]):
                mask &= np.asarray(fn(mesh.points[verts[:, j]], ext), dtype=bool)
            mask &= np.asarray(fn(mesh.points[verts].mean(axis=1), ext), dtype=bool)
            return DirichletBC(subspace_id, value, self._facet_nodes(mask))
        if 'subdomain_boundary' in dirichlet_bc:
            name = dirichlet_bc['subdomain_boundary']
            ids = getattr(self._subdomains, 'subdomain_boundaries_id_dict', {})
            if name in ids:
                mask = self._subdomains.subdomain_boundaries.array() == ids[name]
                return DirichletBC(subspace_id, value, self._facet_nodes(mask))
            self.logger.warning("       - Dirichlet BC '%s': unknown subdomain boundary '%s' -- skipping" % (bc_name, name))
            return None
        if 'named_boundary' in dirichlet_bc:
            bid = getattr(self._subdomains, 'named_boundaries_id_dict', {}).get(dirichlet_bc['named_boundary'])
            if bid is not None:
                mask = self._subdomains.named_boundaries.array() == bid
                return DirichletBC(subspace_id, value, self._facet_nodes(mask))
            self.logger.warning("       - Dirichlet BC '%s': unknown named boundary -- skipping" % bc_name)
            return None
        self.logger.warning("       - Dirichlet BC '%s' incomplete -- skipping (accepted keys: 'boundary', "
                            "'subdomain_boundary', 'named_boundary'; got %s)" % (bc_name, sorted(dirichlet_bc)))
        return None

    def dirichlet_dofs(self, subspace_id):
        """(dof indices, values) for one subspace; vector subspaces use node*dim + component; later BCs win."""
        vs = self._functionspace.value_sizes[subspace_id]
        mesh = self._functionspace._mesh
        table = {}
        for bc in self.dirichlet_bcs:
            if bc.subspace_id != subspace_id or len(bc.nodes) == 0:
                continue
            vals = interpolate_nodal(bc.value, _SubMesh(mesh, bc.nodes), vs)
            if vs == 1:
                for n, v in zip(bc.nodes, vals):
                    table[int(n)] = float(v)
            else:
                for n, v in zip(bc.nodes, vals):
                    for a in range(vs):
                        table[int(n) * vs + a] = float(v[a])
        if not table:
            return np.zeros(0, dtype=np.int64), np.zeros(0)
        keys = np.fromiter(table.keys(), dtype=np.int64, count=len(table))
        order = np.argsort(keys)
        return keys[order], np.fromiter(table.values(), dtype=np.float64, count=len(table))[order]

    # -- von Neumann --------------------------------------------------------------------------------------
    def setup_von_neumann_boundary_conditions(self, von_neumann_bcs=None):
        """helper_classes.py:725-837"""
        self.von_neumann_bcs = {}
        if not von_neumann_bcs:
            return
        self.von_neumann_bcs_dict = von_neumann_bcs
        for bc_name, bc_dict in von_neumann_bcs.items():
            spec = self._construct_von_neumann_bc(bc_dict, bc_name)
            if spec is not None:
                self.von_neumann_bcs[bc_name] = spec

    def _construct_von_neumann_bc(self, bc_dict, bc_name=""):
        bc_value = bc_dict.get('bc_value')
        if bc_value is None:
            self.logger.error("Von Neumann BC dictionary does not contain BC value, key 'bc_value'")
        subspace_id = bc_dict.get('subspace_id') if self._functionspace.has_subspaces else None
        if self._functionspace.has_subspaces and subspace_id is None:
            self.logger.error("Von Neumann BC dictionary does not contain id of function subspace 'subspace_id'")
        measure = None
        if 'boundary' in bc_dict:
            self.logger.error("You specified a function based boundary that has not been set up. The current "
                              "implemention requires such boundaries to be defined as 'named boundaries' upon "
                              "initialisation.")
        elif 'subdomain_boundary' in bc_dict:
            ids = getattr(self._subdomains, 'subdomain_boundaries_id_dict', {})
            if bc_dict['subdomain_boundary'] in ids:
                measure = self._subdomains.ds(ids[bc_dict['subdomain_boundary']])
        elif 'named_boundary' in bc_dict:
            bid = getattr(self._subdomains, 'named_boundaries_id_dict', {}).get(bc_dict['named_boundary'])
            if bid is not None:
                measure = self._subdomains.dsn(bid)
        else:
            self.logger.warning("       - Von Neumann BC '%s' incomplete -- skipping (accepted keys: "
                                "'subdomain_boundary', 'named_boundary'; got %s)" % (bc_name, sorted(bc_dict)))
        needs_sub = self._functionspace.has_subspaces
        if bc_value is not None and measure is not None and (subspace_id is not None or not needs_sub):
            return {'bc_value': bc_value, 'measure': measure, 'subspace_id': subspace_id}
        return None

    def time_update_bcs(self, time, kind='dirichlet'):
        """helper_classes.py:839-859"""
        d = getattr(self, 'dirichlet_bcs_dict' if kind == 'dirichlet' else 'von_neumann_bcs_dict', {})
        for bc_name, bc in d.items():
            try:
                bc['bc_value'].t = time
            except Exception:
                self.logger.debug("Updating expression for %s BC '%s' at time %.2f raised exception" % (kind, bc_name, time))

    def is_time_dependent(self):
        for attr in ('dirichlet_bcs_dict', 'von_neumann_bcs_dict'):
            for bc in getattr(self, attr, {}).values():
                if hasattr(bc.get('bc_value'), 't'):
                    return True
        return False

    def implement_von_neumann_bc(self, coefficient_per_cell=None, subspace_id=None):
        """
        Numeric counterpart of helper_classes.py:861-908: the load vector
            sum_bc  oint_{ds(id)}  g_bc * coef * phi_i  ds      (scalar subspace)
            sum_bc  oint_{ds(id)}  g_bc . (phi_i e_a)   ds      (vector subspace, dof node*dim + a)
        ``ds`` integrates over EXTERIOR facets only (the reference's warning at :747-755), ``coef`` is evaluated in
        the facet's cell.  g is taken P1 on each facet (exact for Constants).
        """
        mesh = self._functionspace._mesh
        vs = self._functionspace.value_sizes[subspace_id]
        n = mesh.num_vertices()
        out = np.zeros(n if vs == 1 else n * vs)
        f = mesh.facets()
        d = mesh.dim
        for spec in getattr(self, 'von_neumann_bcs', {}).values():
            if spec['subspace_id'] != subspace_id:
                continue
            meas = spec['measure']
            ids = meas.subdomain_data().array()
            mask = (ids == meas.subdomain_id_) & f['exterior']
            if not mask.any():
                continue
            verts = f['vertices'][mask]
            area = mesh.facet_measures(verts)
            w = area if coefficient_per_cell is None else area * np.asarray(coefficient_per_cell)[f['cell0'][mask]]
            k = verts.shape[1]
            # facet mass matrix of P1: |F|/(k(k+1)) (1 + delta_ab)
            g = interpolate_nodal(spec['bc_value'], _SubMesh(mesh, verts.reshape(-1)), vs)
            g = g.reshape(len(verts), k) if vs == 1 else g.reshape(len(verts), k, vs)
            gsum = g.sum(axis=1, keepdims=True)
            loc = (g + gsum) / (k * (k + 1))                           # [F, k(, vs)]
            if vs == 1:
                np.add.at(out, verts.reshape(-1), (w[:, None] * loc).reshape(-1))
            else:
                dofs = verts[:, :, None] * vs + np.arange(vs)[None, None, :]
                np.add.at(out, dofs.reshape(-1), (w[:, None, None] * loc).reshape(-1))
        return out


class _SubMesh:
    """Just enough of a mesh (points + count) to interpolate a value object at selected nodes."""

    def __init__(self, mesh, nodes):
        self.points = mesh.points[np.asarray(nodes, dtype=np.int64)]

    def num_vertices(self):
        return len(self.points)


# ---------------------------------------------------------------------------------------------------
class Parameters:
    """helper_classes.py:910-1077"""

    def __init__(self, functionspace, subdomains, time_dependent=False):
        self.logger = logging.getLogger(__name__)
        self.time_dependent = time_dependent
        self._functionspace = functionspace
        self._subdomains = subdomains
        self._iv_base_name = 'iv'
        self.params_required, self.params_optional = [], []
        if self.time_dependent:
            self.sim_time = 1
            self.sim_time_step = 1

    def get_iv_map(self, return_name=True):
        if self._functionspace.has_subspaces:
            return {sid: (self._get_iv_name(sid) if return_name else self.get_iv(sid))
                    for sid in self._functionspace.subspaces.get_subspace_ids()}
        return self._get_iv_name() if return_name else self.get_iv(None)

    def _get_iv_name(self, subspace_id=None):
        if subspace_id is not None:
            return self._iv_base_name + '_' + self._functionspace.subspaces.names.get(subspace_id)
        return self._iv_base_name

    def get_iv(self, subspace_id):
        name = self._get_iv_name(subspace_id)
        if hasattr(self, name):
            return getattr(self, name)
        self.logger.warning("Initial value expression '%s' for subspace %s undefined" % (name, subspace_id))

    def _set_iv(self, iv, subspace_id=None, replace=False):
        name = self._get_iv_name(subspace_id=subspace_id)
        if not hasattr(self, name) or replace:
            setattr(self, name, iv)
        else:
            self.logger.warning("Initial value expression '%s' already exists ... do nothing" % name)

    def set_initial_value_expressions(self, ivs=None, replace=False):
        ivs = ivs or {}
        if isinstance(ivs, dict):
            for subspace_id, iv in ivs.items():
                self._set_iv(iv, subspace_id, replace=replace)
        else:
            self._set_iv(ivs, None, replace=replace)

    def create_initial_value_function(self):
        """helper_classes.py:983-986"""
        iv_map = self.get_iv_map(return_name=False)
        if isinstance(iv_map, dict):
            iv_map = {k: (v if v is not None else 0.0) for k, v in iv_map.items()}
        return self._functionspace.project_over_space(iv_map)

    def define_required_params(self, params_name_list=None):
        param_list = copy.deepcopy(list(params_name_list or []))
        if self.time_dependent:
            param_list.extend(['sim_time', 'sim_time_step'])
        self.params_required = sorted(set(param_list))

    def define_optional_params(self, params_name_list=None):
        self.params_optional = sorted(set(params_name_list or []))

    def _check_param_arguments(self, kw_args):
        required = set(self.params_required)
        available = set(kw_args.keys())
        if required <= available:
            for p in available - required:
                self.logger.info("    - parameter '%s' not needed" % p)
            return True
        for p in required - available:
            self.logger.warning("    - parameter '%s' required but not available" % p)
        return False

    def set_parameter(self, param_name, param):
        """helper_classes.py:1028-1035: a dict value becomes a DiscontinuousScalar (and is kept as <name>_dict)."""
        if isinstance(param, dict):
            value = self._subdomains.create_discontinuous_scalar_from_parameter_map(param, param_name, replace=True)
            setattr(self, param_name + '_dict', param)
            setattr(self, param_name, value)
        else:
            setattr(self, param_name, param)

    def get_parameter(self, param_name):
        if hasattr(self, param_name):
            return getattr(self, param_name)
        self.logger.warning("Parameter '%s' has not been set." % param_name)
        return None

    def init_parameters(self, parameter_dict):
        """helper_classes.py:1045-1053 (an incomplete set is only a warning in the reference, too)."""
        if self._check_param_arguments(parameter_dict):
            for name, value in parameter_dict.items():
                if name in self.params_required or name in self.params_optional:
                    self.set_parameter(name, value)
                else:
                    self.logger.info("Parameter '%s' will be ignored." % name)
            return True
        self.logger.warning("Parameterset incomplete cannot initialize.")
        return False

    def time_update_parameters(self, time):
        """helper_classes.py:1055-1077"""
        iv_map = self.get_iv_map()
        iv_names = iv_map.values() if isinstance(iv_map, dict) else [iv_map]
        for plist in (self.params_required, self.params_optional, iv_names):
            for p in plist:
                obj = getattr(self, p, None)
                if obj is None or isinstance(obj, (int, float)):
                    continue
                try:
                    obj.t = time
                except Exception:
                    pass


# ---------------------------------------------------------------------------------------------------
class TimeSeriesDataTimePoint:
    """helper_classes.py:1083-1107"""

    def __init__(self, time, time_step, recording_step):
        self.time = time
        self.recording_step = recording_step
        self.time_step = time_step

    def set_field(self, field):
        self.field = field

    def get_field(self):
        return getattr(self, 'field', None)

    def get_time(self):
        return self.time

    def get_time_step(self):
        return self.time_step

    def get_recording_step(self):
        return self.recording_step


class TimeSeriesData:
    """helper_classes.py:1110-1181"""

    def __init__(self, name, functionspace):
        self.logger = logging.getLogger(__name__)
        self._functionspace = functionspace
        self.name = name
        self.data = {}

    def exists_recording_step(self, recording_step):
        return recording_step in self.data

    def add_observation(self, field, time, time_step, recording_step, replace=False):
        obs = TimeSeriesDataTimePoint(time=time, time_step=time_step, recording_step=recording_step)
        obs.set_field(field.copy(deepcopy=True))
        if self.exists_recording_step(recording_step) and not replace:
            self.logger.warning("Recording step %i already exists" % recording_step)
            return
        self.data[recording_step] = obs

    def get_observation(self, recording_step):
        if self.exists_recording_step(recording_step):
            return self.data[recording_step]
        self.logger.warning("No solution available for recording step '%d'" % recording_step)

    def get_all_recording_steps(self):
        return sorted(self.data.keys())

    def get_most_recent_observation(self):
        return self.get_observation(max(self.data.keys()))

    def get_solution_function(self, subspace_name=None, subspace_id=None, recording_step=None):
        obs = self.get_most_recent_observation() if recording_step is None else self.get_observation(recording_step)
        if obs is not None:
            return self._functionspace.split_function(obs.get_field(), subspace_id=subspace_id,
                                                      subspace_name=subspace_name)


class TimeSeriesMultiData:
    """helper_classes.py:1184-1308; the HDF5 file of the reference becomes a .npz (no h5py / DOLFIN HDF5File here)."""

    def __init__(self):
        self.logger = logging.getLogger(__name__)
        self.time_series_prefix = 'tds_'

    def exists_time_series(self, name):
        return hasattr(self, self.time_series_prefix + name)

    def exists_recording_step(self, name, recording_step):
        return self.get_time_series(name).exists_recording_step(recording_step)

    def get_all_time_series(self):
        return {a.replace(self.time_series_prefix, ''): getattr(self, a) for a in dir(self)
                if a.startswith(self.time_series_prefix)}

    def register_time_series(self, name, functionspace, replace=False):
        if self.exists_time_series(name) and not replace:
            self.logger.warning("TimeSeries '%s' already exists" % name)
            return
        setattr(self, self.time_series_prefix + name, TimeSeriesData(name=name, functionspace=functionspace))

    def get_time_series(self, name):
        if self.exists_time_series(name):
            return getattr(self, self.time_series_prefix + name)
        self.logger.warning("TimeSeries '%s' does not exist." % name)

    def get_observation(self, name, recording_step):
        tsd = self.get_time_series(name)
        return tsd.get_observation(recording_step) if tsd is not None else None

    def add_observation(self, name, field, time, time_step, recording_step, replace=False):
        tsd = self.get_time_series(name)
        if tsd is not None:
            tsd.add_observation(field, time, time_step, recording_step, replace=replace)

    def get_solution_function(self, name, subspace_name=None, subspace_id=None, recording_step=None):
        tsd = self.get_time_series(name)
        return tsd.get_solution_function(subspace_name, subspace_id, recording_step) if tsd is not None else None

    def get_all_recording_steps(self, name):
        tsd = self.get_time_series(name)
        return tsd.get_all_recording_steps() if tsd is not None else None

    @staticmethod
    def _npz_path(path):
        return path if path.endswith('.npz') else os.path.splitext(path)[0] + '.npz'

    def save_to_hdf5(self, path_to_file, replace=False):
        path = self._npz_path(path_to_file)
        if os.path.exists(path) and not replace:
            self.logger.warning("File '%s' exists already; not overwriting." % path)
            return None
        out = {}
        for name, ts in self.get_all_time_series().items():
            steps = ts.get_all_recording_steps()
            out['%s/steps' % name] = np.asarray(steps, dtype=np.int64)
            out['%s/time' % name] = np.asarray([ts.data[s].time for s in steps], dtype=np.float64)
            out['%s/time_step' % name] = np.asarray([ts.data[s].time_step for s in steps], dtype=np.int64)
            for s in steps:
                for k, v in ts.data[s].field.components.items():
                    out['%s/vector_%d/%s' % (name, s, 'main' if k is None else k)] = v
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        np.savez(path, **out)
        return path

    def load_from_hdf5(self, path_to_file):
        path = self._npz_path(path_to_file)
        if not os.path.exists(path):
            self.logger.warning("File '%s' does not exist" % path)
            return
        z = np.load(path)
        for name, ts in self.get_all_time_series().items():
            if '%s/steps' % name not in z:
                continue
            for s, t, tstep in zip(z['%s/steps' % name], z['%s/time' % name], z['%s/time_step' % name]):
                f = ts._functionspace.new_function()
                for k in list(f.components):
                    f.components[k] = z['%s/vector_%d/%s' % (name, s, 'main' if k is None else k)]
                self.add_observation(name, f, time=float(t), time_step=int(tstep), recording_step=int(s), replace=True)


class Results:
    """helper_classes.py:1312-1453"""

    def __init__(self, functionspace, subdomains=None, output_dir=None):
        self.logger = logging.getLogger(__name__)
        self._functionspace = functionspace
        self.current_time_step = 0
        self.output_dir = output_dir
        self.ts_name = 'solution'
        self.data = TimeSeriesMultiData()
        self.data.register_time_series(self.ts_name, functionspace=functionspace)
        if subdomains is not None:
            self._subdomains = subdomains

    def set_save_output_dir(self, output_dir):
        self.output_dir = output_dir

    def add_to_results(self, current_sim_time, current_time_step, recording_step, field, replace=False):
        self.data.add_observation(name=self.ts_name, time=current_sim_time, time_step=current_time_step,
                                  recording_step=recording_step, field=field, replace=replace)

    def exists_recording_step(self, recording_step):
        return self.data.exists_recording_step(name=self.ts_name, recording_step=recording_step)

    def get_result(self, recording_step):
        return self.data.get_observation(name=self.ts_name, recording_step=recording_step)

    def get_solution_function(self, subspace_name=None, subspace_id=None, recording_step=None):
        return self.data.get_solution_function(name=self.ts_name, subspace_name=subspace_name,
                                               subspace_id=subspace_id, recording_step=recording_step)

    def get_recording_steps(self):
        return self.data.get_all_recording_steps(self.ts_name)

    # -- file output: the reference's layout (helper_classes.py:1350-1452) -------------------------------------
    #   'vtk'  : <output_dir>/<field>/<field>_<step:05d>.pvd  (+ the .vtu piece DOLFIN's File writes next to it)
    #   'xdmf' : <output_dir>/solution.xdmf, every subspace an Attribute of a temporal grid collection; the heavy
    #            data goes to solution.bin instead of solution.h5 (no HDF5 library here, see utils/xdmf_io.py)
    def get_function_save_name(self, function_name, recording_step, method='vtk'):
        if method == 'xdmf':
            return "solution_xdmf"
        return "%s_%05d" % (function_name, recording_step)

    def save_function(self, function, function_name, function_save_name, time, subspace_id=None, method='xdmf'):
        from ..utils.vtu_io import write_vtu
        from ..utils.xdmf_io import XDMFFile
        mesh = self._functionspace._mesh
        is_cell_data = isinstance(function, np.ndarray) or hasattr(function, 'array')
        if not is_cell_data and not isinstance(function, Function):
            function = self._functionspace.project_over_space(function, subspace_id=subspace_id)
        if method == 'xdmf':
            if not hasattr(self, 'output_xdmf_file'):
                self.output_xdmf_file = XDMFFile(os.path.join(self.output_dir, function_save_name + '.xdmf'))
                self.output_xdmf_file.write(mesh)
            if is_cell_data:
                self.logger.warning("cell functions are not written to XDMF")
                return
            self.output_xdmf_file.write_checkpoint(function, function_name, time)
        elif method == 'vtk':
            folder = os.path.join(self.output_dir, function_name)
            os.makedirs(folder, exist_ok=True)
            piece = function_save_name + '000000.vtu'
            if is_cell_data:
                vals = np.asarray(function.array() if hasattr(function, 'array') else function)
                write_vtu(os.path.join(folder, piece), mesh.points, mesh.cells, {}, {function_name: vals})
            else:
                write_vtu(os.path.join(folder, piece), mesh.points, mesh.cells, {function_name: function.values()}, {})
            with open(os.path.join(folder, function_save_name + '.pvd'), 'w') as f:
                f.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n<Collection>\n'
                        '<DataSet timestep="%.17g" part="0" file="%s" />\n</Collection>\n</VTKFile>\n' %
                        (float(time), piece))
        else:
            self.logger.warning("Save method '%s' is not defined" % method)

    def save_solution(self, recording_step, time, function=None, method='xdmf'):
        if method is None or self.output_dir is None:
            return
        if function is None:
            function = self.get_solution_function(recording_step=recording_step)
        fs = self._functionspace
        if fs.has_subspaces:
            for sid in fs.subspaces.get_subspace_ids():
                name = fs.subspaces.get_subspace_name(sid)
                self.save_function(fs.split_function(function, subspace_id=sid), name,
                                   self.get_function_save_name(name, recording_step, method=method), time, sid,
                                   method=method)
        else:
            self.save_function(function, fs.name, self.get_function_save_name(fs.name, recording_step, method=method),
                               time, method=method)

    def save_label_function(self, recording_step, time, method='xdmf'):
        name = 'label_map'
        self.save_function(self._subdomains.subdomains, name, self.get_function_save_name(name, recording_step, method),
                           time, method=method)

    def save_solution_start(self, method='xdmf', clear_all=False):
        """Careful with clear_all: it removes the whole output directory, as in the reference."""
        if method is None or self.output_dir is None:
            return
        if os.path.exists(self.output_dir) and clear_all:
            shutil.rmtree(self.output_dir, ignore_errors=True)
        os.makedirs(self.output_dir, exist_ok=True)
        if method == 'xdmf':
            from ..utils.xdmf_io import XDMFFile
            for ext in ('.xdmf', '.bin'):
                try:
                    os.remove(os.path.join(self.output_dir, 'solution' + ext))
                except OSError:
                    pass
            self.output_xdmf_file = XDMFFile(os.path.join(self.output_dir, 'solution.xdmf'))
            self.output_xdmf_file.write(self._functionspace._mesh)
        if hasattr(self, '_subdomains') and hasattr(self._subdomains, 'subdomains') and method == 'vtk':
            self.save_label_function(0, 0, method=method)

    def save_solution_end(self, method='xdmf'):
        if method == 'xdmf' and hasattr(self, 'output_xdmf_file'):
            self.output_xdmf_file.close()

    def save_solution_hdf5(self, save_path=None):
        if save_path is None:
            if self.output_dir is None:
                return None
            save_path = os.path.join(self.output_dir, 'solution_timeseries.h5')
        return self.data.save_to_hdf5(save_path, replace=True)


class Plotting:
    """Plotting is out of scope; kept as a no-op so run(plot=True) does not fail."""

    def __init__(self, results, output_dir=None):
        self.results = results
        self.output_dir = output_dir

    def plot_all(self, recording_step):
        return None
