"""
Array-based stand-ins for the handful of DOLFIN names the forward path's *callers* use.

The reference does ``from glimslib import fenics_local as fenics`` (glimslib/fenics_local.py:1-25 = ``from dolfin
import *``) and then builds meshes, expressions, constants and boundary ``SubDomain`` classes with it
(e.g. test_cases/test_simulation_tumor_growth/test_case_simulation_tumor_growth_2D_subdomains.py:31-66).  This
module offers the same spellings on plain numpy arrays so that such scripts (and the parity tests) read like the
reference's.  It is NOT a FEniCS re-implementation: no forms, no assembly, no function spaces -- the numerics live
in libglimship.
"""
from __future__ import annotations

import logging
import re

import numpy as np

from .mesh import Mesh, RectangleMesh, BoxMesh, UnitSquareMesh, UnitCubeMesh  # noqa: F401  (re-exported)

WARNING, INFO, PROGRESS, DEBUG, ERROR = logging.WARNING, logging.INFO, 16, logging.DEBUG, logging.ERROR
DOLFIN_EPS = 3.0e-16


def is_version(spec):
    """The reference branches on ``fenics.is_version("<2018.1.x")``; this build behaves like 2017.2 (True)."""
    return spec.startswith("<")


def set_log_level(level):
    logging.getLogger("glimslib_amd").setLevel(level if isinstance(level, int) else logging.WARNING)


class Point:
    def __init__(self, *coords):
        if len(coords) == 1 and np.ndim(coords[0]) == 1:
            coords = tuple(coords[0])
        self._x = np.asarray(coords, dtype=np.float64)

    def array(self):
        return self._x

    def __getitem__(self, i):
        return self._x[i]

    def __len__(self):
        return len(self._x)


class Constant:
    """Scalar or vector constant; evaluates to itself everywhere."""

    def __init__(self, value):
        self.value = np.asarray(value, dtype=np.float64)

    def values(self):
        return self.value.reshape(-1)

    def value_size(self):
        return int(self.value.size)

    def __call__(self, X):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        if self.value.ndim == 0:
            return np.full(len(X), float(self.value))
        return np.tile(self.value.reshape(1, -1), (len(X), 1))

    def __float__(self):
        return float(self.value)

    def __repr__(self):
        return "Constant(%s)" % (self.value.tolist(),)


_TERNARY = re.compile(r"^(.*?)\?(.*):(.*)$", re.S)


def _split_top_level(s, sep):
    """Split at the first top-level occurrence of ``sep`` (outside parentheses)."""
    depth = 0
    for i, ch in enumerate(s):
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        elif ch == sep and depth == 0:
            return s[:i], s[i + 1:]
    return None


def _split_all_top_level(s, op):
    """Split at every top-level occurrence of the 2-character operator ``op``."""
    parts, depth, last, i = [], 0, 0, 0
    while i < len(s):
        ch = s[i]
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        elif depth == 0 and s.startswith(op, i):
            parts.append(s[last:i])
            last = i + len(op)
            i += len(op)
            continue
        i += 1
    parts.append(s[last:])
    return parts


def _translate_parenthesised(s):
    """Recursively translate parenthesised sub-expressions that contain logical / ternary operators."""
    out, i = [], 0
    while i < len(s):
        if s[i] == "(":
            depth, j = 1, i + 1
            while j < len(s) and depth:
                depth += s[j] == "("
                depth -= s[j] == ")"
                j += 1
            inner = s[i + 1:j - 1]
            if any(t in inner for t in ("&&", "||", "?", "!")):
                out.append("(" + _c_to_numpy(inner) + ")")
            else:
                out.append("(" + _translate_parenthesised(inner) + ")")
            i = j
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def _strip_outer_parens(s):
    s = s.strip()
    while s.startswith("(") and s.endswith(")"):
        depth = 0
        ok = True
        for i, ch in enumerate(s):
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
                if depth == 0 and i != len(s) - 1:
                    ok = False
                    break
        if not ok:
            break
        s = s[1:-1].strip()
    return s


def _c_to_numpy(src):
    """Translate the C++ snippet syntax of dolfin.Expression into a numpy expression (x[i] -> column i)."""
    s = _strip_outer_parens(src)
    q = _split_top_level(s, "?")
    if q is not None:
        cond, rest = q
        # matching ':' for this '?': first top-level ':' not belonging to a nested ternary
        depth = nest = 0
        for i, ch in enumerate(rest):
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            elif ch == "?" and depth == 0:
                nest += 1
            elif ch == ":" and depth == 0:
                if nest == 0:
                    a, b = rest[:i], rest[i + 1:]
                    return "np.where(%s, %s, %s)" % (_c_to_numpy(cond), _c_to_numpy(a), _c_to_numpy(b))
                nest -= 1
        raise ValueError("unbalanced ternary in expression %r" % src)
    for op, fn in (("||", "np.logical_or"), ("&&", "np.logical_and")):
        parts = _split_all_top_level(s, op)
        if len(parts) > 1:
            out = _c_to_numpy(parts[0])
            for part in parts[1:]:
                out = "%s(%s, %s)" % (fn, out, _c_to_numpy(part))
            return out
    if s.startswith("!") and not s.startswith("!="):
        return "np.logical_not(%s)" % _c_to_numpy(s[1:])
    s = _translate_parenthesised(s)
    s = re.sub(r"\bx\[(\d+)\]", r"_x\1", s)
    s = re.sub(r"\bpow\(", "np.power(", s)
    for f in ("sqrt", "exp", "log", "sin", "cos", "tan", "fabs", "tanh", "floor", "ceil"):
        s = re.sub(r"\b%s\(" % f, "np.%s(" % ("abs" if f == "fabs" else f), s)
    s = re.sub(r"\bDOLFIN_EPS\b", repr(DOLFIN_EPS), s)
    s = re.sub(r"\bpi\b", "np.pi", s)
    return s


class Expression:
    """
    ``Expression("C++ snippet" | (snippets...), degree=1, **params)`` evaluated with numpy at arrays of points.
    ``degree`` is accepted for signature compatibility; nodal interpolation is what the P1 path uses
    (the reference L2-projects with CG+AMG, i.e. nodal values +- its KSP tolerance).
    User parameters (and ``t``) are attributes, updatable between steps like on a dolfin Expression
    (helper_classes.py:1065-1077 sets ``expression.t``).
    """

    def __init__(self, cpp, degree=1, **params):
        self._src = cpp
        self._codes = [_c_to_numpy(s) for s in ([cpp] if isinstance(cpp, str) else list(cpp))]
        self._scalar = isinstance(cpp, str)
        self.degree = degree
        self._param_names = list(params)
        for k, v in params.items():
            setattr(self, k, v)

    def value_size(self):
        return len(self._codes)

    def __call__(self, X):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        env = {"np": np}
        for a in range(X.shape[1]):
            env["_x%d" % a] = X[:, a]
        for k in self._param_names:
            env[k] = getattr(self, k)
        if hasattr(self, "t") and "t" not in env:
            env["t"] = self.t
        cols = []
        for code in self._codes:
            v = eval(code, {"__builtins__": {}}, env)   # noqa: S307 -- expression text comes from the caller's script
            cols.append(np.broadcast_to(np.asarray(v, dtype=np.float64), (len(X),)).copy())
        return cols[0] if self._scalar else np.stack(cols, axis=1)


class SubDomain:
    """
    Boundary / region predicate.  Subclasses override ``inside(x, on_boundary)`` exactly as with DOLFIN (scalar
    point ``x``); for large meshes they may instead override ``inside_vectorized(X, on_boundary)`` with
    ``X [P, d]`` and a boolean array ``on_boundary [P]``.
    """

    def inside(self, x, on_boundary):
        return False

    def inside_vectorized(self, X, on_boundary):
        try:
            out = self.inside(X.T, on_boundary)       # x[0], x[1] become arrays; works for '&'-style predicates
            out = np.asarray(out)
            if out.shape == (len(X),) and out.dtype == bool:
                return out
        except Exception:
            pass
        return np.fromiter((bool(self.inside(x, bool(ob))) for x, ob in zip(X, on_boundary)), dtype=bool,
                           count=len(X))


class CellFunction:
    """Integer value per cell (stand-in for MeshFunction('size_t', mesh, dim))."""

    def __init__(self, mesh, values=None, dim=None):
        self.mesh = mesh
        n = mesh.num_cells() if values is None else len(values)
        self._a = np.zeros(n, dtype=np.int64) if values is None else np.asarray(values, dtype=np.int64).copy()

    def array(self):
        return self._a

    def set_all(self, v):
        self._a[:] = v

    def __getitem__(self, i):
        return self._a[i]

    def __setitem__(self, i, v):
        self._a[i] = v

    def __len__(self):
        return len(self._a)


MeshFunctionSizet = CellFunction


class FiniteElement:
    """Scalar Lagrange element descriptor (dolfin.FiniteElement as the reference's setup code uses it,
    simulation_tumor_growth.py:67-69); only P1 is implemented by the device path."""
    value_size_hint = 1

    def __init__(self, family="Lagrange", cell=None, degree=1):
        if degree != 1 or str(family) not in ("Lagrange", "CG", "P"):
            raise NotImplementedError("only P1 Lagrange elements are supported by the HIP path")
        self.family, self.cell, self.degree = "Lagrange", cell, 1

    def value_size(self, dim):
        return 1

    def __eq__(self, other):
        return type(other) is type(self) and other.cell == self.cell

    def __hash__(self):
        return hash((type(self).__name__, self.cell))


class VectorElement(FiniteElement):
    """dolfin.VectorElement: one P1 component per space dimension."""

    def value_size(self, dim):
        return dim


class MixedElement:
    """dolfin.MixedElement([e0, e1, ...]); sub-element i belongs to subspace id i."""

    def __init__(self, elements):
        self.elements = list(elements)

    def sub_elements(self):
        return list(self.elements)

    def __eq__(self, other):
        return isinstance(other, MixedElement) and other.elements == self.elements

    def __hash__(self):
        return hash(tuple(self.elements))


class _Vector:
    def __init__(self, owner):
        self._o = owner

    def get_local(self):
        return self._o._flat()

    array = get_local

    def __getitem__(self, idx):
        return self._o._flat()[idx]

    def __setitem__(self, idx, val):
        flat = self._o._flat()
        flat[idx] = val.get_local() if isinstance(val, _Vector) else val
        self._o._from_flat(flat)

    def norm(self, kind="l2"):
        return float(np.linalg.norm(self._o._flat()))


class Function:
    """
    Nodal P1 field(s) on a mesh.  ``components`` maps subspace id -> array ([N] scalar or [N, d] vector); a
    single-space function has one entry under key None.  The mixed solution of the tumour-growth models is
    ``{0: displacement [N, d], 1: concentration [N]}`` (simulation_tumor_growth.py:67-72).
    """

    def __init__(self, mesh, components, names=None, name="f", space=None):
        self.mesh = mesh
        self.components = {k: np.array(v, dtype=np.float64) for k, v in components.items()}
        self.names = names or {}
        self._name = name
        self.label = name
        self._space = space

    def function_space(self):
        """The space object this function was created in (identity is what callers compare)."""
        return self._space

    # -- dolfin-flavoured accessors --------------------------------------------------------------------
    def copy(self, deepcopy=True):
        f = Function(self.mesh, self.components, dict(self.names), self._name, space=self._space)
        f.label = self.label
        return f

    def assign(self, other):
        for k, v in other.components.items():
            self.components[k] = np.array(v, dtype=np.float64)

    def rename(self, name, label):
        self._name = name
        self.label = label

    def name(self):
        return self._name

    def sub(self, i):
        return Function(self.mesh, {None: self.components[i]}, name=self.names.get(i, "sub%s" % i))

    def split(self, deepcopy=False):
        return tuple(self.sub(k) for k in sorted(k for k in self.components if k is not None))

    def vector(self):
        return _Vector(self)

    def _flat(self):
        keys = sorted(self.components, key=lambda k: (-1 if k is None else k))
        return np.concatenate([self.components[k].reshape(-1) for k in keys])

    def _from_flat(self, flat):
        keys = sorted(self.components, key=lambda k: (-1 if k is None else k))
        off = 0
        for k in keys:
            n = self.components[k].size
            self.components[k] = np.asarray(flat[off:off + n], dtype=np.float64).reshape(self.components[k].shape)
            off += n

    def values(self, subspace_id=None):
        if subspace_id is None and None in self.components:
            return self.components[None]
        return self.components[subspace_id]

    def compute_vertex_values(self, mesh=None):
        v = self.values()
        return v.T.reshape(-1) if v.ndim == 2 else v

    def geometric_dimension(self):
        return self.mesh.dim

    def __call__(self, X):
        """P1 evaluation at points (brute-force cell search; meant for small meshes / a few points)."""
        X = np.atleast_2d(np.asarray(X.array() if isinstance(X, Point) else X, dtype=np.float64))
        pts, cells = self.mesh.points, self.mesh.cells
        P = pts[cells]
        T = np.transpose(P[:, 1:, :] - P[:, :1, :], (0, 2, 1))
        Tinv = np.linalg.inv(T)
        out = []
        vals = self.values()
        for x in X:
            lam = np.einsum('mij,mj->mi', Tinv, x[None, :] - P[:, 0, :])
            l0 = 1.0 - lam.sum(axis=1)
            allm = np.concatenate([l0[:, None], lam], axis=1)
            e = int(np.argmax(allm.min(axis=1)))
            out.append(np.tensordot(allm[e], vals[cells[e]], axes=(0, 0)))
        out = np.asarray(out)
        return out[0] if len(out) == 1 else out


class DG1Function:
    """Piecewise-linear discontinuous scalar: one value per (cell, vertex).  What ``project(expr, FunctionSpace(mesh,
    "DG", 1))`` returns; the reference builds its label functions that way (test_baseImplementation.py:18-20,
    test_unit_subDomains.py:14-16).  For data that is P1 on every cell the L2 projection onto DG1 is the cell-wise
    interpolant, which is what is stored."""

    def __init__(self, mesh, cell_vertex_values):
        self.mesh = mesh
        self.cell_vertex_values = np.asarray(cell_vertex_values, dtype=np.float64)

    def __array__(self, dtype=None, copy=None):
        return self.cell_vertex_values if dtype is None else self.cell_vertex_values.astype(dtype)

    def at_midpoints(self):
        return self.cell_vertex_values.mean(axis=1)


class FunctionSpace:
    """``fenics.FunctionSpace(mesh, family, degree)`` descriptor: Lagrange ("CG" / "Lagrange" / "P") or "DG", degree 1
    (scalar).  The mixed solution space of the models is simulation_helpers.helper_classes.FunctionSpace."""

    def __init__(self, mesh, family, degree=1):
        fam = {"Lagrange": "CG", "P": "CG", "CG": "CG", "DG": "DG", "Discontinuous Lagrange": "DG"}.get(str(family))
        if fam is None or int(degree) != 1:
            raise NotImplementedError("only CG1 / DG1 spaces are available")
        self._mesh, self.family, self.degree = mesh, fam, 1

    def mesh(self):
        return self._mesh


def project(expr, space, **kwargs):
    """``fenics.project`` for the two spaces above: nodal interpolation on CG1 (see interpolate_nodal), cell-wise
    interpolation on DG1.  Solver keyword arguments of the reference's calls are accepted and ignored."""
    mesh = space.mesh() if callable(getattr(space, "mesh", None)) else space._mesh
    if getattr(space, "family", "CG") == "DG":
        nodal = interpolate_nodal(expr, mesh, 1)
        return DG1Function(mesh, nodal[mesh.cells])
    return Function(mesh, {None: interpolate_nodal(expr, mesh, 1)}, space=space)


def interpolate_nodal(expr, mesh, value_size=1):
    """Nodal values of a Constant / Expression / callable / array / Function on the mesh vertices."""
    n = mesh.num_vertices()
    if isinstance(expr, Function):
        v = expr.values()
    elif isinstance(expr, (Constant, Expression)) or callable(expr):
        v = np.asarray(expr(mesh.points), dtype=np.float64)
    else:
        v = np.asarray(expr, dtype=np.float64)
        if v.ndim == 0 or v.shape == (value_size,):
            v = np.tile(v.reshape(1, -1), (n, 1))
    if value_size == 1:
        v = np.asarray(v, dtype=np.float64).reshape(n)
    else:
        v = np.asarray(v, dtype=np.float64).reshape(n, value_size)
    return v.copy()


def errornorm(f, g, norm_type="l2"):
    """Discrete relative-free L2 distance of two nodal fields (stand-in for dolfin.errornorm on P1 data)."""
    a, b = np.asarray(f.values() if isinstance(f, Function) else f), np.asarray(g.values() if isinstance(g, Function) else g)
    return float(np.linalg.norm(a - b))
