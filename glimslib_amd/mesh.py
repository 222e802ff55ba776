"""
Array-based simplicial meshes.

The reference takes DOLFIN ``Mesh`` objects (``FenicsSimulation.__init__(mesh)``,
glimslib/simulation/simulation_base.py:91-99) and reads only ``mesh.geometry().dim()`` plus what DOLFIN needs
internally.  Here a mesh is two arrays -- ``points [N, d]`` fp64 and ``cells [M, d+1]`` int32 -- with the few
topology queries the helper classes need (facets, exterior facets, cell midpoints).  ``RectangleMesh`` /
``BoxMesh`` reproduce DOLFIN's vertex numbering and cell splitting so that nodal fields can be compared with
FEniCS output if that is ever available.
"""
from __future__ import annotations

import numpy as np


class _Geometry:
    def __init__(self, d):
        self._d = d

    def dim(self):
        return self._d


class Mesh:
    def __init__(self, points, cells):
        self.points = np.ascontiguousarray(points, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        if self.points.ndim != 2 or self.points.shape[1] not in (2, 3):
            raise ValueError("points must be [N, 2] or [N, 3]")
        if self.cells.ndim != 2 or self.cells.shape[1] != self.points.shape[1] + 1:
            raise ValueError("cells must be [M, dim+1] (P1 simplices)")
        self._facets = None

    # -- the bits of the DOLFIN Mesh interface the reference touches ------------------------------------
    def geometry(self):
        return _Geometry(self.points.shape[1])

    def geometric_dimension(self):
        return self.points.shape[1]

    def coordinates(self):
        return self.points

    def num_vertices(self):
        return self.points.shape[0]

    def num_cells(self):
        return self.cells.shape[0]

    def mpi_comm(self):
        return None

    def ufl_cell(self):
        """Cell name as DOLFIN reports it; only consumed by the element descriptors of fenics_local."""
        return "triangle" if self.points.shape[1] == 2 else "tetrahedron"

    @property
    def dim(self):
        return self.points.shape[1]

    # -- topology ---------------------------------------------------------------------------------------
    def cell_midpoints(self):
        return self.points[self.cells].mean(axis=1)

    def cell_volumes(self):
        X = self.points[self.cells]
        J = X[:, 1:, :] - X[:, :1, :]
        d = self.dim
        return np.abs(np.linalg.det(J)) / (2.0 if d == 2 else 6.0)

    def facets(self):
        """
        Unique facets with their adjacent cells.

        Returns dict(vertices [F, d], cell0 [F], cell1 [F] (-1 on the exterior), exterior [F] bool).
        """
        if self._facets is not None:
            return self._facets
        cells = self.cells.astype(np.int64)
        nv = cells.shape[1]
        m = len(cells)
        allf = np.concatenate([cells[:, [b for b in range(nv) if b != a]] for a in range(nv)])
        owner = np.tile(np.arange(m), nv)
        key = np.sort(allf, axis=1)
        n = self.points.shape[0]
        if nv == 3:
            lin = key[:, 0] * n + key[:, 1]
        else:
            # three sorted vertex ids -> one sortable key (n < 2^21 fits in 63 bits; otherwise lexsort)
            lin = None if n >= (1 << 21) else (key[:, 0] * n + key[:, 1]) * n + key[:, 2]
        if lin is not None:
            order = np.argsort(lin, kind='stable')
            ls = lin[order]
            first = np.ones(len(ls), dtype=bool)
            first[1:] = ls[1:] != ls[:-1]
        else:
            order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
            ks = key[order]
            first = np.ones(len(ks), dtype=bool)
            first[1:] = np.any(ks[1:] != ks[:-1], axis=1)
        start = np.flatnonzero(first)
        count = np.diff(np.append(start, len(order)))
        verts = allf[order[start]]
        cell0 = owner[order[start]]
        cell1 = np.full(len(start), -1, dtype=np.int64)
        two = count == 2
        cell1[two] = owner[order[start[two] + 1]]
        self._facets = dict(vertices=verts.astype(np.int64), cell0=cell0, cell1=cell1, exterior=~two)
        return self._facets

    def facet_measures(self, verts):
        X = self.points[verts]
        if self.dim == 2:
            return np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
        return 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)


def _as_xy(p):
    if hasattr(p, 'array'):
        p = p.array()
    return np.asarray(p, dtype=np.float64).ravel()


def RectangleMesh(p0, p1, nx, ny, diagonal="right"):
    """DOLFIN ``RectangleMesh``: vertex (ix, iy) -> iy*(nx+1)+ix; 'right' diagonal = (v0,v1,v3),(v0,v2,v3)."""
    if diagonal != "right":
        raise NotImplementedError("only DOLFIN's default 'right' diagonal is reproduced")
    p0, p1 = _as_xy(p0), _as_xy(p1)
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing='xy')
    pts = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing='xy')
    v0 = (iy * (nx + 1) + ix).ravel()
    v1, v2 = v0 + 1, v0 + (nx + 1)
    v3 = v2 + 1
    cells = np.empty((nx * ny, 2, 3), dtype=np.int32)
    cells[:, 0, 0], cells[:, 0, 1], cells[:, 0, 2] = v0, v1, v3
    cells[:, 1, 0], cells[:, 1, 1], cells[:, 1, 2] = v0, v2, v3
    return Mesh(pts, cells.reshape(-1, 3))


def UnitSquareMesh(nx, ny):
    return RectangleMesh((0.0, 0.0), (1.0, 1.0), nx, ny)


BOX_TETS = [(0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7)]


def box_points(p0, p1, nx, ny, nz):
    """Vertices of DOLFIN's ``BoxMesh``: vertex (ix, iy, iz) -> (iz*(ny+1) + iy)*(nx+1) + ix."""
    p0, p1 = _as_xy(p0), _as_xy(p1)
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    zs = np.linspace(p0[2], p1[2], nz + 1)
    npts = (nx + 1) * (ny + 1) * (nz + 1)
    pts = np.empty((npts, 3))
    pts[:, 0] = np.tile(xs, (ny + 1) * (nz + 1))
    pts[:, 1] = np.tile(np.repeat(ys, nx + 1), nz + 1)
    pts[:, 2] = np.repeat(zs, (nx + 1) * (ny + 1))
    return pts


def box_cells(nx, ny, nz, xr=None, yr=None, zr=None):
    """Cells of ``BoxMesh`` for the hexahedra ix in xr = (lo, hi), iy in yr, iz in zr (half-open; default: all), in the
    order of the whole mesh (hexahedron (iz*ny + iy)*nx + ix, six tetrahedra each).  Returns (cells [m, 4] of GLOBAL vertex
    ids, first_cell_id [m / 6] = 6 * hexahedron index of each hexahedron visited)."""
    xr = (0, nx) if xr is None else xr
    yr = (0, ny) if yr is None else yr
    zr = (0, nz) if zr is None else zr
    sx, sy = 1, nx + 1
    sz = (nx + 1) * (ny + 1)
    iz, iy, ix = np.meshgrid(np.arange(zr[0], zr[1], dtype=np.int64), np.arange(yr[0], yr[1], dtype=np.int64),
                             np.arange(xr[0], xr[1], dtype=np.int64), indexing='ij')
    v0 = (iz * sz + iy * sy + ix).ravel().astype(np.int32)
    hexa = ((iz * ny + iy) * nx + ix).ravel()
    del ix, iy, iz
    off = {0: 0, 1: sx, 2: sy, 3: sx + sy, 4: sz, 5: sx + sz, 6: sy + sz, 7: sx + sy + sz}
    cells = np.empty((len(v0), 6, 4), dtype=np.int32)
    for t, tet in enumerate(BOX_TETS):
        for m, corner in enumerate(tet):
            cells[:, t, m] = v0 + np.int32(off[corner])
    return cells.reshape(-1, 4), 6 * hexa


def BoxMesh(p0, p1, nx, ny, nz):
    """DOLFIN ``BoxMesh``: 6 tetrahedra per hexahedron, all sharing the v0-v7 diagonal."""
    return Mesh(box_points(p0, p1, nx, ny, nz), box_cells(nx, ny, nz)[0])


def UnitCubeMesh(nx, ny, nz):
    return BoxMesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), nx, ny, nz)
