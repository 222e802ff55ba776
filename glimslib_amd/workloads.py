"""
The BASELINE.json configurations as deterministic synthetic inputs (closed-form fields, no RNG).

The reference's bundled atlas meshes are git-LFS stubs (test_cases/data/brain_atlas_mesh_3d.vtu:1-3), so the
3-D configs use DOLFIN-style box meshes scaled to the brain's extent with two synthetic tissues; parameters
and initial values are the ones of the reference scripts cited per config.
"""
from __future__ import annotations

import numpy as np

from .mesh import BoxMesh, RectangleMesh

# tissue ids of the reference's atlas label maps (test_case_comparison_3D_atlas.py:46-49)
CSF, GM, WM, VENT = 1, 2, 3, 4


class Workload:
    def __init__(self, name, mesh, cell_label, tables, c0, dt, n_steps, mechanics, dirichlet_nodes=None):
        self.name = name
        self.mesh = mesh
        self.cell_label = cell_label          # int32 [M]
        self.tables = tables                  # dict of per-label lists: D, rho, gamma, E, nu
        self.c0 = c0                          # nodal initial concentration
        self.dt = dt
        self.n_steps = n_steps
        self.mechanics = mechanics
        self.dirichlet_nodes = dirichlet_nodes  # nodes with u = 0 (all components) or None

    def per_cell(self, key):
        return np.asarray(self.tables[key], dtype=np.float64)[self.cell_label]


def _exterior_nodes(mesh):
    f = mesh.facets()
    return np.unique(f['vertices'][f['exterior']])


def config_c1():
    """
    test_case_simulation_tumor_growth_2D_subdomains.py:35-107 -- 50x50 square, tissues A (x > -0.2) / B,
    spot initial condition, u = 0 on the whole boundary, dt = 1, 10 steps.
    """
    mesh = RectangleMesh((-5.0, -5.0), (5.0, 5.0), 50, 50)
    # label rule int(label(midpoint)) applied to the DG1 image of (x >= 0 ? 1 : 2): cells whose three vertices all
    # have x < 0 stay 2, any cell touching x >= 0 truncates to 1
    lv = np.where(mesh.points[:, 0] >= 0.0, 1.0, 2.0)[mesh.cells]
    label = lv.mean(axis=1).astype(np.int64).astype(np.int32)
    tables = dict(D=[0.0, 0.1, 0.0], rho=[0.0, 0.1, 0.0], gamma=[0.0, 0.2, 0.0],
                  E=[10e6, 0.001, 0.001], nu=[0.49, 0.40, 0.10])
    r = np.sqrt((mesh.points[:, 0] - 2.5) ** 2 + (mesh.points[:, 1] - 2.5) ** 2)
    c0 = np.where(r < 0.4, 1.0, 0.0)
    return Workload("C1 2D 50x50 two-subdomain square", mesh, label, tables, c0, 1.0, 10, True,
                    _exterior_nodes(mesh))


def config_c2(n=46):
    """Unit cube, homogeneous D = rho = 0.1, centred Gaussian (pattern of ..._2D_uniform.py:56), RD only."""
    mesh = BoxMesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n)
    label = np.ones(mesh.num_cells(), dtype=np.int32)
    tables = dict(D=[0.0, 0.1], rho=[0.0, 0.1], gamma=[0.0, 0.1], E=[1.0, 3e-3], nu=[0.3, 0.45])
    c0 = np.exp(-1.0 * ((mesh.points - 0.5) ** 2).sum(axis=1))
    return Workload("C2 unit cube n=%d" % n, mesh, label, tables, c0, 1.0, 20, False)


def _brain_box(n, mechanics, n_steps, name):
    """
    Box scaled to the atlas extent, white-matter ellipsoid inside a grey-matter shell, parameters of
    test_case_comparison_3D_atlas.py:87-121 and the Gaussian seed of :71-72.
    """
    mesh = BoxMesh((0.0, -240.0, 0.0), (240.0, 0.0, 155.0), n, n, n)
    mid = mesh.cell_midpoints()
    q = ((mid[:, 0] - 120.0) / 80.0) ** 2 + ((mid[:, 1] + 120.0) / 80.0) ** 2 + ((mid[:, 2] - 77.5) / 50.0) ** 2
    label = np.where(q < 1.0, WM, GM).astype(np.int32)
    del mid, q
    tables = dict(D=[0.0, 0.0, 0.01, 0.05, 0.0], rho=[0.0, 0.0, 0.05, 0.05, 0.0],
                  gamma=[0.0, 0.1, 0.1, 0.1, 0.1], E=[1.0, 1000e-6, 3000e-6, 3000e-6, 1000e-6],
                  nu=[0.3, 0.45, 0.45, 0.45, 0.3])
    d2 = ((mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1)
    c0 = np.exp(-0.5 * d2)
    return Workload(name, mesh, label, tables, c0, 1.0, n_steps, mechanics,
                    _exterior_nodes(mesh) if mechanics else None)


class LocalWorkload:
    """One rank's share of a box workload, built without the whole mesh (partition.partition_box_mesh): the fields bench.py
    needs of a Workload, restricted to the rank's local (owned | ghost) nodes and cells."""

    def __init__(self, name, part, cell_label, tables, c0, dt, n_steps, mechanics, dirichlet_local, frame, n_nodes, n_cells):
        self.name = name
        self.part = part                      # partition.LocalPart
        self.cell_label = cell_label          # int32 [local cells]
        self.tables = tables
        self.c0 = c0                          # [local nodes]
        self.dt = dt
        self.n_steps = n_steps
        self.mechanics = mechanics
        self.dirichlet_local = dirichlet_local  # OWNED local nodes with u = 0, or None
        self.frame = frame                    # (lo, hi) of the whole mesh
        self.n_nodes = n_nodes                # whole mesh
        self.n_cells = n_cells


def brain_box_local(n, mechanics, n_steps, name, n_parts, rank):
    """_brain_box's workload as rank ``rank`` of ``n_parts`` sees it: same owner map, same local numbering, same labels and
    seed as partitioning the whole Workload (tests/test_partition_box.py), from ~1 / n_parts of the host work."""
    from .partition import partition_box_mesh
    p0, p1 = (0.0, -240.0, 0.0), (240.0, 0.0, 155.0)
    part = partition_box_mesh(p0, p1, n, n, n, n_parts, rank)
    mid = part.points[part.cells].mean(axis=1)
    q = ((mid[:, 0] - 120.0) / 80.0) ** 2 + ((mid[:, 1] + 120.0) / 80.0) ** 2 + ((mid[:, 2] - 77.5) / 50.0) ** 2
    label = np.where(q < 1.0, WM, GM).astype(np.int32)
    tables = dict(D=[0.0, 0.0, 0.01, 0.05, 0.0], rho=[0.0, 0.0, 0.05, 0.05, 0.0],
                  gamma=[0.0, 0.1, 0.1, 0.1, 0.1], E=[1.0, 1000e-6, 3000e-6, 3000e-6, 1000e-6],
                  nu=[0.3, 0.45, 0.45, 0.45, 0.3])
    d2 = ((part.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1)
    c0 = np.exp(-0.5 * d2)
    dirichlet = None
    if mechanics:   # the exterior nodes of a box: an index 0 or n on some axis
        g = part.global_ids[:part.n_own]
        i, j, k = g % (n + 1), (g // (n + 1)) % (n + 1), g // ((n + 1) * (n + 1))
        dirichlet = np.flatnonzero((i == 0) | (i == n) | (j == 0) | (j == n) | (k == 0) | (k == n))
    return LocalWorkload(name, part, label, tables, c0, 1.0, n_steps, mechanics, dirichlet,
                         (np.array(p0), np.array(p1)), (n + 1) ** 3, 6 * n ** 3)


def local_by_name(name, n, n_parts, rank):
    """LocalWorkload of the box configs (c3, c4, c5), None for workloads that have no generator-side partition."""
    name = name.lower()
    if name == 'c3':
        n = n or 99
        return brain_box_local(n, False, 50, "C3 brain-extent box n=%d, 2 tissues" % n, n_parts, rank)
    if name == 'c4':
        n = n or 215
        return brain_box_local(n, False, 500, "C4 brain-extent box n=%d, 2 tissues" % n, n_parts, rank)
    if name == 'c5':
        n = n or 99
        return brain_box_local(n, True, 50, "C5 coupled (c + u) brain-extent box n=%d" % n, n_parts, rank)
    return None


def config_c3(n=99, mechanics=False):
    return _brain_box(n, mechanics, 50, "C3 brain-extent box n=%d, 2 tissues" % n)


def config_c4(n=215, mechanics=False):
    return _brain_box(n, mechanics, 500, "C4 brain-extent box n=%d, 2 tissues" % n)


def config_c5(n=99):
    return _brain_box(n, True, 50, "C5 coupled (c + u) brain-extent box n=%d" % n)


def config_unstructured(n_points=200000, mechanics=False, seed=0, jitter=None):
    """
    Truly unstructured stand-in for the CGAL atlas meshes: Delaunay tetrahedralisation (scipy / Qhull) of uniformly
    random points (np.random.default_rng(seed)) in the brain-extent box, WM ellipsoid in a GM shell, same parameters
    as C3/C4.  Row lengths range from ~6 to ~45 (mean ~16), cell volumes over three orders of magnitude.
    ``jitter`` (e.g. 0.3): instead, the nodes of a lattice of about n_points nodes, the interior ones moved by up to
    jitter x spacing per axis -- an unstructured mesh of bounded quality, nearer to what a quality-controlled mesher emits.
    """
    from scipy.spatial import Delaunay
    from .mesh import Mesh
    rng = np.random.default_rng(seed)
    ext, org = np.array([240.0, 240.0, 155.0]), np.array([0.0, -240.0, 0.0])
    if jitter is None:
        pts = rng.random((int(n_points), 3)) * ext + org
    else:
        hsp = (ext.prod() / float(n_points)) ** (1.0 / 3.0)
        m = np.maximum(2, np.round(ext / hsp).astype(int))
        ax = [np.linspace(0.0, ext[a], m[a] + 1) for a in range(3)]
        X, Y, Z = np.meshgrid(*ax, indexing='ij')
        pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        inner = np.ones(len(pts), dtype=bool)
        for a in range(3):
            inner &= (pts[:, a] > 0.0) & (pts[:, a] < ext[a])
        pts[inner] += (rng.random((int(inner.sum()), 3)) - 0.5) * 2.0 * float(jitter) * (ext / m)
        pts += org
    cells = Delaunay(pts).simplices.astype(np.int32)
    X = pts[cells]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6.0
    cells = cells[vol > 1e-9 * vol.mean()]                       # drop numerically flat slivers Qhull may emit
    mesh = Mesh(pts, cells)
    mid = mesh.cell_midpoints()
    q = ((mid[:, 0] - 120.0) / 80.0) ** 2 + ((mid[:, 1] + 120.0) / 80.0) ** 2 + ((mid[:, 2] - 77.5) / 50.0) ** 2
    label = np.where(q < 1.0, WM, GM).astype(np.int32)
    tables = dict(D=[0.0, 0.0, 0.01, 0.05, 0.0], rho=[0.0, 0.0, 0.05, 0.05, 0.0],
                  gamma=[0.0, 0.1, 0.1, 0.1, 0.1], E=[1.0, 1000e-6, 3000e-6, 3000e-6, 1000e-6],
                  nu=[0.3, 0.45, 0.45, 0.45, 0.3])
    d2 = ((pts - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1)
    c0 = np.exp(-0.005 * d2)
    return Workload("unstructured Delaunay mesh, %d points%s" % (len(pts), "" if jitter is None else " (lattice jittered by %.2f h)" % jitter),
                    mesh, label, tables, c0, 1.0, 50, mechanics,
                    _exterior_nodes(mesh) if mechanics else None)


def by_name(name, n=None):
    name = name.lower()
    if name == 'c1':
        return config_c1()
    if name == 'c2':
        return config_c2(*( [n] if n else []))
    if name == 'c3':
        return config_c3(*([n] if n else []))
    if name == 'c4':
        return config_c4(*([n] if n else []))
    if name == 'c5':
        return config_c5(*([n] if n else []))
    if name in ('u', 'unstructured'):
        return config_unstructured(*([n] if n else []))
    raise KeyError(name)


def b_spmv_bytes(nnz, n_rows):
    """Algorithmic bytes of one CSR SpMV, fp64 values + int32 columns."""
    return 12 * int(nnz) + 20 * int(n_rows)
