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


def config_c4_octant(n=107):
    """The central octant of config C4 at C4's OWN mesh width (half the extent per axis, n = 107 cells: h = 1.12 mm against
    1.116): what one rank of an 8-GPU run of C4 holds -- same spacing, same tissues, same seed, hence the same spectrum and
    iteration counts as the headline run (the brain-extent box at n = 107 is a coarser PROBLEM with more Krylov passes per step)."""
    ctr = np.array([120.0, -120.0, 77.5])
    half = np.array([60.0, 60.0, 38.75])
    mesh = BoxMesh(tuple(ctr - half), tuple(ctr + half), n, n, n)
    mid = mesh.cell_midpoints()
    q = ((mid[:, 0] - 120.0) / 80.0) ** 2 + ((mid[:, 1] + 120.0) / 80.0) ** 2 + ((mid[:, 2] - 77.5) / 50.0) ** 2
    label = np.where(q < 1.0, WM, GM).astype(np.int32)
    tables = dict(D=[0.0, 0.0, 0.01, 0.05, 0.0], rho=[0.0, 0.0, 0.05, 0.05, 0.0],
                  gamma=[0.0, 0.1, 0.1, 0.1, 0.1], E=[1.0, 1000e-6, 3000e-6, 3000e-6, 1000e-6],
                  nu=[0.3, 0.45, 0.45, 0.45, 0.3])
    d2 = ((mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1)
    c0 = np.exp(-0.5 * d2)
    return Workload("C4 central octant n=%d (C4's mesh width, 1/8 of its rows), 2 tissues" % n, mesh, label, tables, c0, 1.0, 500,
                    False)


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
    if name in ('c4o', 'c4_octant'):
        return config_c4_octant(*([n] if n else []))
    if name in ('u', 'unstructured'):
        return config_unstructured(*([n] if n else []))
    if name in ('bl', 'brain_like', 'brain-like'):     # n = number of points here
        return config_brain_like(*([n] if n else []), isolate=True)
    raise KeyError(name)


def b_spmv_bytes(nnz, n_rows):
    """Algorithmic bytes of one CSR SpMV, fp64 values + int32 columns."""
    return 12 * int(nnz) + 20 * int(n_rows)


# ---------------------------------------------------------------------------------------------------------------------
# A quality-controlled unstructured tetrahedral mesh at brain extent ("brain-like"): stand-in for the CGAL / MeshTool
# atlas meshes the reference's 3-D cases load (glimslib/utils/meshing.py:7-43,
# test_cases/test_simulation_tumor_growth_brain/test_case_comparison_3D_atlas.py:84-121; the bundled files are git-LFS
# stubs).  Nodes of a lattice moved by up to `jitter` x spacing per axis (boundary nodes only inside their face / along
# their edge, so the hull stays the box), Delaunay-tetrahedralised: node spacing and cell quality are bounded as in the
# output of a Delaunay-refinement mesher, row lengths vary (about 8-30, mean ~15.5) and nothing is lattice-aligned.
# Qhull needs ~35 us per point, so the point set is cut into bricks of lattice layers that are triangulated side by side
# in worker processes: brick B keeps the tetrahedra whose circumcentre lies in B.  With a margin of two lattice layers
# around every brick that union IS the Delaunay triangulation of the whole point set (an empty sphere centred inside the
# domain has radius < 1.4 spacings here; tests/test_workloads.py compares with the one-piece triangulation).
# ---------------------------------------------------------------------------------------------------------------------
_BL_EXT = np.array([240.0, 240.0, 155.0])
_BL_ORG = np.array([0.0, -240.0, 0.0])


def _bl_points(n_points, jitter, seed):
    hsp = (_BL_EXT.prod() / float(n_points)) ** (1.0 / 3.0)
    m = np.maximum(2, np.round(_BL_EXT / hsp).astype(int))
    idx = np.stack(np.meshgrid(*[np.arange(m[a] + 1) for a in range(3)], indexing='ij'), axis=-1).reshape(-1, 3)
    h = _BL_EXT / m
    rng = np.random.default_rng(seed)
    disp = (rng.random(idx.shape) - 0.5) * (2.0 * float(jitter))
    free = (idx > 0) & (idx < m[None, :])            # a boundary node moves inside its face / along its edge only
    pts = (idx + disp * free) * h[None, :] + _BL_ORG[None, :]
    return np.ascontiguousarray(pts), idx.astype(np.int32), m, h


def _circumcentres(X):
    """Circumcentres of tetrahedra X [T, 4, 3] (vertices in a canonical order: the same bits whoever computes them)."""
    a = X[:, 1:] - X[:, :1]                                         # [T, 3, 3]
    rhs = 0.5 * (a * a).sum(axis=2)
    det = np.linalg.det(a)
    ok = np.abs(det) > 0.0
    cc = np.zeros((len(X), 3))
    cc[ok] = np.linalg.solve(a[ok], rhs[ok][..., None])[..., 0]
    return cc + X[:, 0], ok


_BL_SHARED = {}


def _bl_brick(job):
    """Delaunay triangulation of one brick (+ margin); returns the tetrahedra (global vertex ids) this brick owns."""
    from scipy.spatial import Delaunay
    lo, hi, margin, nb_lo, nb_hi = job
    pts, idx, m, h = (_BL_SHARED[k] for k in ('pts', 'idx', 'm', 'h'))
    sel = np.ones(len(pts), dtype=bool)
    for a in range(3):
        sel &= (idx[:, a] >= lo[a] - margin) & (idx[:, a] <= hi[a] + margin)
    ids = np.flatnonzero(sel)
    tets = ids[Delaunay(pts[ids]).simplices]
    tets.sort(axis=1)                                               # canonical vertex order
    cc, ok = _circumcentres(pts[tets])
    # ownership: the brick whose half-open coordinate box holds the circumcentre (clamped into the domain; the bricks
    # at the upper faces are closed there)
    rel = (cc - _BL_ORG[None, :]) / h[None, :]
    own = ok.copy()
    for a in range(3):
        r = np.clip(rel[:, a], 0.0, float(m[a]))
        own &= (r >= lo[a]) & ((r < hi[a]) | (nb_hi[a] & (r <= hi[a])))
    tets = tets[own]
    X = pts[tets]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6.0
    # flat tetrahedra among the exactly coplanar hull nodes (volume at rounding level of the cell scale h^3 / 6) are dropped
    return tets[vol > 1e-9 * float(h.prod()) / 6.0].astype(np.int32)


def brain_like_mesh(n_points=1000000, jitter=0.3, seed=0, workers=None, bricks=None):
    """(points [N, 3], cells [M, 4]) of the jittered-lattice Delaunay mesh; deterministic for given arguments (the brick
    layout depends on n_points only, the number of worker processes does not change the result)."""
    import multiprocessing as mp
    import os
    pts, idx, m, h = _bl_points(n_points, jitter, seed)
    if bricks is None:   # ~30 k points per brick
        per_axis = np.maximum(1, np.round(m / max(1.0, (30000.0) ** (1.0 / 3.0))).astype(int))
        bricks = tuple(int(b) for b in np.minimum(per_axis, np.maximum(1, m // 8)))
    edges = [np.linspace(0, m[a], bricks[a] + 1).round().astype(int) for a in range(3)]
    jobs = []
    for i in range(bricks[0]):
        for j in range(bricks[1]):
            for k in range(bricks[2]):
                lo = (edges[0][i], edges[1][j], edges[2][k])
                hi = (edges[0][i + 1], edges[1][j + 1], edges[2][k + 1])
                last = (i == bricks[0] - 1, j == bricks[1] - 1, k == bricks[2] - 1)
                jobs.append((lo, hi, 2, None, last))
    _BL_SHARED.update(pts=pts, idx=idx, m=m, h=h)
    try:
        if workers is None:
            workers = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        workers = max(1, min(int(workers), len(jobs)))
        if workers == 1:
            parts = [_bl_brick(j) for j in jobs]
        else:   # fork: the children inherit the point set; only the owned tetrahedra travel back
            with mp.get_context('fork').Pool(workers) as pool:
                parts = pool.map(_bl_brick, jobs, chunksize=1)
    finally:
        _BL_SHARED.clear()
    return pts, np.ascontiguousarray(np.concatenate(parts, axis=0))


def _brain_like_labels(mid):
    """White matter inside a gyrified (undulating) ellipsoid, grey matter around it: a CURVED interface that no lattice
    plane follows."""
    x, y, z = mid[:, 0] - 120.0, mid[:, 1] + 120.0, mid[:, 2] - 77.5
    q = (x / 80.0) ** 2 + (y / 80.0) ** 2 + (z / 50.0) ** 2
    wave = 1.0 + 0.12 * np.sin(0.21 * x + 0.5) * np.sin(0.17 * y - 0.3) * np.cos(0.19 * z + 0.2)
    return np.where(q < wave * wave, WM, GM).astype(np.int32)


def config_brain_like(n_points=1000000, mechanics=False, jitter=0.3, seed=0, workers=None, isolate=False):
    """
    The unstructured counterpart of config C3: ~n_points nodes at brain extent, quality-controlled Delaunay tetrahedra
    (brain_like_mesh), white matter inside an undulating ellipsoid, parameters and seed of
    test_case_comparison_3D_atlas.py:71-72,87-121 (as C3), Gaussian a = 0.5, dt = 1, 50 steps.
    isolate: build the mesh in a child interpreter (its worker processes are forked there) -- for callers that have
    already initialised the GPU runtime, e.g. bench.py after its timed region.
    """
    from .mesh import Mesh
    import os
    # GLIMS_MESH_CACHE=<dir>: the mesh is read from / written to <dir>/brain_like_<args>.npz.  Needed under a profiler, whose
    # preloaded library has initialised the GPU before this program starts: neither a child interpreter nor forked workers
    # may be started from such a process (tools/collect_profiles.sh fills the cache in an un-profiled step first).
    cache = os.environ.get("GLIMS_MESH_CACHE")
    cfile = os.path.join(cache, "brain_like_%d_%g_%d.npz" % (int(n_points), float(jitter), int(seed))) if cache else None
    if cfile and os.path.exists(cfile):
        z = np.load(cfile)
        pts, cells = z["points"], z["cells"]
    elif isolate:
        import subprocess
        import sys
        import tempfile
        with tempfile.TemporaryDirectory(prefix="glims_bl_") as d:
            code = ("import sys, numpy as np; sys.path.insert(0, %r); from glimslib_amd import workloads as w; "
                    "p, c = w.brain_like_mesh(%d, %r, %d, %r); np.save(%r, p); np.save(%r, c)" %
                    (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), int(n_points), float(jitter), int(seed),
                     workers, os.path.join(d, "p.npy"), os.path.join(d, "c.npy")))
            subprocess.run([sys.executable, "-c", code], check=True)
            pts, cells = np.load(os.path.join(d, "p.npy")), np.load(os.path.join(d, "c.npy"))
    else:
        pts, cells = brain_like_mesh(n_points, jitter, seed, workers)
    if cfile and not os.path.exists(cfile):
        os.makedirs(cache, exist_ok=True)
        np.savez(cfile, points=pts, cells=cells)
    mesh = Mesh(pts, cells)
    label = _brain_like_labels(mesh.cell_midpoints())
    tables = dict(D=[0.0, 0.0, 0.01, 0.05, 0.0], rho=[0.0, 0.0, 0.05, 0.05, 0.0],
                  gamma=[0.0, 0.1, 0.1, 0.1, 0.1], E=[1.0, 1000e-6, 3000e-6, 3000e-6, 1000e-6],
                  nu=[0.3, 0.45, 0.45, 0.45, 0.3])
    d2 = ((pts - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1)
    c0 = np.exp(-0.5 * d2)
    return Workload("brain-like unstructured mesh (jittered-lattice Delaunay, %.2f h), %d nodes, 2 tissues" % (jitter, len(pts)),
                    mesh, label, tables, c0, 1.0, 50, mechanics, _exterior_nodes(mesh) if mechanics else None)
