"""
The symbolic phase of glims_create (node renumbering, SELL-64 sparsity, (row, cell) incidence lists, 16-bit column codes,
mesh metrics) runs on the device (csrc/symbolic.hip).  The host implementation it replaced (csrc/setup_host.cpp) is kept
behind the test hook GLIMS_HOST_SYMBOLIC: both must build the SAME structures, array by array -- checked through
glims_pattern_checksum -- on structured 2-D / 3-D meshes, an unstructured Delaunay mesh, a randomly permuted mesh, a
partitioned sub-mesh with ghosts, and with the window limit that forces slices onto 32-bit columns.
"""
import os

import numpy as np
import pytest

from glimslib_amd import workloads
from glimslib_amd.mesh import BoxMesh, RectangleMesh

pytestmark = pytest.mark.gpu

NAMES = ["slice_ptr", "cols", "cols16", "win_base", "win_ok", "diag_k", "cslice_ptr", "cslots", "celem", "interior",
         "boundary", "numbering", "rlen"]


def _both(backend, points, cells, label, n_own=None, env=None):
    out = []
    for host in (False, True):
        old = {k: os.environ.get(k) for k in ("GLIMS_HOST_SYMBOLIC", "GLIMS_WIN_LIMIT")}
        try:
            if host:
                os.environ["GLIMS_HOST_SYMBOLIC"] = "1"
            else:
                os.environ.pop("GLIMS_HOST_SYMBOLIC", None)
            for k, v in (env or {}).items():
                os.environ[k] = v
            h = backend.Handle(points, cells, label, n_own=n_own)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        st = h.stats()
        out.append((h.pattern_checksum(), h.numbering(), {k: st[k] for k in ('nnz', 'nnz_padded', 'n_corners', 'nnz_idx16')}))
        h.close()
    return out


def _assert_same(res):
    (cd, nd, sd), (ch, nh, sh) = res
    assert sd == sh, (sd, sh)
    assert np.array_equal(nd, nh)
    for name, a, b in zip(NAMES, cd, ch):
        assert a == b, "device and host symbolic phases differ in `%s`" % name


@pytest.mark.parametrize("case", ["box", "rect", "delaunay", "permuted"])
def test_device_symbolic_phase_equals_the_host_one(backend, case):
    if case == "box":
        mesh = BoxMesh((0, 0, 0), (7.0, 5.0, 3.0), 23, 17, 11)
    elif case == "rect":
        mesh = RectangleMesh((-5, -5), (5, 5), 61, 47)
    elif case == "delaunay":
        mesh = workloads.config_unstructured(20000).mesh
    else:
        m0 = BoxMesh((0, 0, 0), (1.0, 1.0, 1.0), 14, 13, 12)
        rng = np.random.default_rng(3)
        perm = rng.permutation(m0.num_vertices())
        pts = np.empty_like(m0.points)
        pts[perm] = m0.points                         # node i of the original is node perm[i] now
        cells = perm[m0.cells][rng.permutation(m0.num_cells())].astype(np.int32)
        from glimslib_amd.mesh import Mesh
        mesh = Mesh(pts, cells)
    label = np.ones(mesh.num_cells(), dtype=np.int32)
    _assert_same(_both(backend, mesh.points, mesh.cells, label))


def test_device_symbolic_phase_with_ghosts_and_a_window_limit(backend):
    from glimslib_amd.partition import partition_mesh
    mesh = BoxMesh((0, 0, 0), (4.0, 3.0, 2.0), 20, 15, 10)
    part = partition_mesh(mesh.points, mesh.cells, 3, 1)
    label = np.ones(len(part.cells), dtype=np.int32)
    _assert_same(_both(backend, part.points, part.cells, label, n_own=part.n_own))
    # slices that need more than 1 window fall back to 32-bit columns: mixed index streams, same decision on both sides
    res = _both(backend, mesh.points, mesh.cells, np.ones(mesh.num_cells(), dtype=np.int32), env={"GLIMS_WIN_LIMIT": "1"})
    _assert_same(res)
    assert 0 < res[0][2]['nnz_idx16'] < res[0][2]['nnz_padded']


def test_device_symbolic_phase_reports_bad_meshes(backend):
    mesh = BoxMesh((0, 0, 0), (1.0, 1.0, 1.0), 4, 4, 4)
    label = np.ones(mesh.num_cells(), dtype=np.int32)
    pts = np.vstack([mesh.points, [[2.0, 2.0, 2.0]]])           # an orphaned vertex
    with pytest.raises(backend.BackendError, match="orphaned"):
        backend.Handle(pts, mesh.cells, np.ones(mesh.num_cells(), dtype=np.int32))
    bad = mesh.cells.copy()
    bad[3, 2] = mesh.num_vertices() + 5
    with pytest.raises(backend.BackendError, match="out of range"):
        backend.Handle(mesh.points, bad, label)
    nanp = mesh.points.copy()
    nanp[7, 1] = np.nan
    with pytest.raises(backend.BackendError, match="non-finite"):
        backend.Handle(nanp, mesh.cells, label)
