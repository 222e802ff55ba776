"""
CPU tests of the synthetic workloads (host logic only).  The brain-like unstructured mesh is triangulated brick by brick in
worker processes; these tests pin that the union of the bricks IS the Delaunay triangulation of the whole point set and a
conforming mesh of the box, whatever the brick layout and the number of workers.
"""
import numpy as np

from glimslib_amd import workloads
from glimslib_amd.mesh import Mesh


def _cell_set(cells):
    return set(map(tuple, np.sort(cells, axis=1)))


def test_brick_union_is_the_one_piece_delaunay_triangulation():
    from scipy.spatial import Delaunay
    pts, cells = workloads.brain_like_mesh(20000, bricks=(2, 3, 2), workers=2)
    whole = Delaunay(pts).simplices
    X = pts[whole]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6.0
    whole = whole[vol > 1e-9 * vol.mean()]
    assert _cell_set(cells) == _cell_set(whole)


def test_brain_like_mesh_is_conforming_fills_the_box_and_does_not_depend_on_the_workers():
    pts, cells = workloads.brain_like_mesh(30000, workers=1)
    pts2, cells2 = workloads.brain_like_mesh(30000, workers=3)
    assert np.array_equal(pts, pts2) and np.array_equal(cells, cells2)
    X = pts[cells]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6.0
    assert abs(vol.sum() - 240.0 * 240.0 * 155.0) < 1e-6 * 240.0 * 240.0 * 155.0
    assert vol.min() > 1e-4 * vol.mean()                         # bounded quality: no slivers
    m = Mesh(pts, cells)
    f = m.facets()
    # every interior facet is shared by exactly two cells (facets() would report a third owner as a further facet), the
    # exterior ones lie on the box
    ext = f['vertices'][f['exterior']]
    P = pts[ext]                                                  # [F, 3, 3]
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    on_face = np.zeros(len(ext), dtype=bool)
    for a in range(3):
        on_face |= np.all(P[:, :, a] == lo[a], axis=1) | np.all(P[:, :, a] == hi[a], axis=1)
    assert on_face.all()
    assert len(np.unique(cells)) == len(pts)                      # no orphaned vertex
    rows = np.bincount(cells.ravel(), minlength=len(pts))
    assert rows.min() >= 1 and rows.max() < 64                    # cells per node: far from the 8-bit slot limit


def test_config_brain_like_has_two_tissues_with_a_curved_interface():
    w = workloads.config_brain_like(20000, mechanics=True, workers=2)
    lab = w.cell_label
    assert set(np.unique(lab)) == {workloads.GM, workloads.WM}
    assert 0.05 < (lab == workloads.WM).mean() < 0.4
    assert w.dirichlet_nodes is not None and len(w.dirichlet_nodes) > 0
    assert 0.0 < w.c0.max() <= 1.0


def test_c4_octant_has_the_mesh_width_of_c4_and_an_eighth_of_its_rows():
    """alt.rank_octant's workload: the central octant of config C4 at C4's own spacing (what a rank of the 8-GPU run holds)."""
    w = workloads.config_c4_octant(12)
    assert w.mesh.num_vertices() == 13 ** 3 and w.mesh.num_cells() == 6 * 12 ** 3
    ext = w.mesh.points.max(axis=0) - w.mesh.points.min(axis=0)
    assert np.allclose(ext, [120.0, 120.0, 77.5])
    # n = 107 on half the extent: within 0.5 % of C4's spacing (n = 215 on the whole extent), and 1/8 of C4's rows to 1.5 %
    assert abs((120.0 / 107) / (240.0 / 215) - 1.0) < 5e-3
    assert abs(8 * 108 ** 3 / 216.0 ** 3 - 1.0) < 1e-12
    assert set(np.unique(w.cell_label)) <= {workloads.GM, workloads.WM} and w.c0.max() <= 1.0
