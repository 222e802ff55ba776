"""
Equal-work partition ("equal-nnz ranges"): per-rank nnz and (row, cell) incidences of the
Morton-range partition on an unstructured Delaunay mesh (rows 6..46 long) at 2 / 4 / 8 parts, against the equal-node
rule of round 1.  CPU only.
"""
import numpy as np
import scipy.sparse as sp

from glimslib_amd import workloads
from glimslib_amd.partition import node_owners, partition_mesh


def _row_nnz(n, cells):
    nv = cells.shape[1]
    rows = np.repeat(cells, nv, axis=1).ravel()
    cols = np.tile(cells, (1, nv)).ravel()
    A = sp.coo_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    return np.diff(A.indptr)


def test_work_balanced_morton_ranges_on_an_unstructured_mesh():
    w = workloads.config_unstructured(60000)
    pts, cells = w.mesh.points, w.mesh.cells
    n = len(pts)
    nnz = _row_nnz(n, cells)
    corners = np.bincount(cells.ravel(), minlength=n)
    assert nnz.min() >= 4 and nnz.max() > 2.5 * nnz.mean() * 0.9 or nnz.max() >= 30      # genuinely ragged rows
    for parts in (2, 4, 8):
        res = {}
        for rule, owner in (("work", node_owners(pts, parts, cells)), ("nodes", node_owners(pts, parts))):
            per_nnz = np.array([nnz[owner == r].sum() for r in range(parts)], dtype=np.float64)
            per_cor = np.array([corners[owner == r].sum() for r in range(parts)], dtype=np.float64)
            per_cnt = np.bincount(owner, minlength=parts)
            res[rule] = (per_nnz.max() / per_nnz.mean() - 1.0, per_cor.max() / per_cor.mean() - 1.0, per_cnt)
            assert per_cnt.min() > 0 and per_cnt.sum() == n
        print("%d parts: nnz imbalance %.2f %% (work rule) vs %.2f %% (node rule); incidences %.2f %% vs %.2f %%"
              % (parts, 100 * res["work"][0], 100 * res["nodes"][0], 100 * res["work"][1], 100 * res["nodes"][1]))
        assert res["work"][0] <= 0.03 and res["work"][1] <= 0.03
        assert res["work"][0] <= res["nodes"][0] + 1e-12


def test_structured_mesh_partition_is_unchanged_in_quality_and_plans_stay_consistent():
    w = workloads.config_c2(20)
    pts, cells = w.mesh.points, w.mesh.cells
    parts = partition_mesh(pts, cells, 4)
    owned = np.concatenate([p.owned_global for p in parts])
    assert len(owned) == len(pts) and len(np.unique(owned)) == len(pts)
    sizes = np.array([p.n_own for p in parts])
    assert sizes.max() <= 1.1 * sizes.mean()
    for p in parts:                                   # ghost groups of p for q  ==  send list of q for p
        for k, q in enumerate(p.peer_rank):
            other = parts[q]
            j = list(other.peer_rank).index(p.rank)
            sent = other.global_ids[other.send_idx[other.send_ptr[j]:other.send_ptr[j + 1]]]
            lo = p.n_own + int(p.recv_count[:k].sum())
            assert np.array_equal(sent, p.global_ids[lo:lo + int(p.recv_count[k])])


def test_rcb_parts_are_boxes_of_equal_work_for_any_rank_count():
    """Recursive coordinate bisection (the default since round 5): on an unstructured mesh and for rank counts that are no powers
    of two every part is the node set inside an axis-aligned box (the parts' bounding boxes overlap in no node), work is
    balanced to a few per cent, and the halo (ghost nodes per rank) is no larger than with contiguous Morton ranges."""
    from glimslib_amd.partition import build_local_part
    w = workloads.config_unstructured(40000)
    pts, cells = w.mesh.points, w.mesh.cells
    n = len(pts)
    corners = np.bincount(cells.ravel(), minlength=n) + 1.0
    for parts in (3, 5, 8):
        own_rcb = node_owners(pts, parts, cells)                       # method='rcb' is the default
        own_mor = node_owners(pts, parts, cells, method='morton')
        assert np.bincount(own_rcb, minlength=parts).min() > 0
        lo = np.array([pts[own_rcb == r].min(axis=0) for r in range(parts)])
        hi = np.array([pts[own_rcb == r].max(axis=0) for r in range(parts)])
        for r in range(parts):                                         # no node of another part inside part r's bounding box
            inside = ((pts >= lo[r]) & (pts <= hi[r])).all(axis=1)
            assert (own_rcb[inside] == r).all(), "part %d of %d is not a box" % (r, parts)
        work = np.array([corners[own_rcb == r].sum() for r in range(parts)])
        assert work.max() / work.mean() - 1.0 < 0.03
        ghosts = {}
        for name, own in (("rcb", own_rcb), ("morton", own_mor)):
            g = [build_local_part(pts, cells, own, r, parts) for r in range(parts)]
            ghosts[name] = (max(p.n_local - p.n_own for p in g), max(len(p.peer_rank) for p in g))
        print("%d parts: largest halo %d ghosts / %d peers (rcb) against %d / %d (morton ranges)" %
              (parts, ghosts["rcb"][0], ghosts["rcb"][1], ghosts["morton"][0], ghosts["morton"][1]))
        assert ghosts["rcb"][0] <= 1.1 * ghosts["morton"][0]
