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
