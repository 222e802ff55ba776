"""The generator-side partition of the box workloads (partition.partition_box_mesh, workloads.local_by_name): identical,
array by array, to partitioning the whole mesh -- what bench.py's multi-GPU runs rely on when every rank builds only its
share of config 3 / 4 / 5."""
import numpy as np
import pytest

from glimslib_amd import workloads
from glimslib_amd.mesh import BoxMesh
from glimslib_amd.partition import partition_mesh, partition_box_mesh, node_weights, box_node_weights

FIELDS = ("points", "cells", "cell_ids", "global_ids", "peer_rank", "send_ptr", "send_idx", "recv_count")


@pytest.mark.parametrize("nx,ny,nz,parts", [(9, 8, 7, 3), (12, 5, 20, 4), (3, 3, 3, 2), (16, 16, 16, 8), (5, 30, 4, 5),
                                            (1, 1, 1, 2)])
def test_box_partition_equals_the_whole_mesh_partition(nx, ny, nz, parts):
    p0, p1 = (0.0, -2.0, 0.0), (3.0, 0.0, 1.5)
    m = BoxMesh(p0, p1, nx, ny, nz)
    assert np.array_equal(node_weights(m.num_vertices(), m.cells), box_node_weights(nx, ny, nz))
    for r in range(parts):
        a = partition_mesh(m.points, m.cells, parts, r)
        b = partition_box_mesh(p0, p1, nx, ny, nz, parts, r, slab=3)
        assert a.n_own == b.n_own
        for f in FIELDS:
            assert np.array_equal(getattr(a, f), getattr(b, f)), (r, f)


@pytest.mark.parametrize("name,n,parts", [("c3", 10, 3), ("c5", 8, 2)])
def test_local_workload_equals_the_partitioned_global_one(name, n, parts):
    w = workloads.by_name(name, n)
    for r in range(parts):
        lw = workloads.local_by_name(name, n, parts, r)
        part = partition_mesh(w.mesh.points, w.mesh.cells, parts, r)
        for f in FIELDS:
            assert np.array_equal(getattr(part, f), getattr(lw.part, f)), (r, f)
        assert np.array_equal(lw.cell_label, w.cell_label[part.cell_ids])
        assert np.array_equal(lw.c0, w.c0[part.global_ids])
        assert lw.n_nodes == w.mesh.num_vertices() and lw.n_cells == w.mesh.num_cells()
        assert np.array_equal(lw.frame[0], w.mesh.points.min(axis=0)) and np.array_equal(lw.frame[1], w.mesh.points.max(axis=0))
        if w.mechanics:
            g2l = np.full(w.mesh.num_vertices(), -1, dtype=np.int64)
            g2l[part.global_ids[:part.n_own]] = np.arange(part.n_own)
            nodes = g2l[np.asarray(w.dirichlet_nodes, dtype=np.int64)]
            assert np.array_equal(np.sort(nodes[nodes >= 0]), lw.dirichlet_local)
        else:
            assert lw.dirichlet_local is None
    assert workloads.local_by_name("c2", None, 2, 0) is None
