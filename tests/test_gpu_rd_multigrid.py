"""
GPU tests of the multigrid-preconditioned RD solves (glims_options.rd_precond): the regime the reference's sparse LU
(simulation_tumor_growth.py:126-130) does not notice and Jacobi-PCG does -- stiffness-dominated steps, dt D / h^2 >> 1,
as in BASELINE config C2 (unit cube, D = 0.1, dt = 1).  The V-cycle is the elasticity solver's auxiliary-grid hierarchy
with 1 x 1 blocks, built on the static part S of the Newton Jacobian.  All through the C-ABI, checked against the CPU
oracle's Newton + sparse LU.
"""
import os
import socket

import numpy as np
import pytest

from glimslib_amd import workloads
from glimslib_amd.mesh import BoxMesh, RectangleMesh
from oracle.glims_oracle import OracleTumorGrowth, rel_l2

pytestmark = pytest.mark.gpu


def _stiff_problem(dim, n):
    """Two tissues with a 5x jump in D and a Dirichlet face for c: dt D / h^2 between 10 and a few hundred."""
    if dim == 3:
        mesh = BoxMesh((0, 0, 0), (1.0, 0.9, 0.8), n, n - 2, n - 4)
    else:
        mesh = RectangleMesh((0, 0), (1.0, 0.9), n, n - 7)
    mid = mesh.cell_midpoints()
    lab = (1 + (np.linalg.norm(mid - mesh.points.mean(0), axis=1) < 0.3)).astype(np.int32)
    tabs = dict(D=[0.0, 0.02, 0.1], rho=[0.0, 0.05, 0.1], gamma=[0, .1, .1], E=[1.0, 1e-3, 3e-3], nu=[.3, .4, .45])
    left = np.flatnonzero(mesh.points[:, 0] < 1e-12)
    c0 = 0.8 * np.exp(-20.0 * ((mesh.points - mesh.points.mean(0)) ** 2).sum(1))
    return mesh, lab, tabs, left, c0


def _handle(backend, mesh, lab, tabs, **opts):
    h = backend.Handle(mesh.points, mesh.cells, lab)
    h.set_materials(tabs['D'], tabs['rho'], tabs['gamma'], tabs['E'], tabs['nu'])
    h.set_options(dt=1.0, **opts)
    return h


@pytest.mark.parametrize("dim,n", [(3, 22), (2, 150)])
def test_stiff_steps_match_the_oracle_lu(backend, dim, n):
    """Three stiff steps with inhomogeneous Dirichlet data on one face: the multigrid-preconditioned path and the Jacobi
    path against Newton + sparse LU; the multigrid one needs a fraction of the Krylov iterations."""
    mesh, lab, tabs, left, c0 = _stiff_problem(dim, n)
    per = {k: np.asarray(v)[lab] for k, v in tabs.items()}
    vals = 0.2 + 0.1 * np.cos(5.0 * mesh.points[left, 1])
    o = OracleTumorGrowth(mesh.points, mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], 1.0,
                          dirichlet_c=(left, vals))
    co = c0.copy()
    for _ in range(3):
        co, _ = o.rd_step(co)
    out = {}
    for pre in (backend.RD_PRECOND_MULTIGRID, backend.RD_PRECOND_JACOBI):
        h = _handle(backend, mesh, lab, tabs, rd_precond=pre)
        h.set_dirichlet_c(left, vals)
        h.setup(False)
        h.set_state(c0)
        assert h.step(3) == 0
        st = h.stats()
        out[pre] = (h.get_state(want_u=False)[0], st['cg_its'] / st['newton_its'], st)
        h.close()
    (cm, im, sm), (cj, ij, sj) = out[backend.RD_PRECOND_MULTIGRID], out[backend.RD_PRECOND_JACOBI]
    print("dim %d: q = %.1f, PCG its per Newton solve: multigrid %.1f (%d levels, complexity %.2f), Jacobi %.1f; "
          "c vs LU %.2e / %.2e" % (dim, sm['rd_stiffness_ratio'], im, sm['rd_mg_levels'], sm['rd_mg_complexity'], ij,
                                   rel_l2(cm, co), rel_l2(cj, co)))
    assert sm['rd_precond_used'] == backend.RD_PRECOND_MULTIGRID and sj['rd_precond_used'] == backend.RD_PRECOND_JACOBI
    assert sm['rd_mg_levels'] >= 3 and sm['rd_mg_cycles'] > 0 and sj['rd_mg_cycles'] == 0
    assert rel_l2(cm, co) < 1e-9 and rel_l2(cj, co) < 1e-9            # north_star: 1e-6
    assert np.array_equal(cm[left], vals)
    assert im <= 15 and im < 0.5 * ij


def test_rd_multigrid_iteration_count_does_not_grow_with_the_mesh(backend):
    """BASELINE config C2's problem (unit cube, D = rho = 0.1, dt = 1) at n = 16 / 32 / 64: Jacobi-PCG iterations per
    Newton solve grow like 1 / h, the multigrid-preconditioned count stays put; same concentration field."""
    its = {}
    for n in (16, 32, 64):
        w = workloads.config_c2(n)
        res = {}
        for pre in (backend.RD_PRECOND_MULTIGRID, backend.RD_PRECOND_JACOBI):
            h = _handle(backend, w.mesh, w.cell_label, w.tables, rd_precond=pre)
            h.setup(False)
            h.set_state(w.c0)
            assert h.step(3) == 0
            st = h.stats()
            res[pre] = (h.get_state(want_u=False)[0], st['cg_its'] / st['newton_its'])
            h.close()
        its[n] = (res[backend.RD_PRECOND_MULTIGRID][1], res[backend.RD_PRECOND_JACOBI][1])
        assert rel_l2(res[backend.RD_PRECOND_MULTIGRID][0], res[backend.RD_PRECOND_JACOBI][0]) < 1e-9
    print("PCG iterations per Newton solve (multigrid, Jacobi):", its)
    assert all(v[0] <= 15 for v in its.values())
    assert its[64][0] <= its[16][0] + 3
    assert its[64][1] > 2.5 * its[16][1]


def test_auto_picks_by_regime(backend):
    """`auto` decides from q = mean S_ii / M_ii: the mass-dominated brain-extent configs (C3 / C4: q ~ 1) keep Jacobi, a
    stiff step on a mesh of the same size gets the hierarchy."""
    w = workloads.config_c3(40)                                         # brain parameters: D <= 0.05 mm^2/d, h = 6 mm
    h = _handle(backend, w.mesh, w.cell_label, w.tables)
    h.setup(False)
    h.set_state(w.c0)
    assert h.step(1) == 0
    st = h.stats()
    assert st['rd_precond_used'] == backend.RD_PRECOND_JACOBI and st['rd_stiffness_ratio'] < 2.0 and st['rd_mg_levels'] == 0
    # the same mesh with 10^6 times the diffusivity (dt D / h^2 ~ 1400 in white matter)
    t = dict(w.tables)
    t['D'] = [1e6 * d for d in w.tables['D']]
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.setup(False)
    h.set_state(w.c0)
    assert h.step(1) == 0
    st2 = h.stats()
    print("q: %.3f -> %.1f, preconditioner %d -> %d" % (st['rd_stiffness_ratio'], st2['rd_stiffness_ratio'],
                                                         st['rd_precond_used'], st2['rd_precond_used']))
    assert st2['rd_precond_used'] == backend.RD_PRECOND_MULTIGRID and st2['rd_stiffness_ratio'] > 500.0
    assert st2['rd_mg_levels'] >= 3
    # ... and an explicit choice overrides
    h.set_options(rd_precond=backend.RD_PRECOND_JACOBI)
    assert h.step(1) == 0 and h.stats()['rd_precond_used'] == backend.RD_PRECOND_JACOBI
    h.close()


def test_a_new_set_of_dirichlet_nodes_rebuilds_the_hierarchy(backend):
    """New VALUES on the same nodes keep the hierarchy (time-dependent boundary data, helper_classes.py:839-859); a new
    SET of constrained nodes changes the operator it was built for."""
    mesh, lab, tabs, left, c0 = _stiff_problem(3, 18)
    per = {k: np.asarray(v)[lab] for k, v in tabs.items()}
    right = np.flatnonzero(mesh.points[:, 0] > 1.0 - 1e-12)
    h = _handle(backend, mesh, lab, tabs, rd_precond=backend.RD_PRECOND_MULTIGRID)
    h.set_dirichlet_c(left, 0.1)
    h.setup(False)
    h.set_state(c0)
    assert h.step(1) == 0
    t1 = h.stats()['ms_rd_mg_setup']
    h.set_dirichlet_c(left, 0.15)
    assert h.step(1) == 0
    assert h.stats()['ms_rd_mg_setup'] == t1                             # not rebuilt
    h.set_dirichlet_c(right, 0.3)
    assert h.step(1) == 0
    c = h.get_state(want_u=False)[0]
    h.close()
    co = c0.copy()
    for nodes, val in ((left, 0.1), (left, 0.15), (right, 0.3)):
        o = OracleTumorGrowth(mesh.points, mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], 1.0,
                              dirichlet_c=(nodes, np.full(len(nodes), val)))
        co, _ = o.rd_step(co)
    assert rel_l2(c, co) < 1e-9


@pytest.mark.parametrize("flags", ["fp32_smoother", "fp64_vectors"])
def test_rd_multigrid_precision_variants(backend, flags):
    mesh, lab, tabs, left, c0 = _stiff_problem(3, 20)
    fl = {"fp32_smoother": backend.FLAG_MG_FP32_SMOOTHER, "fp64_vectors": backend.FLAG_MG_FP64_VECTORS}[flags]
    res = []
    for f in (backend.FLAG_WARM_START, backend.FLAG_WARM_START | fl):
        h = _handle(backend, mesh, lab, tabs, rd_precond=backend.RD_PRECOND_MULTIGRID, flags=f)
        h.setup(False)
        h.set_state(c0)
        assert h.step(2) == 0
        res.append(h.get_state(want_u=False)[0])
        h.close()
    assert rel_l2(res[0], res[1]) < 1e-9


def test_delaunay_mesh_with_a_stiff_step(backend):
    """General (off-lattice) mesh: 125-point coarse stencils; against the Jacobi path and the oracle's LU."""
    w = workloads.config_unstructured(30000)
    t = dict(w.tables)
    t['D'] = [2000.0 * d for d in w.tables['D']]                          # dt D / h^2 of a few hundred at a mean edge of ~6 mm
    per = {k: np.asarray(v)[w.cell_label] for k, v in t.items()}
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], 1.0)
    co = w.c0.copy()
    for _ in range(2):
        co, _ = o.rd_step(co)
    res = {}
    for pre in (backend.RD_PRECOND_MULTIGRID, backend.RD_PRECOND_JACOBI):
        h = _handle(backend, w.mesh, w.cell_label, t, rd_precond=pre)
        h.setup(False)
        h.set_state(w.c0)
        assert h.step(2) == 0
        st = h.stats()
        res[pre] = (h.get_state(want_u=False)[0], st['cg_its'] / st['newton_its'])
        h.close()
    print("Delaunay 30 k points: PCG its per solve multigrid %.1f, Jacobi %.1f" %
          (res[backend.RD_PRECOND_MULTIGRID][1], res[backend.RD_PRECOND_JACOBI][1]))
    for pre in res:
        assert rel_l2(res[pre][0], co) < 1e-9
    assert res[backend.RD_PRECOND_MULTIGRID][1] < res[backend.RD_PRECOND_JACOBI][1]
    # `auto` on a general mesh: the prediction from q is poor there, the OBSERVED Jacobi count of a step decides (four
    # times the lattice break-even: 4 x 120 below 50 k rows)
    h = _handle(backend, w.mesh, w.cell_label, t)
    h.setup(False)
    h.set_state(w.c0)
    assert h.step(2) == 0
    st = h.stats()
    ca = h.get_state(want_u=False)[0]
    h.close()
    print("auto: q = %.1f, used %d" % (st['rd_stiffness_ratio'], st['rd_precond_used']))
    assert rel_l2(ca, co) < 1e-9
    if res[backend.RD_PRECOND_JACOBI][1] > 480.0:
        assert st['rd_precond_used'] == backend.RD_PRECOND_MULTIGRID and st['rd_mg_cycles'] > 0
    elif np.sqrt(2.0 * st['rd_stiffness_ratio']) <= 480.0:
        assert st['rd_precond_used'] == backend.RD_PRECOND_JACOBI


def test_brain_like_mesh_with_stiff_steps(backend):
    """The quality-controlled unstructured mesh (workloads.config_brain_like: jittered-lattice Delaunay tetrahedra, the
    stand-in for the CGAL atlas meshes): off-lattice, 125-point coarse stencils, but bounded node spacing and cell quality --
    there the auxiliary-grid hierarchy keeps the lattice's iteration count (4-5 per Newton solve; a Delaunay mesh of random
    points, slivers included, needs ~40).  Reduced size against the oracle's Newton + sparse LU, then at 200 k nodes against
    the Jacobi path with the iteration counts asserted."""
    w = workloads.config_brain_like(12000, workers=2, isolate=True)
    t = dict(w.tables)
    t['D'] = [3000.0 * d for d in w.tables['D']]                          # dt D / h^2 of 50-300 at a node spacing of ~9 mm
    c0 = np.exp(-0.002 * ((w.mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1))
    per = {k: np.asarray(v)[w.cell_label] for k, v in t.items()}
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], 1.0)
    co = c0.copy()
    for _ in range(2):
        co, _ = o.rd_step(co)
    for pre in (backend.RD_PRECOND_MULTIGRID, backend.RD_PRECOND_JACOBI):
        h = _handle(backend, w.mesh, w.cell_label, t, rd_precond=pre)
        h.setup(False)
        h.set_state(c0)
        assert h.step(2) == 0
        assert rel_l2(h.get_state(want_u=False)[0], co) < 1e-9
        h.close()
    # 200 k nodes, diffusivities x 300 (the stiff case of tools/run_rd_precond.py)
    w = workloads.config_brain_like(200000, isolate=True)
    t = dict(w.tables)
    t['D'] = [300.0 * d for d in w.tables['D']]
    res = {}
    for pre in (backend.RD_PRECOND_MULTIGRID, backend.RD_PRECOND_JACOBI):
        h = _handle(backend, w.mesh, w.cell_label, t, rd_precond=pre)
        h.setup(False)
        h.set_state(w.c0)
        assert h.step(3) == 0
        st = h.stats()
        res[pre] = (h.get_state(want_u=False)[0], st['cg_its'] / st['newton_its'], st['rd_mg_levels'])
        h.close()
    mg_its, j_its = res[backend.RD_PRECOND_MULTIGRID][1], res[backend.RD_PRECOND_JACOBI][1]
    print("brain-like mesh, 200 k nodes, D x 300: PCG its per Newton solve multigrid %.1f, Jacobi %.1f" % (mg_its, j_its))
    assert rel_l2(res[backend.RD_PRECOND_MULTIGRID][0], res[backend.RD_PRECOND_JACOBI][0]) < 1e-9
    assert mg_its <= 8.0 and mg_its < 0.4 * j_its


# ---- partitioned: the RD hierarchy on the global frame (replicated coarse levels) through the transport hook -------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partitioned_problem(n):
    """n > 0: BASELINE config C2's unit cube; n < 0: the unit square with -n cells per edge, same coefficients."""
    if n > 0:
        return workloads.config_c2(n)
    mesh = RectangleMesh((0.0, 0.0), (1.0, 1.0), -n, -n)
    label = np.ones(mesh.num_cells(), dtype=np.int32)
    tables = dict(D=[0.0, 0.1], rho=[0.0, 0.1], gamma=[0.0, 0.1], E=[1.0, 3e-3], nu=[0.3, 0.45])
    c0 = np.exp(-1.0 * ((mesh.points - 0.5) ** 2).sum(axis=1))
    return workloads.Workload("unit square n=%d" % -n, mesh, label, tables, c0, 1.0, 20, False)


def _rd_worker(rank, world, port, out_dir, n):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GLIMS_MG_BOX_MIN_NODES"] = "6001"   # first grids above 6 000 nodes (n = 48): box-limited, neighbour exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glimslib_amd import _backend
        from glimslib_amd.parallel import HostStagedTransport
        from glimslib_amd.partition import partition_mesh
        w = _partitioned_problem(n)
        part = partition_mesh(w.mesh.points, w.mesh.cells, world, rank)
        h = _backend.Handle(part.points, part.cells, w.cell_label[part.cell_ids], n_own=part.n_own, device=0)
        tr = HostStagedTransport(dist)
        h.set_transport(rank, world, tr.halo_cb, tr.allreduce_cb)
        h.set_halo(part.peer_rank, part.send_ptr, part.send_idx, part.recv_count)
        h.set_mg_frame(w.mesh.points.min(axis=0), w.mesh.points.max(axis=0))
        t = w.tables
        h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
        h.set_options(dt=w.dt, rd_precond=_backend.RD_PRECOND_MULTIGRID)
        h.setup(False)
        h.set_state(w.c0[part.global_ids])
        status = h.step(3)
        st = h.stats()
        np.savez(os.path.join(out_dir, "rd_rank%d.npz" % rank), gid=part.global_ids, n_own=part.n_own,
                 c=h.get_state(want_u=False)[0], status=status, cg=st['cg_its'], newton=st['newton_its'])
        h.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n,world", [(24, 2), (32, 3), (48, 2), (-200, 2)])
def test_partitioned_rd_multigrid_equals_the_single_rank_run(tmp_path, backend, n, world):
    """(n = 48 and the 200 x 200 square, n = -200: first grids above 6 000 nodes -- each rank smooths its work box only and
    the first grid's residual travels by neighbour exchange, in 3-D and 2-D with one unknown per node)"""
    import torch.multiprocessing as mp
    w = _partitioned_problem(n)
    # (a framed partitioned run lays a coarser first grid from three ranks on -- replicated levels do not shrink with the
    #  rank count: the single-rank reference uses the same spacing)
    h = _handle(backend, w.mesh, w.cell_label, w.tables, rd_precond=backend.RD_PRECOND_MULTIGRID,
                mg_h_factor=2.0 if world <= 2 else 3.0)
    h.setup(False)
    h.set_state(w.c0)
    assert h.step(3) == 0
    c1 = h.get_state(want_u=False)[0]
    st1 = h.stats()
    h.close()
    mp.spawn(_rd_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    c = np.full(w.mesh.num_vertices(), np.nan)
    cg = None
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rd_rank%d.npz" % r))
        assert int(z['status']) == 0
        own = int(z['n_own'])
        c[z['gid'][:own]] = z['c'][:own]
        cg = (int(z['cg']), int(z['newton']))
    print("n = %d, %d ranks: (PCG, Newton) iterations %s, single rank (%d, %d)" %
          (n, world, cg, st1['cg_its'], st1['newton_its']))
    assert not np.isnan(c).any() and rel_l2(c, c1) < 1e-9
    assert cg[1] == st1['newton_its'] and abs(cg[0] - st1['cg_its']) <= 3
