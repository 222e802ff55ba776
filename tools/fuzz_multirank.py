#!/usr/bin/env python3
"""
Random partitioned runs on ONE GPU (gloo-staged halos + node-mailbox reductions, like tests/test_gpu_multirank.py):
random mesh size / dimension / world size (2-5) per seed (every fifth case an unstructured Delaunay mesh; two of three
cases with the global multigrid frame of glims_set_mg_frame); the gathered result must equal the single-rank device run.
usage: tools/fuzz_multirank.py [n_cases=12]
BIG=1: larger meshes (3-D 30-44 cells per edge, 2-D 150-260, 60-80 k Delaunay points) whose first multigrid grids exceed
6 000 nodes, with GLIMS_MG_BOX_MIN_NODES lowered: the framed cases then run the box-limited first grid with the neighbour
exchange of its residual (2-4 ranks).
"""
import os, socket, sys, tempfile
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch.distributed as dist
import torch.multiprocessing as mp

TABS = dict(D=[0.0, 0.1, 0.02], rho=[0.0, 0.1, 0.05], gamma=[0.0, 0.2, 0.1], E=[1.0, 1e-3, 3e-3], nu=[0.3, 0.40, 0.45])


def problem(seed):
    from glimslib_amd.mesh import BoxMesh, RectangleMesh
    rng = np.random.default_rng(5000 + seed)
    big = bool(os.environ.get("BIG"))
    dim = 2 + seed % 2
    if seed % 5 == 4:   # unstructured: Delaunay mesh of random points (125-point coarse stencils in the multigrid)
        from scipy.spatial import Delaunay
        from glimslib_amd.mesh import Mesh
        dim = 3
        pts = rng.random((int(rng.integers(60000, 80000) if big else rng.integers(600, 2500)), 3)) * np.array([10.0, 8.0, 6.0])
        cells = Delaunay(pts).simplices.astype(np.int32)
        X = pts[cells]
        vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6.0
        mesh = Mesh(pts, cells[vol > 1e-6 * vol.mean()])
    elif dim == 3:
        n = rng.integers(30, 45, size=3) if big else rng.integers(3, 13, size=3)
        mesh = BoxMesh((0, 0, 0), tuple(float(v) for v in n * rng.uniform(0.6, 1.5, size=3)), *[int(v) for v in n])
    else:
        n = rng.integers(150, 261, size=2) if big else rng.integers(4, 40, size=2)
        mesh = RectangleMesh((0, 0), tuple(float(v) for v in n * rng.uniform(0.6, 1.5, size=2)), *[int(v) for v in n])
    label = np.where(mesh.cell_midpoints()[:, 0] > mesh.points[:, 0].mean(), 2, 1).astype(np.int32)
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    c0 = np.exp(-0.3 * ((mesh.points - mesh.points.mean(0)) ** 2).sum(axis=1))
    world = int(rng.integers(2, 5)) if big else int(rng.integers(2, 6))
    return mesh, label, bn, c0, world, dim


def worker(rank, world, port, out_dir, seed):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if os.environ.get("BIG"):
        os.environ["GLIMS_MG_BOX_MIN_NODES"] = "6001"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glimslib_amd import _backend
        from glimslib_amd.parallel import HostStagedTransport, setup_node_mailbox
        from glimslib_amd.partition import partition_mesh
        mesh, label, bn, c0, _, dim = problem(seed)
        part = partition_mesh(mesh.points, mesh.cells, world, rank)
        h = _backend.Handle(part.points, part.cells, label[part.cell_ids], n_own=part.n_own, device=0)
        tr = HostStagedTransport(dist)
        h.set_transport(rank, world, tr.halo_cb, tr.allreduce_cb)
        h.set_halo(part.peer_rank, part.send_ptr, part.send_idx, part.recv_count)
        if seed % 3 != 0:   # two of three cases: global multigrid frame (replicated coarse levels), else rank-local hierarchies
            h.set_mg_frame(mesh.points.min(axis=0), mesh.points.max(axis=0))
        assert setup_node_mailbox(h, dist, rank)
        h.set_materials(TABS['D'], TABS['rho'], TABS['gamma'], TABS['E'], TABS['nu'])
        h.set_options(dt=1.0)
        g2l = {g: l for l, g in enumerate(part.global_ids[:part.n_own])}
        own_bn = np.array([g2l[g] for g in bn if g in g2l], dtype=np.int64)
        dofs = (own_bn[:, None] * dim + np.arange(dim)).ravel() if len(own_bn) else np.zeros(0, dtype=np.int64)
        h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
        h.setup(True)
        h.set_state(c0[part.global_ids])
        st = h.step(2) | h.solve_mechanics() | h.step(1) | h.solve_mechanics()
        c, u = h.get_state()
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), gid=part.global_ids, n_own=part.n_own, c=c,
                 u=u.reshape(-1, dim), st=st)
        h.close()
    finally:
        dist.destroy_process_group()


def main():
    from glimslib_amd import _backend
    from oracle.glims_oracle import rel_l2
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    bad = 0
    for seed in range(n_cases):
        mesh, label, bn, c0, world, dim = problem(seed)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        with tempfile.TemporaryDirectory() as d:
            try:
                mp.spawn(worker, args=(world, port, d, seed), nprocs=world, join=True)
                n = mesh.num_vertices()
                c = np.full(n, np.nan); u = np.full((n, dim), np.nan)
                for r in range(world):
                    z = np.load(os.path.join(d, "r%d.npz" % r))
                    assert int(z['st']) == 0
                    own = int(z['n_own'])
                    c[z['gid'][:own]] = z['c'][:own]; u[z['gid'][:own]] = z['u'][:own]
                h = _backend.Handle(mesh.points, mesh.cells, label)
                h.set_materials(TABS['D'], TABS['rho'], TABS['gamma'], TABS['E'], TABS['nu'])
                h.set_options(dt=1.0)
                dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
                h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
                h.setup(True)
                h.set_state(c0)
                assert (h.step(2) | h.solve_mechanics() | h.step(1) | h.solve_mechanics()) == 0
                c1, u1 = h.get_state()
                h.close()
                ec, eu = rel_l2(c, c1), rel_l2(u.reshape(-1), u1)
                # (BIG: the stiff Delaunay cases need ~300 Jacobi-PCG iterations per solve before `auto` switches to the
                #  V-cycle; two runs that stop on the same residual tolerance then differ by that tolerance times the
                #  conditioning -- 5e-9 observed, north_star asks for 1e-6)
                ok = ec < (2e-8 if os.environ.get("BIG") else 1e-9) and eu < 1e-7
                print("seed %2d: %d-D, %6d nodes, %d ranks: c %.1e u %.1e %s" % (seed, dim, n, world, ec, eu, "ok" if ok else "MISMATCH"), flush=True)
                bad += not ok
            except Exception as e:   # noqa: BLE001
                bad += 1
                print("seed %2d: %d-D, %d ranks FAILED: %r" % (seed, dim, world, e), flush=True)
    print("done, failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
