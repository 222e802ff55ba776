"""
N > 1 path on CPU: two processes (gloo, world_size 2) each build their LocalPart with glimslib_amd.partition and
run the *same* algorithm the device executes per time step -- halo exchange of the iterate, owned-row operator
application, one all-reduce per single-reduction PCG iteration -- with the CPU oracle's operators as the local
kernels.  The distributed result must equal the serial oracle step.  This pins the halo plan (ownership, ghost
ordering, send lists) and the collective sequence that libglimship issues through RCCL on the GPUs.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from glimslib_amd.mesh import BoxMesh
from glimslib_amd.partition import partition_mesh
from oracle.glims_oracle import OracleTumorGrowth, rel_l2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _halo_exchange(part, vec):
    """vec: local vector [n_local]; fills ghost slots from the owners (same plan as glims_set_halo)."""
    reqs, off = [], part.n_own
    for j, q in enumerate(part.peer_rank):
        sb = torch.from_numpy(np.ascontiguousarray(vec[part.send_idx[part.send_ptr[j]:part.send_ptr[j + 1]]]))
        reqs.append(dist.isend(sb, int(q)))
    bufs = []
    for q, cnt in zip(part.peer_rank, part.recv_count):
        rb = torch.empty(int(cnt), dtype=torch.float64)
        reqs.append(dist.irecv(rb, int(q)))
        bufs.append((off, rb))
        off += int(cnt)
    for r in reqs:
        r.wait()
    for o, rb in bufs:
        vec[o:o + len(rb)] = rb.numpy()


def _allsum(vals):
    t = torch.tensor(vals, dtype=torch.float64)
    dist.all_reduce(t)
    return t.numpy()


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mesh = BoxMesh((0, 0, 0), (2.0, 1.0, 1.0), 8, 5, 4)
        lab_g = (mesh.cell_midpoints()[:, 0] > 1.0).astype(int)
        D, rho = np.array([0.05, 0.01]), np.array([0.1, 0.05])
        c0_g = np.exp(-6 * ((mesh.points - np.array([1.0, 0.5, 0.5])) ** 2).sum(1))
        part = partition_mesh(mesh.points, mesh.cells, world, rank)
        lab = lab_g[part.cell_ids]
        o = OracleTumorGrowth(part.points, part.cells, D[lab], rho[lab], 0.1, 1e-3, 0.4, 1.0)
        n_own = part.n_own
        own = slice(0, n_own)
        c = c0_g[part.global_ids].copy()
        b = (o.M @ c)[own]                       # ghosts of c are valid on entry
        target = None
        for newton in range(20):
            _halo_exchange(part, c)
            A = o.rd_jacobian(c)                 # only owned rows are complete -- only those are used
            r = b - (0.5 * (A @ c + o.S @ c))[own]
            nr = np.sqrt(_allsum([r @ r])[0])
            if target is None:
                target = 1e-11 * nr
            if nr <= target:
                break
            # Chronopoulos-Gear PCG, one all-reduce per iteration (solver.hip: k_cg_scalars / k_cg_update)
            dinv = 1.0 / A.diagonal()[own]
            u = np.zeros(part.n_local)
            u[own] = dinv * r
            p = np.zeros(n_own)
            s = np.zeros(n_own)
            x = np.zeros(n_own)
            alpha = gamma_old = 1.0
            for it in range(500):
                _halo_exchange(part, u)
                w = (A @ u)[own]
                gamma, delta, rr = _allsum([r @ u[own], w @ u[own], r @ r])
                if np.sqrt(rr) <= 0.01 * target:
                    break
                beta = 0.0 if it == 0 else gamma / gamma_old
                alpha = gamma / (delta - (0.0 if it == 0 else beta * gamma / alpha))
                gamma_old = gamma
                p = u[own] + beta * p
                s = w + beta * s
                x += alpha * p
                r = r - alpha * s
                u[own] = dinv * r
            c[own] += x
        _halo_exchange(part, c)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=part.global_ids[:n_own], c=c[own],
                 ghost_gid=part.global_ids[n_own:], ghost_c=c[n_own:], newton=newton)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_step_equals_serial(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    mesh = BoxMesh((0, 0, 0), (2.0, 1.0, 1.0), 8, 5, 4)
    lab = (mesh.cell_midpoints()[:, 0] > 1.0).astype(int)
    D, rho = np.array([0.05, 0.01]), np.array([0.1, 0.05])
    o = OracleTumorGrowth(mesh.points, mesh.cells, D[lab], rho[lab], 0.1, 1e-3, 0.4, 1.0)
    c0 = np.exp(-6 * ((mesh.points - np.array([1.0, 0.5, 0.5])) ** 2).sum(1))
    c_ref, _ = o.rd_step(c0)
    c = np.full(mesh.num_vertices(), np.nan)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        c[z['gid']] = z['c']
        assert rel_l2(z['ghost_c'], c_ref[z['ghost_gid']]) < 1e-9          # ghosts current after the step
        assert 1 <= int(z['newton']) <= 6
    assert not np.isnan(c).any()
    assert rel_l2(c, c_ref) < 1e-9


def _uid_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glimslib_amd.parallel import broadcast_unique_id
        uid = broadcast_unique_id(dist, rank)
        with open(os.path.join(out_dir, "uid%d.bin" % rank), "wb") as f:
            f.write(uid)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_unique_id_reaches_every_rank(tmp_path):
    """The id bench.py / the API hand to glims_comm_init: generated on rank 0, identical bytes on all ranks."""
    world = 2
    mp.spawn(_uid_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = open(os.path.join(str(tmp_path), "uid0.bin"), "rb").read()
    b = open(os.path.join(str(tmp_path), "uid1.bin"), "rb").read()
    assert len(a) == 256 and a == b and any(a)


class _FakeHandle:
    """Stands in for _backend.Handle in the collective agreement logic of parallel.setup_node_mailbox."""

    def __init__(self, fail_map, fail_selftest):
        self.fail_map, self.fail_selftest, self.calls = fail_map, fail_selftest, []

    def comm_mailbox(self, name):
        self.calls.append(("map", name))
        if name is not None and self.fail_map:
            raise RuntimeError("hipHostRegister failed (simulated)")

    def comm_mailbox_selftest(self):
        self.calls.append(("selftest",))
        if self.fail_selftest:
            raise RuntimeError("timed out (simulated)")


def _mailbox_worker(rank, world, port, out_dir, scenario):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("GLIMS_ALLREDUCE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from glimslib_amd.parallel import setup_node_mailbox
        h = _FakeHandle(fail_map=(scenario == "map" and rank == 1), fail_selftest=(scenario == "selftest" and rank == 0))
        used = setup_node_mailbox(h, dist, rank)
        names = [c[1] for c in h.calls if c[0] == "map"]
        np.savez(os.path.join(out_dir, "mb_%s_%d.npz" % (scenario, rank)), used=used, n_map=len(names),
                 first=str(names[0]), last=str(names[-1]), selftests=sum(c[0] == "selftest" for c in h.calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["ok", "map", "selftest"])
def test_node_mailbox_is_used_by_all_ranks_or_by_none(tmp_path, scenario):
    """parallel.setup_node_mailbox: every rank maps the same shared-memory name; if mapping or the self-test fails on
    ANY rank, ALL ranks switch the mailbox off again (and fall back to RCCL / the transport's all-reduce)."""
    world = 2
    mp.spawn(_mailbox_worker, args=(world, _free_port(), str(tmp_path), scenario), nprocs=world, join=True)
    z = [np.load(os.path.join(str(tmp_path), "mb_%s_%d.npz" % (scenario, r))) for r in range(world)]
    assert str(z[0]['first']) == str(z[1]['first']) and str(z[0]['first']).startswith("/glims_")
    assert not os.path.exists("/dev/shm" + str(z[0]['first']))                 # rank 0 unlinked the object
    if scenario == "ok":
        assert all(bool(q['used']) for q in z) and all(int(q['selftests']) == 1 for q in z)
        assert all(str(q['last']) == str(q['first']) for q in z)
    else:
        assert not any(bool(q['used']) for q in z)
        assert all(str(q['last']) == "None" for q in z)                        # comm_mailbox(None) on every rank
        if scenario == "map":
            assert all(int(q['selftests']) == 0 for q in z)                    # nobody runs the collective self-test
