"""
SPMD support for the simulation classes: one process per GPU (``torchrun`` / ``torch.distributed``), the mesh
partitioned by ``glimslib_amd.partition``, halos and reductions inside libglimship over RCCL.

The reference's counterpart is "run the same script under ``mpirun -np N``" (README.md:142-183): DOLFIN distributes
the mesh and PETSc does the communication.  Here every rank holds the full (host) mesh, owns a Morton range of the
nodes on its GPU, and ``sync_solution`` all-gathers the owned values so that ``sim.solution`` is the global field
on every rank.

``HostStagedTransport`` is an alternative to RCCL that plugs into ``glims_set_transport``: device buffers are
staged through the host and exchanged with the process group's CPU backend (gloo).  It exists for boxes where RCCL
cannot be used -- e.g. several ranks sharing one GPU, which RCCL rejects ("Duplicate GPU detected") -- and is what
the 2-rank single-GPU tests use.  Select it with ``GLIMS_TRANSPORT=gloo``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _backend
from .partition import partition_mesh


def dist_info():
    """(dist module or None, rank, world).  torch is only imported when a launcher environment is present or the
    caller has imported it already (a single-process run should not pay ~1 s for it)."""
    import sys
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "torch" not in sys.modules:
        return None, 0, 1
    try:
        import torch.distributed as dist
    except Exception:   # pragma: no cover
        return None, 0, 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def broadcast_unique_id(dist, rank):
    """RCCL unique id of rank 0 to every rank, as a CPU uint8 tensor (gloo side of a 'cpu:gloo,cuda:nccl' group)."""
    import torch
    buf = torch.zeros(_backend.GLIMS_UNIQUE_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(_backend.Handle.comm_unique_id()), dtype=torch.uint8).clone()
    try:
        dist.broadcast(buf, src=0)
    except Exception:
        # process group without a CPU backend (plain 'nccl'): go through the GPU
        dev = torch.device("cuda", torch.cuda.current_device())
        gbuf = buf.to(dev)
        dist.broadcast(gbuf, src=0)
        buf = gbuf.cpu()
    return bytes(buf.numpy().tobytes())


def setup_node_mailbox(h, dist, rank):
    """
    Switches the scalar all-reduces of `h` (a _backend.Handle with rank / world already set) to the node-local
    shared-memory mailbox (glims_comm_mailbox), checks it collectively and falls back to RCCL / the transport callback
    on every rank if any rank's check fails.  GLIMS_ALLREDUCE=rccl skips it.  Returns True when the mailbox is in use.
    """
    import torch
    if os.environ.get("GLIMS_ALLREDUCE", "mailbox").lower() in ("rccl", "nccl", "callback"):
        return False
    name = ["/glims_%d_%s" % (os.getpid(), os.urandom(4).hex())] if rank == 0 else [None]
    dist.broadcast_object_list(name, src=0)
    ok = 1
    try:
        h.comm_mailbox(name[0])
    except Exception as e:   # noqa: BLE001
        print("glimslib_amd: node mailbox unavailable on rank %d (%s)" % (rank, e), flush=True)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    _all_reduce_cpu(dist, flag, dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        try:
            h.comm_mailbox_selftest()
        except Exception as e:   # noqa: BLE001
            print("glimslib_amd: node mailbox self-test failed on rank %d (%s)" % (rank, e), flush=True)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        _all_reduce_cpu(dist, flag, dist.ReduceOp.MIN)
    if rank == 0:
        try:
            os.unlink("/dev/shm" + name[0])   # every rank has mapped it (or given up) by now
        except OSError:
            pass
    if int(flag.item()) != 1:
        h.comm_mailbox(None)
        return False
    return True


def _all_reduce_cpu(dist, t, op):
    try:
        dist.all_reduce(t, op=op)
    except Exception:   # process group without a CPU backend
        import torch
        g = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(g, op=op)
        t.copy_(g.cpu())


class HostStagedTransport:
    """glims_halo_fn / glims_allreduce_fn implemented with hipMemcpy staging + torch.distributed CPU collectives."""

    def __init__(self, dist, group=None):
        import torch
        self.torch = torch
        self.dist = dist
        self.group = group
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        self.hip = hip
        self.halo_cb = _backend.HALO_FN(self._halo)
        self.allreduce_cb = _backend.ALLREDUCE_FN(self._allreduce)
        # GLIMS_TRANSPORT_TRACE=<prefix>: every collective this rank enters, one line each, in <prefix><rank>.txt -- ranks whose
        # call sequences differ show where a partitioned run went out of step
        pre = os.environ.get("GLIMS_TRANSPORT_TRACE")
        self._trace = open("%s%d.txt" % (pre, dist.get_rank()), "w") if pre else None
        self._n = 0

    def _note(self, what):
        if self._trace:
            self._n += 1
            self._trace.write("%d %s\n" % (self._n, what))
            self._trace.flush()

    def _d2h(self, ptr, n):
        a = np.empty(n)
        if self.hip.hipMemcpy(a.ctypes.data, ptr, n * 8, 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        return a

    def _h2d(self, ptr, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self.hip.hipMemcpy(ptr, a.ctypes.data, a.size * 8, 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    def _halo(self, user, sendbuf, send_ptr, ghosts, recv_ptr, n_peers, peers, bs, stream):
        try:
            self._note("halo bs=%d peers=%s" % (bs, [int(peers[p]) for p in range(n_peers)]))
            self.hip.hipStreamSynchronize(stream)
            reqs, rbufs = [], []
            for p in range(n_peers):
                lo, hi = send_ptr[p] * bs, send_ptr[p + 1] * bs
                if hi > lo:      # (an empty direction is skipped on both sides: the two ranks derive the same counts)
                    sb = self.torch.from_numpy(self._d2h(sendbuf + lo * 8, hi - lo))
                    reqs.append(self.dist.isend(sb, int(peers[p]), group=self.group))
                nr = int((recv_ptr[p + 1] - recv_ptr[p]) * bs)
                if nr > 0:
                    rb = self.torch.empty(nr, dtype=self.torch.float64)
                    reqs.append(self.dist.irecv(rb, int(peers[p]), group=self.group))
                    rbufs.append((recv_ptr[p] * bs, rb))
            for r in reqs:
                r.wait()
            for off, rb in rbufs:
                self._h2d(ghosts + off * 8, rb.numpy())
            return 0
        except Exception as e:   # noqa: BLE001 -- must not unwind through the C frame
            print("glimslib_amd: halo transport failed: %r" % (e,), flush=True)
            return 1

    def _allreduce(self, user, values, n, stream):
        try:
            self._note("allreduce n=%d" % n)
            self.hip.hipStreamSynchronize(stream)
            t = self.torch.from_numpy(self._d2h(values, n))
            self.dist.all_reduce(t, group=self.group)
            self._h2d(values, t.numpy())
            return 0
        except Exception as e:   # noqa: BLE001
            print("glimslib_amd: allreduce transport failed: %r" % (e,), flush=True)
            return 1


class ThreadGroup:
    """Meeting point of ThreadedTransport's ranks (one process, one thread per rank)."""

    def __init__(self, world):
        import queue
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.pipes = {(a, b): queue.Queue() for a in range(world) for b in range(world) if a != b}
        self.values = [None] * world


class ThreadedTransport:
    """glims_halo_fn / glims_allreduce_fn for ranks that are THREADS of one process sharing one GPU -- a rehearsal transport:
    the GPU boxes allow a handful of processes on a card, so rank counts like 8 cannot be rehearsed with one process per rank
    (tests/test_gpu_multirank.py, tools/rehearse_partition.py).  Everything else is the product path: each rank its own handle
    and streams, its own sub-mesh, the library's halo packing, slice split and reduction points.  Messages travel through
    FIFO queues per ordered rank pair (both sides derive the same sequence of exchanges), all-reduces through a barrier and
    a sum in rank order."""

    def __init__(self, group, rank):
        self.g, self.rank = group, rank
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        self.hip = hip
        self.halo_cb = _backend.HALO_FN(self._halo)
        self.allreduce_cb = _backend.ALLREDUCE_FN(self._allreduce)
        self.failed = None

    def _d2h(self, ptr, n):
        a = np.empty(n)
        if self.hip.hipMemcpy(a.ctypes.data, ptr, n * 8, 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        return a

    def _h2d(self, ptr, a):
        if self.hip.hipMemcpy(ptr, a.ctypes.data, a.size * 8, 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    def _halo(self, user, sendbuf, send_ptr, ghosts, recv_ptr, n_peers, peers, bs, stream):
        try:
            self.hip.hipStreamSynchronize(stream)
            for p in range(n_peers):
                lo, hi = send_ptr[p] * bs, send_ptr[p + 1] * bs
                if hi > lo:
                    self.g.pipes[(self.rank, int(peers[p]))].put(self._d2h(sendbuf + lo * 8, hi - lo))
            for p in range(n_peers):
                nr = int((recv_ptr[p + 1] - recv_ptr[p]) * bs)
                if nr > 0:
                    a = self.g.pipes[(int(peers[p]), self.rank)].get(timeout=120)
                    assert a.size == nr, (a.size, nr)
                    self._h2d(ghosts + recv_ptr[p] * bs * 8, a)
            return 0
        except Exception as e:   # noqa: BLE001 -- must not unwind through the C frame
            self.failed = e
            print("glimslib_amd: threaded halo transport failed on rank %d: %r" % (self.rank, e), flush=True)
            return 1

    def _allreduce(self, user, values, n, stream):
        try:
            self.hip.hipStreamSynchronize(stream)
            self.g.values[self.rank] = self._d2h(values, n)
            self.g.barrier.wait(timeout=120)
            t = np.zeros(n)
            for r in range(self.g.world):       # rank order: the same bits on every rank
                t += self.g.values[r]
            self.g.barrier.wait(timeout=120)    # nobody overwrites its slot before everybody has read it
            self._h2d(values, t)
            return 0
        except Exception as e:   # noqa: BLE001
            self.failed = e
            print("glimslib_amd: threaded allreduce transport failed on rank %d: %r" % (self.rank, e), flush=True)
            return 1


def run_threaded_ranks(world, fn):
    """Runs fn(rank, ThreadedTransport) on `world` threads and returns their results in rank order (exceptions re-raised)."""
    import threading
    group = ThreadGroup(world)
    res, err = [None] * world, [None] * world

    def body(r):
        try:
            res[r] = fn(r, ThreadedTransport(group, r))
        except BaseException as e:   # noqa: BLE001
            err[r] = e
            group.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return res


class DistributedHandle:
    """
    Rank-local ``_backend.Handle`` + the index maps between global (mesh) and local (owned | ghost) numbering.
    Presents the same methods the simulation classes use on a plain Handle, with *global* arrays at the interface.
    """

    def __init__(self, points, cells, cell_label, dist, rank, world, device):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        self.n_global = len(points)
        self.dim = np.asarray(points).shape[1]
        self.part = partition_mesh(points, cells, world, rank)
        p = self.part
        self.h = _backend.Handle(p.points, p.cells, np.asarray(cell_label)[p.cell_ids], n_own=p.n_own, device=device)
        self.cpu_group = None
        use_gloo = os.environ.get("GLIMS_TRANSPORT", "rccl").lower() == "gloo"
        if use_gloo:
            self._transport = HostStagedTransport(dist)
            self.h.set_transport(rank, world, self._transport.halo_cb, self._transport.allreduce_cb)
        else:
            self.h.comm_init(rank, world, broadcast_unique_id(dist, rank))
        self.h.set_halo(p.peer_rank, p.send_ptr, p.send_idx, p.recv_count)
        pts = np.asarray(points)
        self.h.set_mg_frame(pts.min(axis=0), pts.max(axis=0))   # every rank holds the whole host mesh: global box
        self.node_mailbox = setup_node_mailbox(self.h, dist, rank)
        self.g2l_owned = np.full(self.n_global, -1, dtype=np.int64)
        self.g2l_owned[p.global_ids[:p.n_own]] = np.arange(p.n_own)
        self.options = self.h.options

    # -- pass-through -----------------------------------------------------------------------------------------
    def set_materials(self, *a):
        self.h.set_materials(*a)

    def set_options(self, **kw):
        self.h.set_options(**kw)

    def setup(self, with_mechanics=True):
        self.h.setup(with_mechanics)

    def step(self, n=1):
        return self.h.step(n)

    def solve_mechanics(self):
        return self.h.solve_mechanics()

    def stats(self):
        return self.h.stats()

    def reset_stats(self):
        self.h.reset_stats()

    def close(self):
        self.h.close()

    def project(self, rhs, rtol=1e-12):
        rhs = np.asarray(rhs, dtype=np.float64)
        loc = self.h.project(rhs[self.part.global_ids], rtol)
        n_own = self.part.n_own
        parts = [None] * self.world
        self.dist.all_gather_object(parts, (self.part.global_ids[:n_own], loc[:n_own]))
        out = np.empty_like(rhs)
        for gid, v in parts:
            out[gid] = v
        return out

    # -- global <-> local -----------------------------------------------------------------------------------------
    def _local(self, v, bs=1):
        v = np.asarray(v, dtype=np.float64).reshape(self.n_global, bs) if bs > 1 else np.asarray(v, dtype=np.float64)
        return v[self.part.global_ids]

    def set_rd_load(self, f):
        self.h.set_rd_load(None if f is None else self._local(f))

    def set_mech_load(self, f):
        self.h.set_mech_load(None if f is None else self._local(f, self.dim).reshape(-1))

    def set_dirichlet_c(self, nodes, values):
        nodes = np.asarray(nodes, dtype=np.int64)
        loc = self.g2l_owned[nodes] if len(nodes) else nodes
        keep = loc >= 0
        self.h.set_dirichlet_c(loc[keep], np.asarray(values, dtype=np.float64)[keep] if len(nodes) else values)

    def set_dirichlet_u(self, dofs, values):
        dofs = np.asarray(dofs, dtype=np.int64)
        if len(dofs) == 0:
            self.h.set_dirichlet_u(dofs, values)
            return
        node, comp = dofs // self.dim, dofs % self.dim
        loc = self.g2l_owned[node]
        keep = loc >= 0
        self.h.set_dirichlet_u(loc[keep] * self.dim + comp[keep], np.asarray(values, dtype=np.float64)[keep])

    def set_state(self, c, u=None):
        self.h.set_state(self._local(c), None if u is None else self._local(u, self.dim).reshape(-1))

    def get_state(self, want_u=True):
        """All-gathers the owned values: every rank returns the global fields."""
        c, u = self.h.get_state(want_u=want_u)
        n_own = self.part.n_own
        mine = (self.part.global_ids[:n_own], c[:n_own], None if u is None else u.reshape(-1, self.dim)[:n_own])
        parts = [None] * self.world
        self.dist.all_gather_object(parts, mine)
        cg = np.empty(self.n_global)
        ug = np.empty((self.n_global, self.dim)) if want_u else None
        for gid, cc, uu in parts:
            cg[gid] = cc
            if want_u:
                ug[gid] = uu
        return cg, (ug.reshape(-1) if want_u else None)
