"""
Node-based mesh partition for the single-node multi-GPU path (one process per GPU, RCCL halo exchange).

The reference's only parallel mode is DOLFIN's MPI domain decomposition (README.md:142-183; glimslib contributes
nothing but ``mesh.mpi_comm()`` handles).  This module produces the same kind of decomposition as plain arrays:

  * nodes are cut into ``n_parts`` axis-aligned boxes of equal work by recursive coordinate bisection (cumulative
    cells-per-node, i.e. equal nnz / equal (row, cell) incidences; owner map) -- or, method='morton', into contiguous
    ranges of the Morton order;
  * rank r keeps every cell that touches one of its nodes; the cells' foreign vertices are its ghosts;
  * local numbering = [owned (ascending global id) | ghosts grouped by owner rank, ascending global id];
  * the send list to peer p is "my owned nodes that share a cell with a node owned by p", ascending global id --
    which is exactly p's ghost group for this rank, so both sides derive matching lists with no communication.

Everything here is host-side numpy; the exchange itself is done by libglimship (``glims_set_halo``).
"""
from __future__ import annotations

import numpy as np


def _spread_bits(x, d):
    x = x.astype(np.uint64)
    if d == 3:
        x &= np.uint64(0x1fffff)
        x = (x | (x << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        x = (x | (x << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        x = (x | (x << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        x = (x | (x << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
    else:
        x &= np.uint64(0x7fffffff)
        x = (x | (x << np.uint64(16))) & np.uint64(0x0000ffff0000ffff)
        x = (x | (x << np.uint64(8))) & np.uint64(0x00ff00ff00ff00ff)
        x = (x | (x << np.uint64(4))) & np.uint64(0x0f0f0f0f0f0f0f0f)
        x = (x | (x << np.uint64(2))) & np.uint64(0x3333333333333333)
        x = (x | (x << np.uint64(1))) & np.uint64(0x5555555555555555)
    return x


def morton_keys(points):
    points = np.asarray(points, dtype=np.float64)
    d = points.shape[1]
    lo, hi = points.min(axis=0), points.max(axis=0)
    qmax = 2097151.0 if d == 3 else 2147483647.0
    scale = np.where(hi > lo, qmax / np.where(hi > lo, hi - lo, 1.0), 0.0)
    key = np.zeros(len(points), dtype=np.uint64)
    for a in range(d):
        q = ((points[:, a] - lo[a]) * scale[a]).astype(np.uint64)
        key |= _spread_bits(q, d) << np.uint64(a)
    return key


def node_weights(n_nodes, cells):
    """
    Work per matrix row, up to a factor: the number of cells at the node (= its (row, cell) incidences, what the
    assembly sweep streams) + 1.  On simplex meshes the row length is an affine function of it (interior node of a
    tetrahedral mesh: neighbours = cells / 2 + 2), so equal sums of this weight are equal nnz AND equal incidences
    to within a few per cent (tests/test_partition_balance.py) -- without building the adjacency on the host.
    """
    w = np.bincount(np.asarray(cells).reshape(-1), minlength=n_nodes).astype(np.float64)
    return w + 1.0


def box_node_weights(nx, ny, nz):
    """node_weights of ``BoxMesh(.., nx, ny, nz)`` without its cells: a hexahedron's six tetrahedra all contain its
    corners 0 and 7 (the shared diagonal) and two of them each of the other six corners."""
    w = np.zeros((nz + 1, ny + 1, nx + 1), dtype=np.float64)
    for c in range(8):
        dx, dy, dz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        w[dz:dz + nz, dy:dy + ny, dx:dx + nx] += 6.0 if c in (0, 7) else 2.0
    return w.ravel() + 1.0


def rcb_owners(points, n_parts, weights):
    """
    Recursive coordinate bisection: the node set is cut by a plane normal to the longest axis of its
    bounding box into two sets whose WORK is in the ratio of the ranks they will hold (floor / ceil halves of the rank count, so
    any count works), recursively.  Every part is the set of nodes inside an axis-aligned box -- by construction, on any mesh
    and for any rank count; contiguous Morton ranges are boxes only on uniform lattices with power-of-two counts.  That is what
    the partitioned elasticity multigrid wants (a rank's core on the first auxiliary grid is then a compact box:
    MgHierarchy::box_fraction) and it keeps halos near the surface minimum.  Ties (lattice nodes on the cutting plane) go by
    global id, so the result is deterministic.  DOLFIN's counterpart: ParMETIS / SCOTCH under mpirun (README.md:142-158).
    """
    points = np.asarray(points, dtype=np.float64)
    n = len(points)
    owner = np.zeros(n, dtype=np.int32)
    w = np.asarray(weights, dtype=np.float64)
    stack = [(np.arange(n, dtype=np.int64), 0, n_parts)]
    while stack:
        idx, first, k = stack.pop()
        if k == 1:
            owner[idx] = first
            continue
        kl = k // 2
        p = points[idx]
        axis = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
        order = np.lexsort((idx, p[:, axis]))                 # by coordinate, ties by global id
        cum = np.cumsum(w[idx][order])
        cut = int(np.searchsorted(cum, cum[-1] * kl / k, side='left')) + 1
        cut = min(max(cut, kl), len(idx) - (k - kl))          # every rank owns at least one node
        # lattice meshes: the cut would fall INSIDE a plane of equal coordinates (a jagged interface, two parts sharing the plane);
        # take the whole plane to the nearer side if that moves less than 1 % of the set's work (small lattices keep the jagged cut: equal work matters more there)
        xs = p[order, axis]
        if 0 < cut < len(idx) and xs[cut - 1] == xs[cut]:
            lo_ = int(np.searchsorted(xs, xs[cut], side='left'))
            hi_ = int(np.searchsorted(xs, xs[cut], side='right'))
            for cand in sorted((lo_, hi_), key=lambda c_: abs(c_ - cut)):
                moved = abs(cum[cand - 1] - cum[cut - 1]) if cand > 0 else cum[cut - 1]
                if kl <= cand <= len(idx) - (k - kl) and moved <= 0.01 * cum[-1]:
                    cut = cand
                    break
        stack.append((idx[order[:cut]], first, kl))
        stack.append((idx[order[cut:]], first + kl, k - kl))
    return owner


def node_owners(points, n_parts, cells=None, weights=None, method='rcb'):
    """
    owner[node] in [0, n_parts).  method = 'rcb' (default): recursive coordinate bisection, parts are boxes (rcb_owners);
    'morton': contiguous ranges of the Morton order (rounds 1-4).  With ``cells`` / ``weights`` the parts carry equal WORK
    (cumulative node_weights, SURVEY section 8e "equal-nnz ranges"); without, equal node counts.
    DOLFIN/ParMETIS balance vertices too, under mpirun (README.md:142-158); on the structured BASELINE meshes the two
    rules coincide, on an unstructured mesh (rows 6..46 long) equal counts leave ~5-10 % more nnz on some rank.
    """
    n = len(points)
    if n_parts == 1:
        return np.zeros(n, dtype=np.int32)
    if method == 'rcb':
        if weights is None:
            weights = node_weights(n, cells) if cells is not None else np.ones(n)
        return rcb_owners(points, n_parts, weights)
    order = np.argsort(morton_keys(points), kind='stable')
    owner = np.empty(n, dtype=np.int32)
    if cells is None and weights is None:
        bounds = (np.arange(n_parts + 1, dtype=np.int64) * n) // n_parts
    else:
        cum = np.cumsum((node_weights(n, cells) if weights is None else np.asarray(weights, dtype=np.float64))[order])
        targets = cum[-1] * np.arange(1, n_parts, dtype=np.float64) / n_parts
        inner = np.searchsorted(cum, targets, side='left') + 1
        bounds = np.concatenate([[0], inner, [n]]).astype(np.int64)
        bounds = np.maximum.accumulate(bounds)
        for r in range(1, n_parts):                       # every rank owns at least one node
            bounds[r] = max(bounds[r], bounds[r - 1] + 1)
        bounds = np.minimum(bounds, n - (n_parts - np.arange(n_parts + 1)))
        bounds[0], bounds[-1] = 0, n
    for r in range(n_parts):
        owner[order[bounds[r]:bounds[r + 1]]] = r
    return owner


class LocalPart:
    """Rank-local sub-mesh + halo plan (all index arrays are int64/int32 numpy)."""

    def __init__(self, rank, n_parts, points, cells, cell_ids, global_ids, n_own, peer_rank, send_ptr, send_idx,
                 recv_count):
        self.rank = rank
        self.n_parts = n_parts
        self.points = points            # [n_local, d]
        self.cells = cells              # [m_local, d+1] local vertex ids
        self.cell_ids = cell_ids        # [m_local] global cell ids
        self.global_ids = global_ids    # [n_local] global node id of each local node (owned first)
        self.n_own = n_own
        self.peer_rank = peer_rank
        self.send_ptr = send_ptr
        self.send_idx = send_idx        # local (owned) indices
        self.recv_count = recv_count

    @property
    def n_local(self):
        return len(self.global_ids)

    @property
    def owned_global(self):
        return self.global_ids[:self.n_own]


def build_local_part(points, cells, owner, rank, n_parts, cell_global_ids=None):
    """``cells``: every cell of the mesh, or (with ``cell_global_ids``, ascending) a subset that holds at least every cell
    touching a node of ``rank``."""
    points = np.asarray(points)
    cells = np.asarray(cells)
    own_c = owner[cells]                                     # [M, nv]
    mine = own_c == rank
    cell_mask = mine.any(axis=1)
    sel = np.flatnonzero(cell_mask)
    cell_ids = sel if cell_global_ids is None else np.asarray(cell_global_ids)[sel]
    lc = cells[sel]
    lo = own_c[sel]
    owned = np.flatnonzero(owner == rank)                    # ascending global id
    verts = np.unique(lc)
    ghosts = verts[owner[verts] != rank]
    g_owner = owner[ghosts]
    gorder = np.lexsort((ghosts, g_owner))                   # by owner, then global id
    ghosts = ghosts[gorder]
    g_owner = g_owner[gorder]
    global_ids = np.concatenate([owned, ghosts]).astype(np.int64)
    n_own = len(owned)
    # global -> local (only for nodes present here)
    g2l = np.full(len(points), -1, dtype=np.int64)
    g2l[global_ids] = np.arange(len(global_ids))
    local_cells = g2l[lc].astype(np.int32)
    peers, recv_count = np.unique(g_owner, return_counts=True)
    send_idx, send_ptr = [], [0]
    for p in peers:
        touch = (lo == p).any(axis=1)                        # my local cells that contain a node owned by p
        cand = lc[touch][(lo[touch] == rank)]
        s = np.unique(cand)                                  # ascending global id == p's ghost order for me
        send_idx.append(g2l[s])
        send_ptr.append(send_ptr[-1] + len(s))
    send_idx = np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, dtype=np.int32)
    return LocalPart(rank, n_parts, np.ascontiguousarray(points[global_ids]), local_cells, cell_ids, global_ids,
                     n_own, peers.astype(np.int32), np.asarray(send_ptr, dtype=np.int64), send_idx,
                     recv_count.astype(np.int64))


def partition_mesh(points, cells, n_parts, rank=None, balance='work', method='rcb'):
    """Returns the LocalPart of ``rank`` (or the list of all parts when rank is None).  ``balance``: 'work' (equal
    cumulative row work per rank, the default) or 'nodes' (equal node counts); ``method``: 'rcb' | 'morton' (node_owners)."""
    owner = node_owners(points, n_parts, cells if balance == 'work' else None, method=method)
    if rank is not None:
        return build_local_part(points, cells, owner, rank, n_parts)
    return [build_local_part(points, cells, owner, r, n_parts) for r in range(n_parts)]


def partition_box_mesh(p0, p1, nx, ny, nz, n_parts, rank, slab=8, method='rcb'):
    """The LocalPart ``partition_mesh(BoxMesh(p0, p1, nx, ny, nz).points, .cells, n_parts, rank)`` returns, without
    building the whole mesh's cells: node weights from the box's connectivity rule, and only the hexahedra around the
    rank's own nodes are generated (slabs of ``slab`` layers, each cut to the (ix, iy) range of the own nodes it holds).
    What a rank of a partitioned run pays at set-up drops from the whole mesh (config 4: 60 M cells, ~8 GB of
    intermediates) to its share."""
    from .mesh import box_points, box_cells
    points = box_points(p0, p1, nx, ny, nz)
    owner = node_owners(points, n_parts, weights=box_node_weights(nx, ny, nz), method=method)
    owned = np.flatnonzero(owner == rank)
    oi = owned % (nx + 1)
    oj = (owned // (nx + 1)) % (ny + 1)
    ok = owned // ((nx + 1) * (ny + 1))
    cells, ids = [], []
    for z0 in range(max(0, int(ok.min()) - 1), min(nz, int(ok.max()) + 1), slab):
        z1 = min(nz, z0 + slab, int(ok.max()) + 1)
        m = (ok >= z0) & (ok <= z1)                           # own nodes on the node layers z0 .. z1 of these hexahedra
        if not m.any():
            continue
        xr = (max(0, int(oi[m].min()) - 1), min(nx, int(oi[m].max()) + 1))
        yr = (max(0, int(oj[m].min()) - 1), min(ny, int(oj[m].max()) + 1))
        c, first = box_cells(nx, ny, nz, xr, yr, (z0, z1))
        keep = (owner[c] == rank).any(axis=1)
        cells.append(c[keep])
        ids.append((first[:, None] + np.arange(6)).reshape(-1)[keep])
    cells = np.concatenate(cells) if cells else np.zeros((0, 4), dtype=np.int32)
    ids = np.concatenate(ids) if ids else np.zeros(0, dtype=np.int64)
    return build_local_part(points, cells, owner, rank, n_parts, cell_global_ids=ids)
