"""Multi-GPU sharding of independent node-AVIs.

One process per GPU.  Rank g owns the contiguous node range [g*B/G, (g+1)*B/G) and assembles its own
stacked blocks on its own GPU, so no M/q traffic crosses GPUs.  The only exchange per outer sweep is
the primal blocks (count x n fp64 per rank): every rank must hold the full iterate x for the next
sweep's R_i w / B_i w terms.  Two routes:

* SharedIterate (the GPU route): the iterate lives in one IPC-shared buffer per GPU and the solve
  kernel itself stores every primal block into all of them (qpn_set_primal_mirrors) -- xGMI is
  point-to-point and the blocks are 8n bytes, so the exchange rides inside the launch; the sweep ends
  with qpn_sweep_status, a 24-byte mailbox exchange that is both the stop/raise decision and the
  barrier.  No collective on the data path.
* GatheredIterate (the default of bench.py for N > 1): ONE in-place all-gather per sweep of
  [primal blocks | sweep status] -- the route north_star names (RCCL over xGMI; gloo in the CPU tests).
* all_gather_primal + all_reduce_status: the same exchange as two collectives; ragged shards.

What does NOT shard: a single Nash pool is ONE AVI (src/avi.jl:399-400) -- shard over instances
instead.
"""
from __future__ import annotations


def node_range(total: int, world: int, rank: int):
    """Contiguous, balanced node range of `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_gather_primal(x_all, x_local, ranges, dist):
    """Reassemble the primal iterate: x_all[lo_r:hi_r] <- rank r's x_local, on every rank.

    `ranges` = [node_range(total, world, r) for r in range(world)].  Equal shard sizes use one
    all_gather_into_tensor straight into x_all (no staging); ragged shards gather padded blocks."""
    sizes = [hi - lo for lo, hi in ranges]
    if x_local.is_cuda and dist.get_backend() == "gloo":
        # rehearsal only (several ranks on one GPU, host channel gloo): stage through the host
        import torch
        got = [None] * len(sizes)
        dist.all_gather_object(got, x_local.cpu())
        for (lo, hi), b in zip(ranges, got):
            x_all[lo:hi].copy_(b.to(x_all.device))
        return x_all
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(x_all, x_local.contiguous())
        return x_all
    import torch
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]].copy_(x_local)
    bufs = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(bufs, pad)
    for (lo, hi), b in zip(ranges, bufs):
        x_all[lo:hi].copy_(b[: hi - lo])
    return x_all


def all_reduce_status(n_failed_local: int, max_resid_local: float, device, dist):
    """Tiny all-reduce of (any failure, max residual) -- the outer loop's stop/raise decision."""
    import torch
    t = torch.tensor([float(n_failed_local), float(max_resid_local)], dtype=torch.float64, device=device)
    s = t.clone()
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s[0].item()), float(t[1].item())


class GatheredIterate:
    """The iterate x [total, n] on every rank, reassembled by ONE collective per sweep (the route BASELINE.json's north_star
    names: an RCCL all-gather over xGMI; gloo in the CPU tests).

    Each rank's contribution is its primal blocks followed by an 8-double tail that carries the sweep's stop/raise pair
    (items not solved, max residual -- src/algorithm.jl:95-109 ends a sweep with solved = false when any solve of the level
    failed), so the status needs no collective of its own: buffer [world, count * n + 8], rank r's row = [x of its nodes |
    tail].  The all-gather is in place (every rank's send buffer is its own row of the receive buffer).
    Equal shard sizes only (all_gather_into_tensor); ragged shards use all_gather_primal + all_reduce_status.

    x_local          [count, n] view the solve writes (solve_nodes(x_out=...))
    finish_sweep(status, resid)   fills the tail (qpn_sweep_status on the GPU, no host sync) and issues the all-gather
    x_of(r)          rank r's [count, n] block after the sweep;  x_all() -> [total, n] copy
    sweep_result()   (items not solved over all ranks, max residual) -- host sync
    """
    TAIL = 8

    def __init__(self, eng, dist, total, n, device):
        import torch
        self.eng, self.dist = eng, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        ranges = [node_range(total, self.world, r) for r in range(self.world)]
        sizes = {hi - lo for lo, hi in ranges}
        if len(sizes) != 1:
            raise ValueError("GatheredIterate needs equal shard sizes")
        self.count, self.n, self.total = sizes.pop(), int(n), int(total)
        self.row = self.count * self.n + self.TAIL
        self.buf = torch.zeros((self.world, self.row), dtype=torch.float64, device=device)
        self.x_local = self.buf[self.rank, : self.count * self.n].view(self.count, self.n)
        self._tail = self.buf[self.rank, self.count * self.n: self.count * self.n + 4]

    def finish_sweep(self, status, resid):
        if self.buf.is_cuda:
            self.eng.sweep_status(status, resid, self._tail)       # [not solved, max resid, 1, 0], one small launch
        else:                                                      # CPU rehearsal (gloo tests): the same pair, torch ops
            self._tail[0] = float((status != 1).sum())
            self._tail[1] = float(resid.max()) if resid.numel() else 0.0
            self._tail[2] = 1.0
        if self.buf.is_cuda and self.dist.get_backend() == "gloo":
            # rehearsal only (several ranks on one GPU, host channel gloo): stage through the host
            got = [None] * self.world
            self.dist.all_gather_object(got, self.buf[self.rank].cpu())
            for r, b in enumerate(got):
                if r != self.rank:
                    self.buf[r].copy_(b.to(self.buf.device))
        else:
            self.dist.all_gather_into_tensor(self.buf.view(-1), self.buf[self.rank])
        return self.buf

    def x_of(self, r):
        return self.buf[r, : self.count * self.n].view(self.count, self.n)

    def x_all(self):
        return self.buf[:, : self.count * self.n].reshape(self.total, self.n)

    def sweep_result(self):
        t = self.buf[:, self.count * self.n: self.count * self.n + 2]
        return int(t[:, 0].sum().item()), float(t[:, 1].max().item())


def _device_tensor(addr: int, shape, device):
    """fp64 torch view of raw device memory owned by the library (qpn_shared_alloc)."""
    import torch

    class _Mem:
        pass
    m = _Mem()
    m.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f8", "data": (int(addr), False),
                                  "version": 3, "strides": None}
    return torch.as_tensor(m, device=device)


class SharedIterate:
    """The iterate x [total, n] (fp64), one replica per rank, kept identical by the solve kernels.

    x            torch view of the replica half the CURRENT sweep writes; pass x[lo:hi] as solve_nodes(x_out=...)
    finish_sweep(status, resid) -> device tensor [not solved, max resid, all ranks arrived, missed barriers so far], asynchronous;
                 after it (in stream order) x_done -- the half just written -- holds all ranks' blocks of this
                 sweep on every rank, and x flips to the other half.
    Two halves because a rank may run one sweep ahead of its slowest peer: its next sweep's stores must not land
    in the buffer that peer still reads (it cannot get two ahead -- finish_sweep waits for everyone).  A sweep
    that solves only some nodes carries the other rows over itself (x[rows] = x_done[rows]).
    Handles travel through `dist` (any backend: they are 64-byte host objects)."""

    def __init__(self, eng, dist, total, n, device, timeout_ms=1000):
        import torch
        from ._lib import MAX_MIRRORS, SWEEP_BOX_BYTES
        self.eng, self.dist = eng, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if self.world - 1 > MAX_MIRRORS:
            raise ValueError(f"SharedIterate: world {self.world} > {MAX_MIRRORS + 1}")
        self.nbytes = 2 * int(total) * int(n) * 8
        self.timeout_ms = int(timeout_ms)
        self.epoch = 0
        self._opened = []
        self.addr = self.box = 0
        # every rank takes part in both handshakes whatever happens locally, and all ranks raise together
        mine, err = None, None
        try:
            # fine-grained: peers store into this buffer while kernels of this GPU read it in later sweeps -- no stale
            # L2 lines to rely on a kernel-boundary invalidate for
            self.addr, hx = eng.shared_alloc(self.nbytes, fine_grained=True)
            self.box, hb = eng.shared_alloc(SWEEP_BOX_BYTES, fine_grained=True)
            mine = (hx, hb)
        except Exception as e:                      # noqa: BLE001 -- reported to all ranks below
            err = repr(e)
        handles = [None] * self.world
        dist.all_gather_object(handles, mine)
        self.boxes, peers = [], []
        if all(h is not None for h in handles):
            try:
                for r in range(self.world):
                    if r == self.rank:
                        self.boxes.append(self.box)
                        continue
                    px = eng.shared_open(handles[r][0]); self._opened.append(px)
                    pb = eng.shared_open(handles[r][1]); self._opened.append(pb)
                    peers.append(px); self.boxes.append(pb)
                eng.set_primal_mirrors(self.addr, self.nbytes, peers)
            except Exception as e:                  # noqa: BLE001
                err = repr(e)
        elif err is None:
            err = "a peer could not allocate its shared buffers"
        errs = [None] * self.world
        dist.all_gather_object(errs, err)
        if any(e is not None for e in errs):
            self.close()
            raise RuntimeError("SharedIterate: " + "; ".join(f"rank {r}: {e}" for r, e in enumerate(errs) if e))
        both = _device_tensor(self.addr, (2, total, n), device)
        self._halves = [both[0], both[1]]
        self.out = torch.zeros(4, dtype=torch.float64, device=device)     # [not solved, max resid, arrived, missed barriers]

    @property
    def x(self):
        return self._halves[self.epoch & 1]

    @property
    def x_done(self):
        return self._halves[(self.epoch & 1) ^ 1]

    def finish_sweep(self, status, resid):
        self.epoch += 1
        return self.eng.sweep_status(status, resid, self.out, self.rank, self.world, self.boxes, self.epoch,
                                     self.timeout_ms)

    def close(self):
        """Collective: unmap the peers' buffers, then (after a barrier) free the own ones."""
        import torch
        torch.cuda.synchronize()
        self.dist.barrier()                          # nobody unmaps while a peer's kernels may still store
        self.eng.set_primal_mirrors()
        for a in self._opened:
            self.eng.shared_close(a)
        self._opened = []
        self.dist.barrier()
        self._halves = []
        for a in (self.addr, self.box):
            if a:
                self.eng.shared_free(a)
        self.addr = self.box = 0


# ---------------------------------------------------------------------------------------------------------------------
# sharding the NET by cluster: whole independent subtrees per rank, exchange by need
# ---------------------------------------------------------------------------------------------------------------------
def net_clusters(qpn):
    """Connected components of the net's dependency graph over ALL levels: nodes joined by an edge (a parent reads its child's
    solution graph, src/programs.jl:274-285) or by one reading a variable the other owns (cost or constraint rows:
    src/avi.jl:335-340).  BASELINE config 4's 5 000 leader-follower pairs are 5 000 clusters of two nodes.  A cluster never
    reads another cluster's variables, so its outer loop (src/algorithm.jl:13-117) runs to its end without seeing the rest
    of the net.  -> list of sorted node-id lists, ordered by smallest id."""
    import numpy as np
    ids = sorted(qpn.qps)
    parent = {i: i for i in ids}

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    def join(a, b):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)

    owner = {}
    for i in ids:
        for v in qpn.qps[i].var_indices:
            if v in owner:
                join(owner[v], i)
            else:
                owner[v] = i
    for i in ids:
        for j in qpn.network_edges.get(i, ()):
            join(i, j)
        qp = qpn.qps[i]
        reads = [qp.f.local()[0]] + [qpn.constraints[c].poly.support() for c in qp.constraint_indices]
        for v in np.unique(np.concatenate(reads)).tolist():
            o = owner.get(v)
            if o is not None:
                join(o, i)
    comps = {}
    for i in ids:
        comps.setdefault(find(i), []).append(i)
    return [sorted(c) for _, c in sorted(comps.items())]


def sub_net(qpn, node_ids):
    """The net restricted to `node_ids` (whole clusters): a QPNet of its own over the variables those nodes own or read
    (variables nobody in the set owns stay what they are: parameters), nodes and constraints renumbered in order.
    -> (sub-net, variables [local -> global index], node ids [local id - 1 -> global id])."""
    import numpy as np
    from .programs import Poly, QPNet
    node_ids = sorted(node_ids)
    keep = set(node_ids)
    vs = []
    cons = []
    for i in node_ids:
        qp = qpn.qps[i]
        vs.append(np.asarray(qp.var_indices, dtype=np.int64)); vs.append(qp.f.local()[0])
        for c in qp.constraint_indices:
            if c not in cons:
                cons.append(c)
            vs.append(qpn.constraints[c].poly.support())
    var = np.unique(np.concatenate(vs))
    pos = {int(v): k for k, v in enumerate(var.tolist())}
    net = QPNet(var.size)
    cmap = {}
    for c in sorted(cons):
        P = qpn.constraints[c].poly
        cols, A = P.local()
        cid = max(net.constraints.keys(), default=0) + 1
        from .programs import Constraint
        net.constraints[cid] = Constraint(Poly.from_local(var.size, [pos[int(v)] for v in cols.tolist()], A, P.l, P.u, normalise=False,
                                                          open_lo=P.open_lo, open_hi=P.open_hi))
        cmap[c] = cid
    nmap = {}
    for i in node_ids:
        qp = qpn.qps[i]
        idx, Ql, ql = qp.f.local()
        nmap[i] = net.add_qp(Ql, ql, [cmap[c] for c in qp.constraint_indices], [pos[int(v)] for v in qp.var_indices], k=qp.f.k,
                             idx=[pos[int(v)] for v in idx.tolist()])
    net.add_edges([(nmap[i], nmap[j]) for i in node_ids for j in qpn.network_edges.get(i, ()) if j in keep])
    net.assign_constraint_groups()
    net.options = qpn.options
    net.default_initialization = np.asarray(qpn.default_initialization, dtype=np.float64)[var].copy()
    return net, var, node_ids


def assign_clusters(clusters, world):
    """Whole clusters to ranks, balanced by node count (largest first, each to the lightest rank).  -> list per rank of cluster indices."""
    load = [0] * world
    mine = [[] for _ in range(world)]
    for k in sorted(range(len(clusters)), key=lambda k: (-len(clusters[k]), k)):
        r = min(range(world), key=lambda r: (load[r], r))
        mine[r].append(k); load[r] += len(clusters[k])
    return [sorted(m) for m in mine]


def solve_sharded(qpn, x_init=None, dist=None, engine=None):
    """solve(qpn) with the net sharded by cluster over the ranks of `dist` (one process per GPU): every rank runs the outer loop
    on the sub-net of its own clusters -- no exchange while it sweeps, because no cluster reads another's variables -- and the
    ranks meet ONCE, at the end: the blocks of the iterate each of them owns (all-gather; gloo in the CPU tests, RCCL on GPUs)
    and the solved flags.  Per sweep nothing crosses a link; the all-gather of the whole iterate per sweep that the node-range
    sharding of the bench does (GatheredIterate) is what a parent on ANOTHER rank would need -- clusters keep parents and children
    together.  dist == None: one rank.  -> dict(solved, x_opt, clusters, owned [the cluster indices of this rank])."""
    import numpy as np
    from . import algorithm
    x0 = np.asarray(qpn.default_initialization if x_init is None else x_init, dtype=np.float64)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    clusters = net_clusters(qpn)
    mine = assign_clusters(clusters, world)[rank]
    x = x0.copy()
    ok = True
    err = None
    if mine:
        nodes = sorted(i for k in mine for i in clusters[k])
        net, var, _ = sub_net(qpn, nodes)
        ret = algorithm.solve(net, x0[var], engine=engine)
        ok = bool(ret["solved"])
        err = ret.get("error")
        owned = np.unique(np.concatenate([np.asarray(qpn.qps[i].var_indices, dtype=np.int64) for i in nodes]))
        if ok:
            loc = {int(v): k for k, v in enumerate(var.tolist())}
            x[owned] = ret["x_opt"][[loc[int(v)] for v in owned.tolist()]]
    else:
        owned = np.zeros(0, np.int64)
    if dist is not None and world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (ok, err, owned, x[owned]))      # one exchange for the whole solve
        ok = all(p[0] for p in parts)
        err = next((p[1] for p in parts if p[1]), None)
        for p in parts:
            x[p[2]] = p[3]
    return dict(solved=ok, x_opt=x if ok else None, x_fail=None if ok else x, error=err, clusters=clusters, owned=mine)
