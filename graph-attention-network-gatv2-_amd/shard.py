"""Destination-range sharding of the GATv2 step across the GPUs of one node.

One process per GPU.  All in-edges of a destination live in one CSR row, so score, softmax,
aggregation and every per-destination gradient are local to the row's owner; only source-side
data crosses shards, once per layer and direction:

    forward  layer l : project own rows -> ALL-GATHER the PL table        -> edge forward
    backward layer l : edge backward (adds into the full gPL table)
                       -> REDUCE-SCATTER the gPL table                    -> dense backward
    end of backward  : ALL-REDUCE of the packed parameter gradients (+ loss, #correct)

Layer 0 is the exception: its input (the node features) is static, so every rank keeps the
features of ALL table rows (``ShardPlan.table_features`` / ``set_source_features``), projects the
whole layer-0 table itself and accumulates gradW_left of layer 0 from its partial gPL table — the
gradient all-reduce completes that sum.  Both layer-0 exchanges disappear: a 2-layer model moves
one table per direction and step instead of two (xGMI is the scaling limiter: a [2.45M][64] fp32
table is 627 MB against ~5 ms of per-rank compute at 8 GPUs).

The reference has no distributed code at all (single process, default stream); this is the build's
own scaling axis.  Rows are split on ``row_ptr`` so every rank holds ~E/P edges (power-law graphs
are edge-, not node-balanced).  Rank p's rows are padded to ``max_rows`` so that the exchange
tables are plain ``[P][max_rows][H*D]`` arrays: source ids are remapped ONCE to table rows
``owner*max_rows + local`` and both collectives become the fixed-count, in-place forms
(``all_gather_into_tensor`` / ``reduce_scatter_tensor``) that RCCL runs as direct xGMI exchanges.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np


@dataclass
class ShardPlan:
    world: int
    rank: int
    bounds: np.ndarray        # [world+1] global row boundaries
    max_rows: int

    @property
    def row0(self) -> int:
        return int(self.bounds[self.rank])

    @property
    def n_rows(self) -> int:
        return int(self.bounds[self.rank + 1] - self.bounds[self.rank])

    @property
    def n_table(self) -> int:
        return self.world * self.max_rows

    @property
    def table_row0(self) -> int:
        return self.rank * self.max_rows

    def to_table_ids(self, src_global: np.ndarray) -> np.ndarray:
        owner = np.searchsorted(self.bounds, src_global, side="right") - 1
        return (owner.astype(np.int64) * self.max_rows + (src_global - self.bounds[owner])).astype(np.int32)

    def from_table_ids(self, tid: np.ndarray) -> np.ndarray:
        owner = tid // self.max_rows
        return (self.bounds[owner] + tid % self.max_rows).astype(np.int32)

    def table_features(self, x_global: np.ndarray) -> np.ndarray:
        """Global [n][F] features -> the padded source-table layout [world*max_rows][F]."""
        x_global = np.asarray(x_global)
        out = np.zeros((self.n_table, x_global.shape[1]), x_global.dtype)
        for p in range(self.world):
            lo, hi = int(self.bounds[p]), int(self.bounds[p + 1])
            out[p * self.max_rows: p * self.max_rows + (hi - lo)] = x_global[lo:hi]
        return out


def edge_balanced_bounds(row_ptr: np.ndarray, world: int) -> np.ndarray:
    """Row boundaries such that every shard holds ~E/world edges and at least one row."""
    n = len(row_ptr) - 1
    e = int(row_ptr[-1])
    if world > n:
        raise ValueError("more ranks than rows")
    targets = (np.arange(1, world, dtype=np.float64) * e / world)
    cuts = np.searchsorted(row_ptr, targets, side="left").astype(np.int64)
    b = np.concatenate([[0], cuts, [n]])
    # b[0] = 0 and b[world] = n stay fixed.  First leave room on the right (a hub in the last rows puts
    # several cuts at n), then make the cuts strictly increasing; world <= n makes both satisfiable.
    for p in range(world - 1, 0, -1):
        b[p] = min(b[p], n - (world - p))
    for p in range(1, world):
        b[p] = max(b[p], b[p - 1] + 1)
    return b


def make_plan(row_ptr: np.ndarray, world: int, rank: int) -> ShardPlan:
    b = edge_balanced_bounds(np.asarray(row_ptr), world)
    return ShardPlan(world, rank, b, int(np.diff(b).max()))


def local_csr(plan: ShardPlan, row_ptr: np.ndarray, col_idx: np.ndarray):
    """-> (row_ptr_local int32[n_rows+1], col_idx_local int32 as TABLE row ids)"""
    s, t = plan.row0, plan.row0 + plan.n_rows
    e0, e1 = int(row_ptr[s]), int(row_ptr[t])
    rp = (np.asarray(row_ptr[s:t + 1], np.int64) - e0).astype(np.int32)
    ci = plan.to_table_ids(np.asarray(col_idx[e0:e1], np.int64))
    return rp, ci


class TorchComm:
    """The three exchanges over torch.distributed.  backend "nccl" (= RCCL over xGMI): in place on
    device tensors.  Any other backend (gloo, used by the CPU tests and the 2-ranks-on-one-GPU
    test): staged through host tensors, all-reduce standing in for reduce-scatter."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.native = dist.get_backend(group) == "nccl"

    def all_gather_rows(self, table, row_floats: int):
        """table: flat tensor [world*max_rows*row_floats]; rank's slice already filled."""
        t = table.view(self.world, -1)
        if self.native:
            self.dist.all_gather_into_tensor(table, t[self.rank], group=self.group)
            return
        mine = t[self.rank].cpu().contiguous()
        parts = [mine.new_empty(mine.shape) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        for p, part in enumerate(parts):
            if p != self.rank:
                t[p].copy_(part)

    def reduce_scatter_rows(self, table, row_floats: int):
        """table: every rank's full-length partial sums; afterwards rank's slice holds the total."""
        t = table.view(self.world, -1)
        if self.native:
            # out-of-place (a slice-sized staging buffer, then one D2D copy into the rank's slice):
            # an output that aliases the input is legal for RCCL but not guaranteed by every
            # torch.distributed version, and the copy is ~1/P of the table
            n = t[self.rank].numel()
            if getattr(self, "_rs_out", None) is None or self._rs_out.numel() < n:
                self._rs_out = table.new_empty(n)
            out = self._rs_out[:n]
            self.dist.reduce_scatter_tensor(out, table, group=self.group)
            t[self.rank].copy_(out)
            return
        host = table.cpu()
        self.dist.all_reduce(host, group=self.group)
        t[self.rank].copy_(host.view(self.world, -1)[self.rank])

    # ---- halo form of the two table exchanges (the library's own is csrc/gat_halo.hip: gat_comm_option GAT_COMM_HALO) ----
    def halo_setup(self, plan: "ShardPlan", col_idx_local):
        """Collective, once per graph.  col_idx_local: this shard's sources as TABLE row ids (local_csr).  Per peer q the sorted
        rows of q's slice this shard references (need[q]); the lists are all-gathered so that every rank also knows what each peer
        wants from it (send[p] = need_p[rank], as local row ids)."""
        import torch
        ref = np.unique(np.asarray(col_idx_local, np.int64))
        owner = ref // plan.max_rows
        need = [ref[owner == q] if q != self.rank else ref[:0] for q in range(self.world)]
        lists = [None] * self.world
        self.dist.all_gather_object(lists, [(n % plan.max_rows).astype(np.int64) for n in need], group=self.group)
        self._halo = dict(
            max_rows=plan.max_rows,
            need=[torch.from_numpy(n.astype(np.int64)) for n in need],                      # table rows, per owner
            send=[torch.from_numpy(np.asarray(lists[p][self.rank], np.int64)) for p in range(self.world)])   # own local rows, per requester
        rows = sum(len(lists[p][q]) for p in range(self.world) for q in range(self.world))
        self.halo_fraction = rows / max(1, self.world * (self.world - 1) * plan.max_rows)
        return self.halo_fraction

    def _pairwise(self, send_bufs, recv_bufs):
        """One send / receive per peer (host tensors on a non-RCCL backend)."""
        ops = []
        for q in range(self.world):
            if q == self.rank:
                continue
            if send_bufs[q].numel():
                ops.append(self.dist.P2POp(self.dist.isend, send_bufs[q], q, group=self.group))
            if recv_bufs[q].numel():
                ops.append(self.dist.P2POp(self.dist.irecv, recv_bufs[q], q, group=self.group))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def halo_gather_rows(self, table, row_floats: int):
        """Forward: only the rows each peer's edges reference travel; they land in the same table rows as a full all-gather's."""
        h = self._halo
        t = table.view(self.world * h["max_rows"], row_floats)
        own = t[self.rank * h["max_rows"]:(self.rank + 1) * h["max_rows"]]
        dev = table.device
        stage = (lambda x: x) if self.native else (lambda x: x.cpu())
        send = [stage(own.index_select(0, h["send"][q].to(dev))).contiguous() for q in range(self.world)]
        recv = [send[0].new_empty((len(h["need"][q]), row_floats)) for q in range(self.world)]
        self._pairwise(send, recv)
        for q in range(self.world):
            if q != self.rank and len(h["need"][q]):
                t.index_copy_(0, h["need"][q].to(dev), recv[q].to(dev))

    def halo_reduce_rows(self, table, row_floats: int):
        """Backward: the mirror — the partial sums of the rows this shard references go back to their owners, which add the arrivals
        in ascending rank order with the own partial at its rank's position (the order of a rank-ordered reduce-scatter)."""
        h = self._halo
        t = table.view(self.world * h["max_rows"], row_floats)
        own = t[self.rank * h["max_rows"]:(self.rank + 1) * h["max_rows"]]
        dev = table.device
        stage = (lambda x: x) if self.native else (lambda x: x.cpu())
        send = [stage(t.index_select(0, h["need"][q].to(dev))).contiguous() for q in range(self.world)]
        recv = [send[0].new_empty((len(h["send"][q]), row_floats)) for q in range(self.world)]
        self._pairwise(send, recv)
        acc = own.new_zeros(own.shape)
        for q in range(self.world):
            if q == self.rank:
                acc += own
            elif len(h["send"][q]):
                acc.index_add_(0, h["send"][q].to(dev), recv[q].to(dev))
        own.copy_(acc)

    def all_reduce_(self, tensor):
        if self.native:
            self.dist.all_reduce(tensor, group=self.group)
            return tensor
        host = tensor.cpu()
        self.dist.all_reduce(host, group=self.group)
        tensor.copy_(host)
        return tensor

    def barrier(self):
        self.dist.barrier(group=self.group)


class ShardedGat:
    """Host-side driver of one rank: same step as ``GatContext.forward/backward`` but with the
    exchange steps between the phases.  ``ctx`` is a GatContext (or any object with the same phase
    API — the CPU tests drive this class with a numpy stand-in to check the exchange logic);
    ``tables`` maps (which, layer) -> flat torch tensor bound to the context."""

    def __init__(self, ctx, plan: ShardPlan, comm, heads: Sequence[int], outdims: Sequence[int],
                 alloc: Callable[[int], "object"], halo: bool = False):
        self.ctx, self.plan, self.comm = ctx, plan, comm
        self.halo = bool(halo)                         # comm.halo_setup(plan, col_idx_local) must have run (collective)
        self.hd = [int(h) * int(d) for h, d in zip(heads, outdims)]
        self.L = len(self.hd)
        self.pl = []
        sb = getattr(ctx, "storage_bytes", 4)          # bf16 storage: PL rows are H*D*2 bytes
        for l in range(self.L):
            t = alloc(plan.n_table * self.hd[l] * sb // 4)
            ctx.bind_table(0, l, t.data_ptr(), t.numel() * 4)
            self.pl.append(t)
        self.gpl = alloc(plan.n_table * max(self.hd))
        ctx.bind_table(1, 0, self.gpl.data_ptr(), self.gpl.numel() * 4)
        # staging buffer of the end-of-step all-reduce: packed gradients + [loss, correct lo, correct hi]
        self._packed = alloc(ctx.n_params + 3)
        self.grads = self._packed[: ctx.n_params]
        self._alloc, self._prev = alloc, None
        # layer 0 of a context with replicated input needs no exchange (see module docstring)
        self.exchange = [bool(ctx.layer_exchange(l)) if hasattr(ctx, "layer_exchange") else True
                         for l in range(self.L)]

    def _forward_phases(self):
        for l in range(self.L):
            self.ctx.layer_project(l)
            if self.exchange[l]:
                sb = getattr(self.ctx, "storage_bytes", 4)
                if self.halo:
                    self.comm.halo_gather_rows(self.pl[l], self.hd[l] * sb // 4)
                else:
                    self.comm.all_gather_rows(self.pl[l], self.hd[l])
            self.ctx.layer_forward_edges(l)

    def _backward_phases(self):
        self.ctx.head_backward()
        for l in range(self.L - 1, -1, -1):
            self.ctx.layer_backward_edges(l)
            if self.exchange[l]:
                if self.halo:
                    self.comm.halo_reduce_rows(self.gpl[: self.plan.n_table * self.hd[l]], self.hd[l])
                else:
                    self.comm.reduce_scatter_rows(self.gpl[: self.plan.n_table * self.hd[l]], self.hd[l])
            self.ctx.layer_backward_dense(l)

    def forward(self):
        """-> (global loss sum, global #correct); synchronises to return them (use step() in a loop)."""
        import torch
        self._forward_phases()
        loss, correct = self.ctx.head_forward()
        s = torch.tensor([loss, float(correct)], dtype=torch.float64)
        if getattr(self.comm, "native", False):
            s = s.to(self.gpl.device)
        self.comm.all_reduce_(s)
        return float(s[0]), int(round(float(s[1])))

    def _set_aside(self):
        """The context's gradient buffer accumulates until zero_grad (E:1631-1633) and already holds
        REDUCED sums of earlier steps: set them aside and run this step into a zeroed buffer, so that
        the all-reduce sums only this step's per-shard contributions (else earlier steps would be
        multiplied by the world size on every further step)."""
        n = self.grads.numel()
        if self._prev is None:
            self._prev = self._alloc(n)
        self.ctx.grads_export(self._prev.data_ptr(), n)
        self.ctx.zero_grad()

    def backward(self):
        self._set_aside()
        self._backward_phases()
        # W/a/Wo gradients: one packed buffer (tens of KB: latency-bound, a single all-reduce)
        self.ctx.grads_export(self.grads.data_ptr(), self.grads.numel())
        self.comm.all_reduce_(self.grads)
        self.grads.add_(self._prev)
        self.ctx.grads_import(self.grads.data_ptr(), self.grads.numel())

    def step(self):
        """forward + backward with no host synchronisation until the end: loss and #correct ride in
        the tail of the packed-gradient all-reduce.  -> (global loss sum, global #correct)"""
        n = self.grads.numel()
        self._set_aside()
        self._forward_phases()
        self.ctx.head_forward(want_loss=False)
        self._backward_phases()
        self.ctx.grads_export(self._packed.data_ptr(), n)
        self.ctx.result_export(self._packed.data_ptr() + 4 * n)
        self.comm.all_reduce_(self._packed)
        self.grads.add_(self._prev)
        self.ctx.grads_import(self._packed.data_ptr(), n)
        tail = self._packed[n:].cpu()
        return float(tail[0]), int(round(float(tail[1]) + 4096.0 * float(tail[2])))
