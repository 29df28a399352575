"""P virtual ranks as P threads of ONE process, exchanging through shared tensors (TEST
INFRASTRUCTURE; SURVEY §8e "single-GPU box can run P virtual ranks ... with a loopback comm").

Same interface as ``shard.TorchComm``; no transport: every exchange publishes the rank's tensor,
meets the others at a barrier and reads their slices directly.  Sums run in ascending rank order on
every rank, so all ranks hold bitwise identical results."""
import threading


class Hub:
    def __init__(self, world, sync=None):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.sync = sync or (lambda: None)      # device-wide sync for GPU tensors on per-thread streams


class LoopbackComm:
    native = False

    def __init__(self, hub: Hub, rank: int):
        self.hub, self.rank, self.world = hub, rank, hub.world

    def _publish(self, tensor):
        self.hub.slots[self.rank] = tensor
        self.hub.sync()
        self.hub.barrier.wait()

    def _done(self):
        self.hub.sync()
        self.hub.barrier.wait()

    def all_gather_rows(self, table, row_floats):
        self._publish(table)
        t = table.view(self.world, -1)
        for p in range(self.world):
            if p != self.rank:
                t[p].copy_(self.hub.slots[p].view(self.world, -1)[p])
        self._done()

    def reduce_scatter_rows(self, table, row_floats):
        self._publish(table)
        acc = self.hub.slots[0].view(self.world, -1)[self.rank].clone()
        for p in range(1, self.world):
            acc += self.hub.slots[p].view(self.world, -1)[self.rank]
        self._done()                               # everyone has read every slice
        table.view(self.world, -1)[self.rank].copy_(acc)
        self._done()

    def all_reduce_(self, tensor):
        self._publish(tensor)
        acc = self.hub.slots[0].clone()
        for p in range(1, self.world):
            acc += self.hub.slots[p]
        self._done()
        tensor.copy_(acc)
        self._done()
        return tensor

    def barrier(self):
        self.hub.barrier.wait()
