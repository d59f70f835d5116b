"""Multi-GPU driver: persons of each round sharded over the ranks of one node, one process per GPU.

The path shards *within a round* (SURVEY.md section 8e): bidders are independent given the common price
vector (Jacobi auction, reference auction_.pyx:339-365 reads prices that are only written at :397).  All
solver state is replicated; rank r bids for the positions [K*r/W, K*(r+1)/W) of the unassigned list, and
the only exchange step of a round is the per-object arg-max of the bids (auction_.pyx:375-385):

    round_bid       local bids + local per-object maximum (int64 keys = bid bits + 1)
    all_reduce MAX  over best_key[M]          (RCCL over xGMI; `nccl` backend of torch.distributed)
    round_tiebreak  positions of local bidders that hold the global maximum
    all_reduce MIN  over best_pos[M]          (earliest list position wins equal bids, strict '>' of :379)
    round_apply     assignment + list compaction, run redundantly -- and deterministically -- by every
                    rank, which is what keeps the replicas identical (this *is* the price broadcast)

Only the big rounds (K >= shard_min_K = 0.3 N: a handful per eps-phase, where the bid phase is a
bandwidth-bound CSR scan) are sharded and exchanged.  All smaller rounds are replicated: every rank bids for
every list position (grid kernels) or runs the persistent tail kernel (K <= tail threshold; > 98 % of all
rounds), with no communication -- a dense all-reduce per small round would cost far more than the round, and
the replicas stay bit-identical because every step is deterministic.  The driver is written against a small backend interface so that its control flow and
collective sequence are covered by world_size-2 `gloo` tests on CPU tensors (tests/test_dist_gloo.py).
"""
import torch
import torch.distributed as dist


class GpuBackend:
    """One AuctionSolver handle (created with shard=(rank, world)) + torch views of its exchange buffers."""

    def __init__(self, solver):
        self.s = solver
        key_ptr, pos_ptr, m = solver.exchange_buffers()
        dev = torch.device("cuda", torch.cuda.current_device())
        self.best_key = _alias(key_ptr, m, "<i8", dev)
        self.best_pos = _alias(pos_ptr, m, "<i4", dev)
        # all kernels of the handle go to torch's current stream so that they order with the collectives
        solver.set_stream(torch.cuda.current_stream().cuda_stream)
        self.thr = solver.tail_threshold
        self.rounds_per_sync = solver.rounds_per_sync
        self.shard_min_K = solver.shard_min_K

    def status(self):
        st = self.s.status()
        return int(st.K), int(st.its)

    @property
    def max_iter(self):
        return int(self.s._opts.max_iter) if self.s._opts.max_iter >= 1 else 1

    def round_bid(self):
        self.s.round_bid()

    def round_tiebreak(self):
        self.s.round_tiebreak()

    def round_apply(self):
        self.s.round_apply()

    def run_tail(self):
        self.s.run_tail()

    def phase_end(self):
        return self.s.phase_end()

    def finish(self):
        return self.s.finish()


class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _alias(ptr, n, typestr, dev):
    """torch tensor aliasing library-owned device memory (no copy)."""
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=dev)


def solve_sharded(solver_or_backend, group=None):
    """AuctionSolver.solve() (reference auction_.pyx:268-306) over all ranks of `group`.

    Every rank passes its own handle / backend built on the same input; every rank returns the same
    person_to_object array.  Control decisions are taken from replicated state, so all ranks issue the
    same sequence of collectives.
    """
    b = solver_or_backend if hasattr(solver_or_backend, "round_bid") and hasattr(solver_or_backend, "best_key") \
        else GpuBackend(solver_or_backend)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    while True:
        while True:  # rounds of one eps-phase
            K, its = b.status()
            if K == 0 or its >= b.max_iter:
                break
            if K >= b.shard_min_K and K > b.thr:
                # a big round: bidders sharded over the ranks, per-object arg-max exchanged (K is exact here,
                # so the device-side decision "K >= shard_min_K" is the same on every rank)
                b.round_bid()
                if multi:
                    dist.all_reduce(b.best_key, op=dist.ReduceOp.MAX, group=group)
                b.round_tiebreak()
                if multi:
                    dist.all_reduce(b.best_pos, op=dist.ReduceOp.MIN, group=group)
                b.round_apply()
            elif K > b.thr:
                # K never grows inside a phase: from here on every rank bids for everybody (replicated,
                # deterministic), no exchange; several rounds per status read
                for _ in range(b.rounds_per_sync):
                    b.round_bid()
                    b.round_tiebreak()
                    b.round_apply()
            else:
                b.run_tail()
        if b.phase_end():
            break
    sol = b.finish()
    return sol
