"""Multi-GPU front-end: persons of each round sharded over the ranks of one node, one process per GPU.

The whole sharded solve -- shard / replicate decision per round, the two all-reduces of a sharded round, loop
control -- runs inside libmisslap.so (`misslap_solve_sharded`, csrc/host_comm.hpp); this module only creates the
communicator and calls it.  The path shards *within a round* (SURVEY.md section 8e): bidders are independent given
the common price vector (Jacobi auction, reference auction_.pyx:339-365 reads prices that are only written at :397).
All solver state is replicated; rank r bids for the positions [K*r/W, K*(r+1)/W) of the unassigned list, and the
only exchange step of a round is the per-object arg-max of the bids (auction_.pyx:375-385):

    k_bid / k_bid_tiled   local bids + local per-object maximum (int64 keys = bid bits + 1)
    all-reduce MAX        over best_key[M]     (RCCL over xGMI, on the solver's stream)
    k_tiebreak            positions of local bidders that hold the global maximum
    all-reduce MIN        over best_pos[M]     (earliest list position wins equal bids, strict '>' of :379)
    k_apply + compaction  run redundantly -- and deterministically -- by every rank, which keeps the replicas
                          identical (this *is* the price broadcast)

Only the big rounds (K >= shard_min_K = 0.3 N: a handful per eps-phase, where the bid phase is a bandwidth-bound CSR
scan) are sharded and exchanged.  All smaller rounds are replicated: every rank bids for every list position (grid
kernels) or runs the persistent tail kernel (> 98 % of all rounds), with no communication.

Communicators:
    Comm.rccl(rank, world, device, share_id)   RCCL; `share_id(id_bytes_or_None) -> id_bytes` hands rank 0's
                                               128-byte id to every rank (any transport the application has)
    Comm.from_torch_distributed(device)        the same, the id travels through torch.distributed
    Comm.custom(rank, world, max_cb, min_cb)   caller-provided all-reduces (ptr, count, stream) -- other transports
    Comm.torch_collectives(group=None)         torch.distributed's own all-reduces on the device buffers (nccl backend =
                                               the RCCL inside torch): the transport bench.py falls back to
    Comm.gloo_staged(group=None)               rehearsal on ONE GPU: several ranks share cuda:0, the exchange is
                                               staged through the host with gloo (RCCL needs a GPU per rank)
    Comm.in_process(rank, group)               rehearsal with the ranks as THREADS of one process (ThreadGroup): what a
                                               one-GPU box allows beyond the handful of processes it admits on its card
                                               (W = 8); the exchange is staged through the host and reduced with numpy
"""
import ctypes as C

from . import _lib

RCCL_ID_BYTES = 128


class Comm:
    """Owner of a misslap_comm* (include/misslap.h)."""

    def __init__(self, handle, rank, world, keep=()):
        self._c, self.rank, self.world, self._keep = handle, rank, world, keep

    def __del__(self):
        c = getattr(self, "_c", None)
        if c:
            try:
                _lib.load().misslap_comm_destroy(c)
            except Exception:
                pass
            self._c = None

    def info(self):
        """dict(kind='rccl' | 'custom', rank, world, transport_ranks): transport_ranks is what the transport itself
        reports -- ncclCommCount for RCCL."""
        k, r, w, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.load().misslap_comm_info(self._c, C.byref(k), C.byref(r), C.byref(w), C.byref(n)))
        return dict(kind="rccl" if k.value else "custom", rank=r.value, world=w.value, transport_ranks=n.value)

    @staticmethod
    def unique_id():
        """Rank 0: a fresh RCCL id (128 bytes) to hand to every rank."""
        buf = C.create_string_buffer(RCCL_ID_BYTES)
        _lib.check(_lib.load().misslap_rccl_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, rank, world, device, share_id):
        uid = share_id(cls.unique_id() if rank == 0 else None)
        h = C.c_void_p()
        _lib.check(_lib.load().misslap_comm_init_rccl(C.byref(h), uid, int(rank), int(world), int(device)))
        return cls(h, rank, world)

    @classmethod
    def from_torch_distributed(cls, device, group=None):
        """RCCL communicator for the ranks of an initialised torch.distributed group (any backend: it only carries
        the 128-byte id; the solve's collectives are the library's own)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def share(uid):
            box = [uid]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            return box[0]
        return cls.rccl(rank, world, device, share)

    @classmethod
    def custom(cls, rank, world, allreduce_max_i64, allreduce_min_i32):
        """`allreduce_*(ptr, count, stream)`: in-place all-reduce of a buffer of the round operations in use (device
        memory for a GPU handle); exceptions are reported as a failed solve."""
        def wrap(fn):
            def cb(_ctx, ptr, count, stream):
                try:
                    fn(int(ptr), int(count), stream)
                    return 0
                except Exception:  # noqa: BLE001 -- must not propagate through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1
            return _lib._AR(cb)
        ops = _lib.CommOps()
        ops.struct_size = C.sizeof(_lib.CommOps)
        ops.rank, ops.world = int(rank), int(world)
        ops.allreduce_max_i64, ops.allreduce_min_i32 = wrap(allreduce_max_i64), wrap(allreduce_min_i32)
        h = C.c_void_p()
        _lib.check(_lib.load().misslap_comm_init_custom(C.byref(h), C.byref(ops)))
        return cls(h, rank, world, keep=(ops,))

    @classmethod
    def gloo_staged(cls, group=None):
        """Several ranks on ONE GPU (rehearsal / tests): device buffers are staged through the host and reduced with
        the gloo backend of torch.distributed."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def staged(op, typestr):
            def fn(ptr, count, _stream):
                t = torch.as_tensor(_DevArray(ptr, count, typestr), device=torch.device("cuda", torch.cuda.current_device()))
                torch.cuda.synchronize()  # the solver's stream has produced the buffer
                h = t.cpu()
                dist.all_reduce(h, op=op, group=group)
                t.copy_(h)
                torch.cuda.synchronize()  # ... and may read it again
            return fn
        return cls.custom(rank, world, staged(dist.ReduceOp.MAX, "<i8"), staged(dist.ReduceOp.MIN, "<i4"))

    @classmethod
    def torch_collectives(cls, group=None):
        """The exchange through torch.distributed's OWN collectives on the device buffers, ordered on the solver's
        stream (backend nccl = the RCCL inside torch, one GPU per rank; gloo takes CUDA tensors as well, which is how the
        tests run it on one GPU).  Second transport for a node on which the library's own RCCL communicator cannot be
        created (bench.py falls back to it and says so in its line); same two callbacks as every custom communicator."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def coll(op, typestr):
            def fn(ptr, count, stream):
                # (the collective is enqueued behind the bid kernel on the solver's stream and that stream waits for
                # its result: torch orders a collective with the CURRENT stream)
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0))):
                    t = torch.as_tensor(_DevArray(ptr, count, typestr), device=torch.device("cuda", torch.cuda.current_device()))
                    dist.all_reduce(t, op=op, group=group)
            return fn
        return cls.custom(rank, world, coll(dist.ReduceOp.MAX, "<i8"), coll(dist.ReduceOp.MIN, "<i4"))

    @classmethod
    def in_process(cls, rank, group):
        """Rank `rank` of a ThreadGroup: the ranks are threads of ONE process that share a GPU (rehearsal / tests; a
        one-GPU box admits only a handful of processes on its card, so W = 8 goes this way).  The exchange is the same
        pair of callbacks as gloo_staged, reduced with numpy: the device buffer is copied out ON THE SOLVER'S STREAM
        (ordered behind the bid kernel), reduced over the threads, copied back on that stream."""
        import numpy as np
        import torch

        def staged(op, typestr):
            def fn(ptr, count, stream):
                st = torch.cuda.ExternalStream(int(stream or 0))
                with torch.cuda.stream(st):
                    t = torch.as_tensor(_DevArray(ptr, count, typestr), device=torch.device("cuda", torch.cuda.current_device()))
                    h = t.cpu().numpy()  # (synchronous: the stream has produced the buffer)
                    r = group.all_reduce(rank, h, op)
                    t.copy_(torch.from_numpy(np.ascontiguousarray(r)))
                    st.synchronize()  # the host array must outlive the copy
            return fn
        return cls.custom(rank, group.world, staged("max", "<i8"), staged("min", "<i4"))


class ThreadGroup:
    """The ranks of an in-process rehearsal (one thread per rank): barrier, all-reduce of host arrays, all-gather of
    Python objects.  Every wait is bounded: a rank that fails breaks the barrier and the others raise instead of
    hanging."""

    def __init__(self, world, timeout_s=600.0):
        import threading
        self.world, self.timeout_s = int(world), float(timeout_s)
        self._bar = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self._result = None

    def barrier(self):
        self._bar.wait(self.timeout_s)

    def abort(self):
        self._bar.abort()

    def all_gather(self, rank, obj):
        self._slots[rank] = obj
        self._bar.wait(self.timeout_s)
        out = list(self._slots)
        self._bar.wait(self.timeout_s)  # (everybody has read the slots before the next call rewrites them)
        return out

    def all_reduce(self, rank, arr, op):
        """op: 'max' | 'min' | 'sum' over numpy arrays (or scalars) of equal shape; every rank gets the result."""
        import numpy as np
        parts = self.all_gather(rank, arr)
        f = {"max": np.maximum, "min": np.minimum, "sum": np.add}[op]
        out = parts[0]
        for x in parts[1:]:  # (every rank reduces in the same order: identical results)
            out = f(out, x)
        return out


class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def solve_sharded(solver, comm=None):
    """AuctionSolver.solve() (reference auction_.pyx:268-306) over all ranks of `comm`.  Every rank passes its own
    solver (created on the same input with shard=(rank, world)); every rank returns the same person_to_object."""
    return solver.solve_sharded(comm)
