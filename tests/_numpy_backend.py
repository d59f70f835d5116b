"""Test-only stand-in for the per-rank GPU work of sslap_amd.dist (numpy on the host), so that the
driver's control flow, sharding and collective sequence can be exercised with the `gloo` backend on CPU.
It restates one sharded round the way the kernels do it (keys = bid bits + 1, MAX then MIN exchange,
per-object apply, k-th hole <- k-th mover compaction).  Small inputs only (Python loops)."""
import numpy as np
import torch

POS_NONE = np.int32(0x7FFFFFFF)


class NumpyBackend:
    def __init__(self, loc, val, problem, rank, world, max_iter=10**8, eps_start=0.0, thr=0, rounds_per_sync=2,
                 shard_min_K=0):
        loc = loc.astype(np.int32)
        self.N, self.M = int(loc[:, 0].max()) + 1, int(loc[:, 1].max()) + 1
        self.row_ptr = np.searchsorted(loc[:, 0], np.arange(self.N + 1)).astype(np.int64)
        self.col = loc[:, 1].copy()
        self.maximize = problem == "max"
        self.val = val.copy() if self.maximize else val * -1
        C = np.float32(np.abs(self.val).max())
        self.eps = np.float32(np.float64(C) / 2.0)
        self.target_eps = np.float32(1.0 / self.N)
        self.theta = np.float32(0.15)
        if eps_start > 0:
            self.eps = np.float32(eps_start)
        self.p = np.zeros(self.M)
        self.p2o = np.full(self.N, -1, np.int32)
        self.o2p = np.full(self.M, -1, np.int32)
        self.U = np.arange(self.N, dtype=np.int32)
        self.K, self.its, self.nreductions = self.N, 0, 0
        self.max_iter = max(1, int(max_iter))
        self.rank, self.world, self.thr, self.rounds_per_sync = rank, world, thr, rounds_per_sync
        self.shard_min_K = shard_min_K
        self.best_key = torch.zeros(self.M, dtype=torch.int64)
        self.best_pos = torch.full((self.M,), int(POS_NONE), dtype=torch.int32)
        self._bk, self._bp = self.best_key.numpy(), self.best_pos.numpy()  # shared memory views
        self.collectives = 0

    def _live(self):
        return self.K > self.thr and self.K > 0 and self.its < self.max_iter

    def status(self):
        return self.K, self.its

    def _shard(self):
        if self.K < self.shard_min_K:  # small rounds are replicated, not sharded
            return 0, self.K
        return (self.K * self.rank) // self.world, (self.K * (self.rank + 1)) // self.world

    def round_bid(self):
        if not self._live():
            return
        lo, hi = self._shard()
        self.bid_key = np.zeros(self.K, np.int64)
        self.bid_obj = np.full(self.K, -1, np.int32)
        for n in range(lo, hi):
            i = self.U[n]
            s, e = self.row_ptr[i], self.row_ptr[i + 1]
            v = self.val[s:e] - self.p[self.col[s:e]]
            vb = v.max()
            g = np.nonzero(v == vb)[0][-1]           # last maximum (auction_.pyx:351)
            rest = np.delete(v, g)
            w = rest.max() if rest.size else -np.inf
            bid = (self.val[s + g] - w) + np.float64(self.eps)
            key = np.array([bid]).view(np.int64)[0] + 1
            self.bid_key[n], self.bid_obj[n] = key, self.col[s + g]
            self._bk[self.col[s + g]] = max(self._bk[self.col[s + g]], key)

    def round_tiebreak(self):
        if not self._live():
            return
        lo, hi = self._shard()
        for n in range(lo, hi):
            j = self.bid_obj[n]
            if self.bid_key[n] == self._bk[j]:
                self._bp[j] = min(self._bp[j], n)

    def round_apply(self):
        if not self._live():
            return
        holes = 0
        for j in np.nonzero(self._bp != POS_NONE)[0]:
            n = self._bp[j]
            i = self.U[n]
            self.p[j] = np.array([self._bk[j] - 1]).view(np.float64)[0]
            prev = self.o2p[j]
            if prev != -1:
                self.p2o[prev] = -1
                self.U[n] = prev
            else:
                self.U[n] = -1
                holes += 1
            self.p2o[i], self.o2p[j] = j, i
            self._bk[j], self._bp[j] = 0, POS_NONE
        Kn = self.K - holes
        hl = [n for n in range(Kn) if self.U[n] == -1]
        mv = [n for n in range(Kn, self.K) if self.U[n] != -1]
        assert len(hl) == len(mv)
        for h, m in zip(hl, mv):
            self.U[h], self.U[m] = self.U[m], -1
        self.K = Kn
        self.its += 1

    def run_tail(self):  # the gloo test runs with thr = 0: every round is an exchanged round
        pass

    def _ece(self):
        if self.K > 0:
            return False
        for i in range(self.N):
            s, e = self.row_ptr[i], self.row_ptr[i + 1]
            j = self.p2o[i]
            g = np.nonzero(self.col[s:e] == j)[0][-1]
            lhs = (self.val[s + g] - self.p[j]) + 1e-7
            if (lhs < (self.val[s:e] - self.p[self.col[s:e]]) - np.float64(self.target_eps)).any():
                return False
        return True

    def phase_end(self):
        if self.its >= self.max_iter:
            return True
        if self.K == 0:
            if self._ece() or self.eps < self.target_eps:
                return True
            self.eps = np.float32(self.eps * self.theta)
            self.p2o[:], self.o2p[:] = -1, -1
            self.U = np.arange(self.N, dtype=np.int32)
            self.K = self.N
            self.nreductions += 1
        return False

    def finish(self):
        return self.p2o.copy()
