"""Sharding of RANSAC bases over the GPUs of one node, one process per GPU.

Every rank walks the *same* sequence of bases (same RNG, same pair-octree state); the rank
that owns a trial runs the fused device pass, the others only advance host state.  After a
window of `world` consecutive trials (one per rank) a single all-reduce(MAX) of one packed
64-bit key over RCCL/xGMI selects the winner exactly as the sequential reference would
(match4pcsBase.hpp:467-484: the first strictly greater LCP wins; :255: stop at the first
trial whose best LCP exceeds the terminate threshold); the winner's 4x4 travels by one
broadcast, only when the window improved the best LCP.

The collective is 8 bytes: latency-bound, xGMI link bandwidth is irrelevant.
"""
import ctypes as C

import numpy as np

_CROSS_BIT = 1 << 62


def window_key(count, has_best, usable, trial_in_window, threshold_count):
    """Packs one trial's outcome so that max() over the window reproduces the sequential reference.

    usable: pairs1, pairs2 and quads all non-empty (otherwise TryOneBase returned before TryCongruentSet).
    A trial whose count exceeds the terminate threshold outranks everything and, among those, the
    earliest wins; otherwise higher count wins and ties go to the earliest trial.
    """
    if not (has_best and usable):
        return 0
    inv_t = 0xFFFF - trial_in_window
    if count > threshold_count:
        return _CROSS_BIT | (inv_t << 32) | int(count)
    return (int(count) << 16) | inv_t


def decode_key(key):
    """-> (trial_in_window, count, crossed) or None"""
    if key == 0:
        return None
    if key & _CROSS_BIT:
        return 0xFFFF - ((key >> 32) & 0xFFFF), key & 0xFFFFFFFF, True
    return 0xFFFF - (key & 0xFFFF), key >> 16, False


class ShardedRansac:
    """Drives a matcher (super4pcs_amd.capi.Matcher or a test double with the same three methods
    next_base / commit / info) in windows of `world` trials."""

    RECORD_FLOATS = 16 + 3 + 3 + 4 + 6   # T, c2, c1, quad, (m1, m2, K, C, count, has_best)

    def __init__(self, matcher, rank=0, world=1, dist=None, device=None):
        self.m, self.rank, self.world, self.dist, self.device = matcher, rank, world, dist, device
        self.trials_done = 0
        self.local_candidates = 0
        self.terminated = False

    def _threshold_count(self):
        info = self.m.info()
        # lcp > terminate_threshold  <=>  count/n > thr (float): find the largest count that does not cross
        n = info.n_sampled_q
        thr = np.float32(self.m.opt.terminate_threshold)
        c = int(np.floor(float(thr) * n))
        while c < n and not (np.float32(c + 1) / np.float32(n) > thr):
            c += 1
        while c >= 0 and (np.float32(c) / np.float32(n) > thr):
            c -= 1
        return c

    def run_window(self):
        """One window = `world` consecutive trials.  Returns the number of candidates this rank verified."""
        if self.world == 1:                       # plain TryOneBase
            found, base, r = self.m.next_base(run_device=True)
            self.terminated = self.m.commit(found, base, r) or self.terminated
            self.trials_done += 1
            self.local_candidates += int(r.n_verified)
            return int(r.n_verified)
        import torch
        thr_c = self._threshold_count()
        mine = None
        bases = []
        for j in range(self.world):
            found, base, r = self.m.next_base(run_device=(j == self.rank))
            bases.append((found, base))
            if j == self.rank:
                mine = (found, base, r)
        found, base, r = mine
        usable = bool(found and r.n_pairs1 and r.n_pairs2 and r.n_quads)
        key = window_key(r.best_count, bool(r.has_best), usable, self.rank, thr_c)
        self.local_candidates += int(r.n_verified)
        verified = int(r.n_verified)
        if self.world > 1:
            t = torch.tensor([key], dtype=torch.int64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            key = int(t.item())
        win = decode_key(key)
        if win is not None:
            w_trial, w_count, crossed = win
            cur = self.m.info().best_count
            if w_count > cur:
                rec = torch.zeros(self.RECORD_FLOATS, dtype=torch.float64, device=self.device)
                if w_trial == self.rank:
                    vals = list(r.best_transform) + list(r.best_centroid2) + list(r.centroid1) + [float(v) for v in r.best_quad] + \
                        [float(r.n_pairs1), float(r.n_pairs2), float(r.n_quads), float(r.n_verified), float(r.best_count), float(r.has_best)]
                    rec = torch.tensor(vals, dtype=torch.float64, device=self.device)
                if self.world > 1:
                    self.dist.broadcast(rec, src=w_trial)
                v = rec.cpu().numpy()
                from . import capi
                wr = capi.BaseResult()
                for i in range(16):
                    wr.best_transform[i] = np.float32(v[i])
                for i in range(3):
                    wr.best_centroid2[i] = np.float32(v[16 + i]); wr.centroid1[i] = np.float32(v[19 + i])
                for i in range(4):
                    wr.best_quad[i] = int(v[22 + i])
                wr.n_pairs1, wr.n_pairs2, wr.n_quads, wr.n_verified = int(v[26]), int(v[27]), int(v[28]), int(v[29])
                wr.best_count, wr.has_best = int(v[30]), int(v[31])
                ok = self.m.commit(True, bases[w_trial][1], wr)
                self.terminated = self.terminated or ok or crossed
        self.trials_done += self.world
        return verified
