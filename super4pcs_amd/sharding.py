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

    def __init__(self, matcher, rank=0, world=1, dist=None, device=None, producer_threads=True):
        self.m, self.rank, self.world, self.dist, self.device = matcher, rank, world, dist, device
        self.trials_done = 0
        self.local_candidates = 0
        self.terminated = False
        if producer_threads and hasattr(matcher, "set_sharding"):
            matcher.set_sharding(rank, world, True)     # base selection + octree staging on helper threads

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

    # ------------------------------------------------------------------------------------------
    def run_windows(self, n):
        """Runs n windows (n*world trials).  Returns the candidates this rank verified.

        world == 1: the engine's own pipelined Perform_N_steps.
        world  > 1: window w+1 is prepared (own device pass enqueued, other ranks' bases advanced on
        the host) BEFORE window w's result is waited for and reduced, so GPU pass, host-side base
        selection and the 8-byte collective overlap.
        """
        if self.world == 1:
            before = self.m.info().candidates_verified
            _, _, done = self.m.perform_n_steps(n)
            self.trials_done += n
            got = int(self.m.info().candidates_verified - before)
            self.local_candidates += got
            return got
        total = 0
        pending = []
        depth = getattr(self.m, "pipeline_depth", lambda: 2)()
        for _ in range(n):
            pending.append(self._prepare_window())
            while len(pending) >= depth:
                total += self._finish_window(pending.pop(0))
        while pending:
            total += self._finish_window(pending.pop(0))
        return total

    def run_window(self):
        return self.run_windows(1)

    def _prepare_window(self):
        bases, mine = [], None
        for j in range(self.world):
            if j == self.rank:
                found, base = self.m.next_base_async(True)
                mine = found
            else:
                found, base, _ = self.m.next_base(run_device=False)
            bases.append((found, base))
        return bases, mine

    def _finish_window(self, prepared):
        import torch
        from . import capi
        bases, mine_found = prepared
        r = self.m.wait_base() if mine_found else capi.BaseResult()
        thr_c = self._threshold_count()
        usable = bool(mine_found and r.n_pairs1 and r.n_pairs2 and r.n_quads)
        key = window_key(r.best_count, bool(r.has_best), usable, self.rank, thr_c)
        verified = int(r.n_verified)
        self.local_candidates += verified
        t = torch.tensor([key], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)           # the one data-path collective: 8 bytes
        win = decode_key(int(t.item()))
        if win is not None:
            w_trial, w_count, crossed = win
            if w_count > self.m.info().best_count:
                if w_trial == self.rank:
                    vals = list(r.best_transform) + list(r.best_centroid2) + list(r.centroid1) + [float(v) for v in r.best_quad] + \
                        [float(r.n_pairs1), float(r.n_pairs2), float(r.n_quads), float(r.n_verified), float(r.best_count), float(r.has_best)]
                    rec = torch.tensor(vals, dtype=torch.float64, device=self.device)
                else:
                    rec = torch.zeros(self.RECORD_FLOATS, dtype=torch.float64, device=self.device)
                self.dist.broadcast(rec, src=w_trial)                 # winner's 4x4 etc., only when the window improved
                v = rec.cpu().numpy()
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
