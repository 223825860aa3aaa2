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


class _KeySlot:
    """One in-flight 8-byte reduction.  On a GPU rank the key goes pinned host -> device -> all_reduce -> pinned host
    with non-blocking copies and an event, so the launch thread never blocks on the collective it has just issued;
    with CPU tensors (gloo: tests, single-GPU dry runs) the all_reduce itself is synchronous."""

    def __init__(self, device):
        import torch
        self.on_gpu = device is not None and getattr(device, "type", str(device)) == "cuda"
        if self.on_gpu:
            self.host = torch.zeros(1, dtype=torch.int64).pin_memory()
            self.dev = torch.zeros(1, dtype=torch.int64, device=device)
            self.event = torch.cuda.Event()
        else:
            self.host = torch.zeros(1, dtype=torch.int64)
            self.dev = self.host
        self.view = self.host.numpy()

    def post(self, dist, key):
        self.view[0] = key
        if self.on_gpu:
            self.dev.copy_(self.host, non_blocking=True)
        dist.all_reduce(self.dev, op=dist.ReduceOp.MAX)             # the one data-path collective: 8 bytes
        if self.on_gpu:
            self.host.copy_(self.dev, non_blocking=True)
            self.event.record()

    def result(self):
        if self.on_gpu:
            self.event.synchronize()
        return int(self.view[0])


class ShardedRansac:
    """Drives a matcher (super4pcs_amd.capi.Matcher or a test double with the same methods
    next_base / next_base_async / wait_base / commit / info) in windows of `world` trials."""

    RECORD_FLOATS = 16 + 3 + 3 + 4 + 6   # T, c2, c1, quad, (m1, m2, K, C, count, has_best)

    def __init__(self, matcher, rank=0, world=1, dist=None, device=None, producer_threads=True, force_windows=False):
        self.m, self.rank, self.world, self.dist, self.device = matcher, rank, world, dist, device
        self.force_windows = force_windows      # run the windowed (collective) path even for world == 1 (tests)
        self.trials_done = 0
        self.local_candidates = 0
        self.terminated = False
        self._thr_count = None
        self._best_count = None
        self._slots = None
        self._slot_rr = 0
        if hasattr(matcher, "set_sharding"):
            # base selection + octree staging on helper threads: True / False force it, "auto" leaves it to the engine
            # (threads where they pay: 4 or more ranks, or base selection on the device)
            matcher.set_sharding(rank, world, 2 if producer_threads == "auto" else int(bool(producer_threads)))

    def _threshold_count(self):
        """Largest inlier count that does NOT cross the terminate threshold (fixed once the clouds are sampled)."""
        if self._thr_count is None:
            info = self.m.info()
            # lcp > terminate_threshold  <=>  count/n > thr (float): find the largest count that does not cross
            n = info.n_sampled_q
            thr = np.float32(self.m.opt.terminate_threshold)
            c = int(np.floor(float(thr) * n))
            while c < n and not (np.float32(c + 1) / np.float32(n) > thr):
                c += 1
            while c >= 0 and (np.float32(c) / np.float32(n) > thr):
                c -= 1
            self._thr_count = c
        return self._thr_count

    # ------------------------------------------------------------------------------------------
    def run_windows(self, n):
        """Runs n windows (n*world trials).  Returns the candidates this rank verified.

        world == 1: the engine's own pipelined Perform_N_steps.
        world  > 1: a three-stage software pipeline per rank.  Window w+d is *prepared* (own device pass enqueued,
        the other ranks' bases advanced on the host) while the device passes of windows w+1..w+d-1 are in flight
        (d = lanes of the context); window w's result is then waited for and its 8-byte key *posted* to the
        all-reduce, and only after that is the reduction of window w-1 *completed* (read back, winner committed).
        GPU pass, host-side base selection and the collective therefore overlap, and the launch thread never waits
        on a collective it has just issued.  Every rank executes the same sequence of collectives.
        After the terminate threshold is crossed the remaining in-flight windows are drained without being
        committed (the sequential loop would not have run them, match4pcsBase.hpp:255).
        """
        if self.world == 1 and not self.force_windows:
            before = self.m.info().candidates_verified
            _, _, done = self.m.perform_n_steps(n)
            self.trials_done += n
            got = int(self.m.info().candidates_verified - before)
            self.local_candidates += got
            return got
        if self._slots is None:
            self._slots = [_KeySlot(self.device), _KeySlot(self.device)]
        self._best_count = int(self.m.info().best_count)
        total = 0
        prepared, posted = [], None
        depth = getattr(self.m, "pipeline_depth", lambda: 2)()
        for _ in range(n):
            prepared.append(self._prepare_window())
            if len(prepared) >= depth:
                nxt = self._post_window(prepared.pop(0))
                if posted is not None:
                    total += self._complete_window(posted)
                posted = nxt
        while prepared:
            nxt = self._post_window(prepared.pop(0))
            if posted is not None:
                total += self._complete_window(posted)
            posted = nxt
        if posted is not None:
            total += self._complete_window(posted)
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

    def _post_window(self, prepared):
        from . import capi
        bases, mine_found = prepared
        r = self.m.wait_base() if mine_found else capi.BaseResult()
        verified = int(r.n_verified)
        self.local_candidates += verified
        slot = None
        if not self.terminated:
            usable = bool(mine_found and r.n_pairs1 and r.n_pairs2 and r.n_quads)
            key = window_key(r.best_count, bool(r.has_best), usable, self.rank, self._threshold_count())
            slot = self._slots[self._slot_rr]
            self._slot_rr ^= 1
            slot.post(self.dist, key)
        return bases, r, verified, slot

    def _complete_window(self, posted):
        import torch
        from . import capi
        bases, r, verified, slot = posted
        self.trials_done += self.world
        if slot is None or self.terminated:
            return verified
        win = decode_key(slot.result())
        if win is not None:
            w_trial, w_count, crossed = win
            if w_count > self._best_count:
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
                self._best_count = int(wr.best_count)
                self.terminated = self.terminated or ok or crossed
        return verified
