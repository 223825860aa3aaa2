"""cpu_baseline: the reference's own sources on one core and the oracle on all host cores, on a bounded sample of the same workload."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

from benchlib.workload import *          # noqa: F401,F403 -- the workload's constants and byte models
from benchlib.workload import BENCH_PY, GOLDEN_SCALE, ROOT


def cpu_baseline(P, Q, budget_s, sample, ttr_candidates):
    """CPU path on the same workload, bounded samples.
    A (reference-faithful): 1 thread -- what MatchSuper4PCS does (super4pcs.cc:68-73), kd-tree Verify with early exit.
       kind "reference": the reference's own sources (oracle/_ref/libs4p_ref.so) run ComputeTransformation and are cut by a
       visitor exception after budget_s of RANSAC time; kind "port" (the oracle) if the prebuilt library is absent.
    B (best-effort CPU): the oracle with its CANDIDATE LOOP ONLY under `omp parallel for` on all host
       cores, as the legacy Match4PCS does by default (match4pcsBase.h:190-192) -- pair extraction and quad enumeration
       stay serial, as in the reference; also gives the per-stage split."""
    from oracle import oracle as O
    from oracle import reflib
    O.build()
    nproc = os.cpu_count() or 1

    def port_run(threads, seconds):
        om = O.Matcher(O.make_options(DELTA, OVERLAP, sample), full_counts=False, use_kdtree=True, keep_trace=False)
        om.set_threads(threads)
        om.init(P, Q)
        om.set_budget(seconds)
        t0 = time.perf_counter()
        bases = 0
        while time.perf_counter() - t0 < seconds:
            om.try_one_base()
            bases += 1
        dt = time.perf_counter() - t0
        s = om.stats()
        return {"value": s.n_verified / dt, "unit": "candidates/s", "cores": threads, "kind": "port",
                "sample": "oracle restatement, first %d base(s) of the same seeded sequence, TryCongruentSet cut after %.0f s wall "
                          "(%d candidates verified, kd-tree Verify with the reference's early exit)" % (bases, seconds, s.n_verified),
                "seconds": dt,
                "stage_seconds": {"select": s.t_select, "pairs": s.t_pairs, "quads": s.t_quads, "verify": s.t_verify}}

    if reflib.available():
        rm = reflib.RefMatcher(O.make_options(DELTA, OVERLAP, sample))
        cut, n, sec = rm.bench(P, Q, budget_s)
        a = {"value": n / max(sec, 1e-9), "unit": "candidates/s", "cores": 1, "kind": "reference",
             "sample": "reference ComputeTransformation (kd-tree Verify with early exit) on the same clouds/seed, "
                       "stopped after %.1f s of RANSAC time: %d candidates verified%s" % (sec, n, "" if cut else " (ran to completion)"),
             "seconds": sec}
    else:
        a = port_run(1, budget_s)
    a["host_cores"] = nproc
    a["note"] = ("CPU and GPU both abandon a candidate that cannot beat the best LCP so far (match4pcsBase.cc:558-560; the GPU with an "
                 "order-independent bound against the best at launch time, config.early_exit): a reported baseline, not a target")
    b = port_run(nproc, max(budget_s * 0.6, 3.0))
    b["label"] = "candidate loop only under OpenMP (pairs and quads serial, as in the reference)"
    a["openmp_all_cores"] = b
    if ttr_candidates:
        # time-to-register on the CPU (BASELINE.md section 3 "Reported"; the reference's tests allow 600 s): the whole
        # registration verifies ttr_candidates candidates (counted by the GPU run above, equal to the oracle's by the parity
        # tests); at the sampled rates that is an EXTRAPOLATION, not a run -- a measured run is in profiles/ (README there)
        a["time_to_register"] = {"measured": False, "candidates_of_the_registration": int(ttr_candidates),
                                 "extrapolated_seconds_1_core": ttr_candidates / max(a["value"], 1e-9),
                                 "extrapolated_seconds_all_cores": ttr_candidates / max(b["value"], 1e-9),
                                 "cap_seconds": 600,
                                 "note": "candidates of the whole registration / sampled candidates-per-second (the first bases are "
                                         "the slowest per candidate: no best LCP to exit early against yet); the run itself, all host "
                                         "cores, 600 s cap: tools/r3_cpu_ttr.py -> profiles/r03_cpu_time_to_register.json (307 s on 256 cores)"}
    return a
