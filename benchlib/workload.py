"""The benchmarked workload (BASELINE.json configs[2]) and the byte models of the roofline figures."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_POINTS = 1_000_000
DELTA = 0.004
OVERLAP = 0.5
SAMPLE = 2000
SEED = 20140814
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md: aggregate L2 bandwidth, ~34.5 TB/s
N_SIMDS = 256 * 4              # 256 CUs x 4 SIMDs
MAX_PAIRS, MAX_QUADS = 2 << 20, 8 << 20        # per lane: ~10 x what the workload's largest base needs (a base that needed more would grow its lane)


GOLDEN_SCALE = os.path.join(ROOT, "tests", "golden", "scale_config2_n20000.json")


def survey_bytes_per_candidate(n_q, kbar, cells=27):
    """SURVEY.md 8d, no cache credit: B_cand = 16 (quad read) + 8 (count write) + n_Q * (12 + c*8 + kbar*12)."""
    return 16 + 8 + n_q * (12 + cells * 8 + kbar * 12)


def structure_bytes_per_candidate(n_q, f_l0, f_l1, f_l2, groups_per_query):
    """Bytes the three-level LCP structure REQUIRES per verified candidate:
    reach word 8 B per L0 survivor; 32 B list header + 16 B exact query per L1 survivor; 48 B (one group of four points:
    three 16-byte loads) per group a sub-cell-mask survivor walks; 48 B transform + 8 B tag + 4 B index in, 4 B count out.
    The sweep's own query reads (8 B per query out of the LDS copy) never leave the CU and are reported separately."""
    sweep = 8.0 * n_q
    gathers = n_q * (8.0 * f_l0 + 48.0 * f_l1 + 48.0 * groups_per_query) + 64.0
    return sweep, gathers


def seg_len32(a, b):
    """float32 |a - b| in the reference's evaluation order x + (y + z) (match4pcsBase.hpp:318-321, Eigen 3-vector norm)."""
    d = (np.asarray(a, np.float32) - np.asarray(b, np.float32)).astype(np.float32)
    s = np.float32(d[0] * d[0]) + (np.float32(d[1] * d[1]) + np.float32(d[2] * d[2]))
    return float(np.sqrt(np.float32(s)))
