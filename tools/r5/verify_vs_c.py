#!/usr/bin/env python
"""k_verify's launch time against the number of candidates of the base, one base in flight (lab aid): does a launch cost
`fixed + tail + C x marginal` with a small marginal term?  perform_n_steps(1) per base with the events on."""
import json
import os
import sys

os.environ.setdefault("S4P_LANES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from super4pcs_amd import capi, datasets   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
P, Q, _ = datasets.bumpy_pair(1_000_000, overlap=0.5, delta=0.004, seed=20140814)
m = capi.Matcher(capi.make_options(0.004, 0.5, 2000), device=0, max_pairs=8 << 20, max_quads=64 << 20)
m.init_full(P, Q)
m.set_sharding(0, 1, 0)
m.perform_n_steps(5)
m.profile_enable(True, False)
m.profile_get(reset=True)
rows = []
for _ in range(n):
    m.perform_n_steps(1)
    p = m.profile_get(reset=True)
    if p.verify_launches:
        rows.append((int(p.verify_candidates), round(p.verify_ms_total * 1e3, 1), round(p.pairs_ms_total * 1e3, 1), round(p.quads_ms_total * 1e3, 1)))
rows.sort()
print(json.dumps({"rows_C_verify_us_pairs_us_quads_us": rows}))
m.close()
