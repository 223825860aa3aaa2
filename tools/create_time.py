import time, sys
sys.path.insert(0, '/root/repo')
from super4pcs_amd import capi
import numpy as np
opt = capi.make_options(0.01, 0.5, 200)
for mp, mq in ((0, 0), (1 << 20, 4 << 20), (8 << 20, 64 << 20)):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        m = capi.Matcher(opt, max_pairs=mp, max_quads=mq)
        t1 = time.perf_counter()
        m.close()
        ts.append(t1 - t0)
    print("limits", mp, mq, "create s", [round(t, 4) for t in ts], flush=True)
