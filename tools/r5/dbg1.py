import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from super4pcs_amd import capi, datasets
P, Q, _ = datasets.bumpy_pair(20000, overlap=0.6, delta=0.01, noise_sigma=0.003, seed=3)
gm = capi.Matcher(capi.make_options(0.01, 0.6, 200), device=0)
print("created", flush=True)
gm.init_full(P, Q)
print("init done", flush=True)
ok, r = gm.try_one_base()
print("try_one_base", ok, r.n_pairs1, r.n_pairs2, r.n_quads, r.n_verified, r.best_count, flush=True)
for _ in range(3):
    ok, r = gm.try_one_base()
    print("try_one_base", ok, r.n_pairs1, r.n_pairs2, r.n_quads, r.n_verified, r.best_count, flush=True)
gm.perform_n_steps(20)
i = gm.info()
print("perform_n_steps", i.best_count, i.candidates_verified, flush=True)
