#!/usr/bin/env python
"""HBM-bound secondary kernels measured honestly: ONE cold launch each, on a working set far beyond the
256 MiB Infinity Cache, meant to run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and once more with WRITE_SIZE):
  apply    k_apply on N points (SoA, in place: 24 B per point), device-resident input
  sampler  UniformDistSampler on N host points (includes the upload; the k_vox_* kernels are what the trace isolates)
Usage: hbm_points.py apply|sampler N"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from super4pcs_amd import capi   # noqa: E402

what, n = sys.argv[1], int(sys.argv[2])
if what == "apply":
    import torch
    ctx = capi.Context(capi.make_options(0.01, 0.5, 200))
    x = torch.rand(n, device="cuda", dtype=torch.float32)
    y = torch.rand(n, device="cuda", dtype=torch.float32)
    z = torch.rand(n, device="cuda", dtype=torch.float32)
    big = torch.zeros(512 << 20, device="cuda", dtype=torch.uint8)      # sweep the Infinity Cache with something else
    big += 1
    torch.cuda.synchronize()
    M = np.eye(4, dtype=np.float32); M[0, 3] = 0.5; M[:3, :3] = [[0.8, -0.6, 0], [0.6, 0.8, 0], [0, 0, 1]]
    fp = capi.C.POINTER(capi.C.c_float)
    t0 = time.perf_counter()
    rc = ctx.L.s4p_transform_points_device(ctx.h, M.reshape(16).ctypes.data_as(fp), capi.C.c_void_p(x.data_ptr()), capi.C.c_void_p(y.data_ptr()),
                                           capi.C.c_void_p(z.data_ptr()), n)
    dt = time.perf_counter() - t0
    assert rc == 0
    print("apply n=%d bytes=%d host_call_s=%.6f checksum=%.6f" % (n, 24 * n, dt, float(x[:8].sum())))
else:
    rng = np.random.default_rng(1)
    P = (rng.random((n, 3), dtype=np.float32) * np.float32(40.0)).astype(np.float32)
    t0 = time.perf_counter()
    idx = capi.uniform_dist_sample(P, 0.05)
    dt = time.perf_counter() - t0
    print("sampler n=%d kept=%d call_s=%.4f" % (n, len(idx), dt))
