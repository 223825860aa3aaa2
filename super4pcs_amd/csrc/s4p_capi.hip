// s4p_capi.hip -- context + C ABI (include/s4p_capi.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see super4pcs_amd/build.py)
#include "s4p_capi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "s4p_host_structs.hpp"
#include "s4p_kernels.hip.hpp"

using namespace s4p;

// The file is a list of parts (round 6): each is a section of the former 2.2 k-line source, in dependency order.
//   s4p_capi_ctx.inc           small helpers, the context (s4p_ctx: host mirrors, device structures, lanes and groups, staging ring, knobs), the error macros
//   s4p_capi_launch.inc        parameter records and launches of the kernels of a device pass; the wait for a result record; lane growth
//   s4p_capi_pass.inc          what happens around a device pass: profiling events, borderline candidates, per-candidate records, chunked bases, relaunch after a growth, the result of a base, prepare / flush of base groups, lane buffers
//   s4p_capi_abi_context.inc   C ABI: context life cycle, limits and knobs, s4p_set_clouds (LCP structure, Q-side uploads)
//   s4p_capi_abi_stages.inc    C ABI: stage-level entry points (ExtractPairs / FindCongruentQuadrilaterals / TryCongruentSet / Verify), the fused asynchronous base pass, per-candidate records
//   s4p_capi_abi_misc.inc      C ABI: final apply (s4p_transform_points), device base selection, profile counters, IEEE self-test
#include "s4p_capi_ctx.inc"
#include "s4p_capi_launch.inc"
#include "s4p_capi_pass.inc"
#include "s4p_capi_abi_context.inc"
#include "s4p_capi_abi_stages.inc"
#include "s4p_capi_abi_misc.inc"
