// s4p_kernels.hip.hpp -- gfx950 device code of the Super4PCS hot path (HIP, wave64).
//
// Compiled ONLY for gfx950 with -ffp-contract=off: every float expression below is
// evaluated as separate IEEE mul/add (no FMA), with correctly rounded sqrtf and '/',
// in the exact association order the reference's Eigen 3.3 fixed-size expressions
// use (see DESIGN.md "Numerics contract").  That is what makes integer inlier counts
// bit-exact against the CPU path.
//
// Reference functions restated here (paths under /root/reference/src/super4pcs/):
//   k_pairs2       accelerators/pairExtraction/intersectionFunctor.h:197-233 (loop 2),
//                  intersectionPrimitive.h:117-157, algorithms/pairCreationFunctor.h:151-218
//   k_prep (and the append step of k_pairs2)  algorithms/super4pcs.cc:118-146, accelerators/normalset.hpp:110-127,162-203
//   k_quads        algorithms/super4pcs.cc:151-163
//   k_gate/k_quads algorithms/match4pcsBase.cc:365-500 (ComputeRigidTransformation + rms gate, match4pcsBase.hpp:436-439)
//   k_verify       match4pcsBase.cc:508-567 (Verify), accelerators/kdtree.h:417-421 (predicate),
//                  match4pcsBase.hpp:467-484 (first strictly greater LCP wins)
//   k_apply        algorithms/match4pcsBase.hpp:265-267
// The code lives in parts (round 6: the 2.6 k-line file was a trap for its own comments), included below in dependency order:
//   s4p_k_common.hip.hpp       exact-order float helpers, the per-base counters / result record, the LCP structure (LcpGrid) and its locate helpers; the -DS4P_PROF stamps
//   s4p_k_gridbuild.hip.hpp    device build of the LCP structure (s4p_set_clouds)
//   s4p_k_lcp.hip.hpp          LCP scoring of one candidate by one wave: the fused / staged sweeps (full counts) and the lean sweep (an early-exit bound in force), LDS staging
//   s4p_k_rigid.hip.hpp        ComputeRigidTransformation + the rms / Euler-angle gate
//   s4p_k_prep.hip.hpp         FindCongruentQuadrilaterals, preparation side: cell hash, set-1 records, cone masks; k_prep
//   s4p_k_pairs.hip.hpp        ExtractPairs loop 2 + PairCreationFunctor::process: k_pairs2
//   s4p_k_quads.hip.hpp        the gate of one quad, k_gate, k_quads (enumeration + gate)
//   s4p_k_verify.hip.hpp       k_sweep (counting first pass), k_verify (scoring + winner + result record), k_verify_T
//   s4p_k_misc.hip.hpp         k_apply, device base selection (k_select_*), k_pack_points, k_selftest, k_reset_counters
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "s4p_k_common.hip.hpp"
#include "s4p_k_gridbuild.hip.hpp"
#include "s4p_k_lcp.hip.hpp"
#include "s4p_k_rigid.hip.hpp"
#include "s4p_k_prep.hip.hpp"
#include "s4p_k_pairs.hip.hpp"
#include "s4p_k_quads.hip.hpp"
#include "s4p_k_verify.hip.hpp"
#include "s4p_k_misc.hip.hpp"
