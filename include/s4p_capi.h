/*
 * s4p_capi.h -- C ABI of the MI355X-native Super4PCS hot path (libsuper4pcs_amd.so).
 *
 * Plain pointers and sizes only; no C++/torch/Eigen types.  Every entry point names
 * the reference interface (file:line under nmellado/Super4PCS v1.1.3) it replaces.
 * The reference is a single-process C++ library with no FFI of its own; the "binding"
 * a maintainer adds is the facade in include/super4pcs/ (see INTEGRATION.md), whose
 * Match4PCSBase/MatchSuper4PCS bodies call exactly these functions.
 *
 * Conventions
 *   - all point arrays are SoA float32 host pointers (x[], y[], z[]), caller-owned;
 *   - 4x4 matrices are row-major float[16];
 *   - every function returns S4P_OK (0) or a negative s4p_status; s4p_last_error()
 *     gives the message.  Nothing throws across the ABI and nothing allocated
 *     inside is handed to the caller.
 *   - one s4p_ctx = one matcher = one host thread = one GPU.
 *   - there is NO CPU fallback: s4p_create fails if no gfx950 device is present.
 */
#ifndef S4P_CAPI_H_
#define S4P_CAPI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct s4p_ctx s4p_ctx;

typedef enum {
  S4P_OK = 0,
  S4P_ERR_BAD_ARG = -1,
  S4P_ERR_NO_DEVICE = -2,
  S4P_ERR_HIP = -3,
  S4P_ERR_OOM = -4,
  S4P_ERR_CAPACITY = -5,     /* an on-device pair/quad buffer overflowed; raise s4p_limits */
  S4P_ERR_UNSUPPORTED = -6,  /* option not supported on the device path */
  S4P_ERR_STATE = -7
} s4p_status;

/* Mirrors GlobalRegistration::Match4PCSOptions, src/super4pcs/shared4pcs.h:148-190. */
typedef struct {
  float delta;
  float max_normal_difference;
  float max_translation_distance;
  float max_angle;
  float max_color_distance;
  uint64_t sample_size;
  int32_t max_time_seconds;
  uint32_t random_seed;
  float terminate_threshold;
  float overlap_estimation;
} s4p_options;

/* Device buffer capacities (entries).  0 = library default. */
typedef struct {
  uint64_t max_pairs;   /* ordered pairs per ExtractPairs call           */
  uint64_t max_quads;   /* congruent quads per base                      */
  uint64_t max_grid_cells; /* cap on the LCP grid size (cells)           */
} s4p_limits;

/* Result of one RANSAC base (one TryOneBase after base selection),
 * src/super4pcs/algorithms/match4pcsBase.hpp:328-351. */
typedef struct {
  uint64_t n_pairs1;      /* |pairs1|  (ExtractPairs #1)                                  */
  uint64_t n_pairs2;      /* |pairs2|  (ExtractPairs #2)                                  */
  uint64_t n_quads;       /* congruent quads found (FindCongruentQuadrilaterals)          */
  uint64_t n_verified;    /* candidates that passed the rms gate and were LCP-scored      */
  uint32_t best_count;    /* max inlier count over verified candidates (0 if none)        */
  int32_t  has_best;      /* 1 if n_verified > 0                                          */
  uint64_t best_rank;     /* rank of the winner in reference candidate order, or ~0       */
  int32_t  best_quad[4];  /* indices into sampled Q of the winning congruent quad         */
  float    best_transform[16];  /* row-major, centred frame (transform_)                  */
  float    best_centroid2[3];   /* qcentroid2_                                            */
  float    centroid1[3];        /* qcentroid1_                                            */
  /* Order-independent checksums of the fused device pass: sum over every congruent quad (a,b,c,d) of s4p_quad_mix(a,b,c,d)
   * mod 2^64, and the same sum over the quads that passed the rms gate (the verified candidates).  They let a base be
   * compared with the reference at sizes where the lists themselves (10^9 quads at a 20 000-point sample) cannot be. */
  uint64_t quad_checksum;
  uint64_t cand_checksum;
} s4p_base_result;

/* The checksum term: a 64-bit mix of the four sampled-Q indices of a quad (host-callable; tests and the oracle use it). */
uint64_t s4p_quad_mix(int32_t a, int32_t b, int32_t c, int32_t d);

/* ---- lifecycle ------------------------------------------------------------ */
int32_t s4p_create(const s4p_options* opt, const s4p_limits* limits /*nullable*/, int32_t device, s4p_ctx** out);
void    s4p_destroy(s4p_ctx* ctx);
const char* s4p_last_error(const s4p_ctx* ctx);   /* ctx may be NULL: last create error */
int32_t s4p_device_name(const s4p_ctx* ctx, char* buf, int32_t buflen);
/* Which k_verify instantiation the trial loops launch on the clouds that are set, with its grid, block and LDS sizes
 * (measurement provenance; no reference counterpart). */
int32_t s4p_verify_kernel_info(const s4p_ctx* ctx, char* buf, int32_t buflen);

/* Device buffer capacities follow the data, as the reference's std::vectors do (super4pcs.cc:166-174, :196,
 * match4pcsBase.hpp:340-351):
 *  - a fused pass (s4p_try_base*, and everything built on it) whose PAIR lists overflow a lane's buffers grows that lane
 *    to what the base's own counters ask for (+25 %, at least double) and runs the base again, inside the wait -- no other
 *    lane and no host state is involved;
 *  - a base whose congruent QUADS do not fit (at the "GPU-scale" sample of 20 000 points a base has ~10^9) is processed in
 *    CHUNKS: ranges of the second pair set are enumerated into the quad buffers, gated, LCP-scored and folded one after
 *    the other with the reference's first-maximum rule, so winner, best count, n_quads, n_verified and the checksums are
 *    those of an unbounded pass; afterwards the lane's quad buffers grow towards grow_cap_quads (default 32 Mi entries)
 *    so that later bases of that size take one pass.  The per-candidate records of such a base (s4p_last_candidates /
 *    s4p_last_verified, the reference's per-candidate visitor calls) are available too: see s4p_keep_candidate_records.
 * Growth is refused (S4P_ERR_CAPACITY, loudly) when the lanes together would take more than 60 % of the device memory.
 * s4p_set_auto_grow(0) / s4p_set_quad_chunking(0, ..) restore the strict contract of the stage-level entry points: an
 * overflowing base fails with S4P_ERR_CAPACITY.  s4p_chunk_stats: {chunked bases, chunk passes, range splits, quads of
 * chunked bases}; s4p_lane_growths: regrowths so far; s4p_get_limits: the largest capacities in force on any lane.
 * s4p_grow_limits raises every lane to at least min_pairs / min_quads (0 = what the last overflow reported) up front;
 * nothing may be in flight. */
int32_t s4p_grow_limits(s4p_ctx* ctx, uint64_t min_pairs, uint64_t min_quads);
int32_t s4p_get_limits(const s4p_ctx* ctx, s4p_limits* out);
int32_t s4p_set_auto_grow(s4p_ctx* ctx, int32_t enable);
int64_t s4p_lane_growths(const s4p_ctx* ctx);
int32_t s4p_set_quad_chunking(s4p_ctx* ctx, int32_t enable, uint64_t grow_cap_quads);
int32_t s4p_chunk_stats(const s4p_ctx* ctx, uint64_t* out4);

/* Early abandonment of candidates that cannot win.  The reference's Verify stops scoring a candidate as soon as it cannot
 * reach the best LCP found so far (match4pcsBase.cc:520,558-560).  s4p_set_best_hint(ctx, n) lets the fused pass do the
 * same from the next launched base on, with a bound that does not depend on candidate order: a candidate is abandoned once
 * (confirmed inliers + queries still waiting for their exact test + queries not swept yet) <= n, i.e. once it can no longer
 * EXCEED n inliers and therefore cannot become the best (match4pcsBase.hpp:468).  n must not be larger than the best inlier
 * count of the registration so far (the engine passes exactly that).  Unaffected: the winner and its count whenever it
 * exceeds n, n_quads, n_verified (the reference counts abandoned candidates as verified too), the checksums.  Lower
 * bounds only: the counts of abandoned candidates and a base's best_count when it does not exceed n.  0 = off (default). */
int32_t s4p_set_best_hint(s4p_ctx* ctx, uint32_t best_count);

/* One base over several GPUs (SURVEY.md 8e level 2).  After s4p_set_quad_slice(ctx, part, parts) every fused pass of this
 * context enumerates, gates and scores only its share of the base's SECOND pair set -- the pairs whose order key (their
 * rank in the reference's emission order, the same number on every GPU) is congruent to `part` modulo `parts`; both pair
 * sets and the set-1 structure are still built in full: they are cheap next to the candidates.  s4p_base_result then
 * describes that share: its quads, candidates, checksums, and its best candidate with the order tag (best_rank) that lets a
 * driver pick, among the shares, the greatest count and -- at equal counts -- the smallest tag, i.e. the reference's first
 * maximum (match4pcsBase.hpp:467-484).  The sharded driver does (s4p_shard_set_mode, s4p_matcher.h). */
int32_t s4p_set_quad_slice(s4p_ctx* ctx, uint32_t part, uint32_t parts);

/* options.max_angle (shared4pcs.h:160).  > 0: the segment-angle pair filter acosf(segment1 . segment2) <= max_angle
 * (pairCreationFunctor.h:203-212) runs on the device as an exact cosine threshold (the smallest float whose libm acosf
 * passes, found with libm at s4p_create).  >= 0: the Euler-angle bound of ComputeRigidTransformation
 * (match4pcsBase.cc:457-472) is decided on the device up to a margin of 1e-6 rad; the about one candidate in 10^6 inside
 * the margin is scored on the device but admitted or rejected by the HOST with the reference's own libm expression, so
 * the set of verified candidates is the reference's.  s4p_border_stats: {candidates settled by the host, rejected}. */
int32_t s4p_border_stats(const s4p_ctx* ctx, uint64_t* out2);

/* ---- state ---------------------------------------------------------------- */
/* Uploads the sampled, centred clouds and builds the device structures.
 * Replaces Match4PCSBase::initKdTree (match4pcsBase.cc:353-363; the kd-tree becomes
 * a uniform grid with the same inlier predicate, kdtree.h:417-421) and
 * PairCreationFunctor::synch3DContent (pairCreationFunctor.h:90-122).
 * Q normals / rgb may be NULL (treated as zero normals / rgb=-1 as Point3D does). */
int32_t s4p_set_clouds(s4p_ctx* ctx,
                       const float* px, const float* py, const float* pz, int64_t n_p,
                       const float* qx, const float* qy, const float* qz,
                       const float* qnx, const float* qny, const float* qnz,
                       const float* qr, const float* qg, const float* qb, int64_t n_q);

/* Wall time of the last s4p_set_clouds in seconds: {host copies + unit frame + grid plan, device build of the LCP
 * structure, Q-side uploads, total}.  Measurement aid, no reference counterpart. */
int32_t s4p_set_clouds_timing(const s4p_ctx* ctx, double* out4);

/* base_3D_ of the current RANSAC base (4 points, ordered as TryQuadrilateral left them):
 * positions, normals, rgb as float[12] each (normals/rgb nullable).
 * Replaces PairCreationFunctor::setBase (pairCreationFunctor.h:135-143). */
int32_t s4p_set_base(s4p_ctx* ctx, const float* base_xyz, const float* base_nrm, const float* base_rgb);

/* ---- hot loop A: MatchSuper4PCS::ExtractPairs (super4pcs.cc:183-224) ------- */
/* Writes ordered pairs (first,second) in the reference's emission order
 *  to out_pairs (2 ints per pair, capacity cap pairs); *n_out = m. */
int32_t s4p_extract_pairs(s4p_ctx* ctx, float pair_distance, float pair_normals_angle,
                          float pair_distance_epsilon, int32_t base_point1, int32_t base_point2,
                          int32_t* out_pairs, int64_t cap, int64_t* n_out);

/* ---- hot loop B: MatchSuper4PCS::FindCongruentQuadrilaterals (super4pcs.cc:80-177) */
/* pairs1/pairs2: 2 ints per pair; out_quads: 4 ints per quad in std::set (id,i) order. */
int32_t s4p_find_congruent(s4p_ctx* ctx, float invariant1, float invariant2,
                           float distance_threshold1, float distance_threshold2,
                           const int32_t* pairs1, int64_t m1, const int32_t* pairs2, int64_t m2,
                           int32_t* out_quads, int64_t cap, int64_t* n_out);

/* ---- hot loop C: Match4PCSBase::TryCongruentSet (match4pcsBase.hpp:363-497)
 *      = ComputeRigidTransformation (match4pcsBase.cc:365-500) + Verify (:508-567) */
/* base_ids: the 4 sampled-P indices of the base; quads: 4 ints each.
 * per_candidate (nullable, K entries): -1 if the rms gate rejected the quad, else the
 * integer inlier count of Verify (no early exit).  result: best of this set only. */
int32_t s4p_try_congruent_set(s4p_ctx* ctx, const int32_t* base_ids, const int32_t* quads, int64_t K,
                              int32_t* per_candidate, s4p_base_result* result);

/* Match4PCSBase::Verify (match4pcsBase.cc:508-567) for B explicit row-major 4x4
 * transforms: counts[b] = number of sampled-Q points with a sampled-P point within delta. */
int32_t s4p_verify_transforms(s4p_ctx* ctx, const float* transforms, int64_t B, uint32_t* counts);
/* Same, through the instrumented kernel (slower): stats4 = {exact point tests, queries that passed the coarse bitmap,
 * the reach bitmap, the sub-cell mask} summed over the batch -- the measured inputs of the roofline's byte model
 * (DESIGN.md section 7).  Measurement aid, no reference counterpart. */
int32_t s4p_verify_transforms_counted(s4p_ctx* ctx, const float* transforms, int64_t B, uint32_t* counts, uint64_t* stats4);

/* ---- fused, device-resident A -> B -> C for one base -------------------------
 * Equivalent to the body of Match4PCSBase::TryOneBase after SelectQuadrilateral
 * (match4pcsBase.hpp:313-351): two ExtractPairs, FindCongruentQuadrilaterals,
 * TryCongruentSet; nothing but the s4p_base_result leaves the GPU.
 * base_ids: sampled-P indices (ordered); base3D via s4p_set_base beforehand. */
int32_t s4p_try_base(s4p_ctx* ctx, const int32_t* base_ids, float invariant1, float invariant2,
                     s4p_base_result* result);

/* Pipelined form of s4p_try_base: _async enqueues the whole device pass of the base set by s4p_set_base
 * and returns at once (the host can select the next base and build its pair octree meanwhile); _wait
 * returns results in submission order.  Up to s4p_pipeline_depth() bases may be in flight, each with private device
 * buffers (a "lane").  Consecutive lanes form GROUPS (default: 14 lanes in groups of 2; S4P_LANES, S4P_GROUP): the bases
 * of a group go through every kernel of the pass in ONE launch, on one HIP stream per group, so that the small kernels
 * of one group overlap the LCP scoring of another and every launch carries more than one base's work.  A group is
 * launched when its last base has been submitted -- or earlier, with the bases it has, as soon as _wait is called for
 * one of them: any call pattern (one base at a time, a full pipeline) gets the same results, only the packing differs.
 * The number of streams in use should not exceed GPU_MAX_HW_QUEUES (read by the HIP runtime when it initialises; this
 * library does not touch the environment: export it, e.g. 8, before the first HIP call -- INTEGRATION.md). */
int32_t s4p_pipeline_depth(const s4p_ctx* ctx);
int32_t s4p_try_base_async(s4p_ctx* ctx, const int32_t* base_ids, float invariant1, float invariant2);
int32_t s4p_try_base_wait(s4p_ctx* ctx, s4p_base_result* result);

/* Two-step form for a driver that prepares bases on another host thread:
 *   s4p_stage_base   host-only part of the two ExtractPairs calls of a base (octree loop 1, which also advances the
 *                    persistent permutation); with want_device_data the flat sequences are left in pinned staging
 *                    slot `slot` (0 .. s4p_stage_slots()-1).  Touches no device state: it may run on a different
 *                    thread than the one that owns the context, provided bases are staged in trial order.
 *   s4p_try_base_staged_async   uploads slot `slot` and enqueues the device pass (base set by s4p_set_base).
 * A slot may be re-staged once the s4p_try_base_wait of the base that used it has returned.
 * s4p_try_base_async itself uses the lower half of the slots round-robin; a threaded driver uses the upper half
 * (s4p_stage_slots() / 2 and up). */
int32_t s4p_stage_slots(const s4p_ctx* ctx);
int32_t s4p_stage_base(s4p_ctx* ctx, const float* base_xyz, const float* base_nrm, int32_t want_device_data, int32_t slot);
int32_t s4p_try_base_staged_async(s4p_ctx* ctx, int32_t slot, const int32_t* base_ids, float invariant1, float invariant2);

/* Snapshot / restore of the persistent pair-octree permutation (PairCreationFunctor::ids,
 * pairCreationFunctor.h:36): lets a speculative, pipelined driver roll the host state back to the
 * exact point where the sequential reference stopped. */
int32_t s4p_pair_state_words(const s4p_ctx* ctx);
int32_t s4p_pair_state_save(const s4p_ctx* ctx, uint32_t* out);
int32_t s4p_pair_state_restore(s4p_ctx* ctx, const uint32_t* in);

/* Multi-GPU sharding by base: a rank that does NOT own the current base still has to
 * advance the persistent pair-octree permutation exactly as the two ExtractPairs calls of that base
 * would (intersectionNode.h:156-176 partitions functor.ids in place), so that later bases emit pairs in
 * the reference order on every rank.  Host-only; launches nothing. */
int32_t s4p_skip_base(s4p_ctx* ctx);

/* Debug/parity access to the last s4p_try_base: per-candidate records in reference order.
 * quads (4 ints), counts (-1 = gate failed); returns K via n_out. */
int32_t s4p_last_candidates(s4p_ctx* ctx, int32_t* quads, int32_t* counts, int64_t cap, int64_t* n_out);

/* For visitors that want every verified candidate (the reference calls v(-1, lcp, T) per candidate,
 * match4pcsBase.hpp:458-465): inlier counts and row-major 4x4 transforms (centred frame) of the candidates verified
 * by the base whose s4p_try_base_wait returned last, in reference order.  Valid until that lane is reused. */
int32_t s4p_last_verified(s4p_ctx* ctx, uint32_t* counts, float* transforms16, int64_t cap, int64_t* n_out);

/* Per-candidate records of a base that takes SEVERAL device passes -- a fused base whose quads exceed the lane's buffers
 * (chunked, see above), a caller's quad list longer than them (s4p_try_congruent_set scores it in slices).  The reference
 * delivers v(-1, lcp, T) for every candidate of every base (match4pcsBase.hpp:458-465) and returns whole quad lists
 * (super4pcs.cc:166-174), whatever their size.
 *  - s4p_set_candidate_sink: while a sink is set, every base hands its verified candidates to it from inside the wait
 *    (s4p_try_base_wait / s4p_try_base / s4p_try_congruent_set), in REFERENCE ORDER, in one call or -- for a multi-pass
 *    base -- one call per pass: counts[n] and row-major 4x4 transforms (centred frame).  A chunked base is then cut along the
 *    order key of its first pair set, the primary key of the reference's candidate order, so pass after pass comes out in
 *    that order and the host never holds more than one pass.  The matcher's per-candidate visitor runs on this.
 *  - s4p_keep_candidate_records(ctx, 1): the same ordered passes, the records (and the quads with their counts) kept on the
 *    host for s4p_last_candidates / s4p_last_verified (memory: 20 B per quad + 68 B per candidate).
 *  - with neither, those two calls replay a multi-pass fused base once in that mode (possible until its staging slot is
 *    rewritten: S4P_ERR_STATE afterwards). */
typedef void (*s4p_candidate_sink)(void* user, const uint32_t* counts, const float* transforms16, int64_t n);
int32_t s4p_set_candidate_sink(s4p_ctx* ctx, s4p_candidate_sink sink, void* user);
int32_t s4p_keep_candidate_records(s4p_ctx* ctx, int32_t enable);

/* ---- base selection: Match4PCSBase::SelectRandomTriangle + the 4th-point scan of SelectQuadrilateral
 * (match4pcsBase.cc:185-218, 279-338) as device reductions over the sampled P resident in HBM: ONE attempt.
 * draws: the 2001 indices (first, then 1000 x (second, third)) of `rand() % n_P` the reference would draw for this
 * attempt -- the random stream stays with the caller; limit_sq = max_base_diameter^2, too_small = (0.2 * max_base_diameter)^2.
 * ids[4] / xyz[12]: the three triangle points and the 4th point (sampling-order indices, -1 where none; centred
 * coordinates).  status: 0 found, 1 no wide triangle (SelectRandomTriangle returned false), 2 coplanar-with-origin
 * triangle (denominator 0: the reference retries), 3 no admissible 4th point (retries).  The winner of either search is
 * the reference's: first strictly wider triangle / first strictly closer point in index order.
 * May be called from a thread of its own while bases are in flight. */
int32_t s4p_select_base_points(s4p_ctx* ctx, const uint32_t* draws, float limit_sq, float too_small,
                               int32_t* ids, float* xyz, int32_t* status);
/* The same for n_attempts (<= s4p_select_batch_max()) CONSECUTIVE attempts of the random stream in one set of launches and
 * one synchronisation: draws = n_attempts x 2001 indices, ids / xyz / status = one record per attempt.  The stream does
 * not depend on results, so a driver can draw ahead; which attempt ends which trial is decided from the statuses, in
 * order, exactly as the reference's loop would (a failed attempt is followed by the next one; match4pcsBase.cc:283-349).
 * At n_P = 4.2 M one attempt costs ~130 us of launches and synchronisation; eight attempts per call cost little more. */
int32_t s4p_select_base_points_batch(s4p_ctx* ctx, const uint32_t* draws, int32_t n_attempts, float limit_sq, float too_small,
                                     int32_t* ids, float* xyz, int32_t* status);
int32_t s4p_select_batch_max(void);

/* ---- final apply: Match4PCSBase::Perform_N_steps tail (match4pcsBase.hpp:265-267) */
/* xyz SoA in place: p <- (M * [p;1]).head<3>() for n points. */
int32_t s4p_transform_points(s4p_ctx* ctx, const float* M, float* x, float* y, float* z, int64_t n);
/* Same for a cloud that already lives in HBM: x, y, z are DEVICE pointers, transformed in place. */
int32_t s4p_transform_points_device(s4p_ctx* ctx, const float* M, float* dev_x, float* dev_y, float* dev_z, int64_t n);
/* Measurement aid: times k_apply (out_ms[0], the product's VALU kernel) and its v_mfma_f32_4x4x1 variant (out_ms[1]) on n
 * device-resident synthetic points, `reps` launches each, and counts the coordinates on which the two differ
 * (why the final apply is not on the matrix cores). */
int32_t s4p_apply_bench(s4p_ctx* ctx, int64_t n, int32_t reps, double* out_ms2, uint64_t* mismatch, float* max_abs);

/* ---- instrumentation ------------------------------------------------------- */
typedef struct {
  uint64_t verify_launches;      /* number of LCP-verify kernel launches timed          */
  double   verify_ms_total;      /* HIP-event time of those launches (ms)               */
  uint64_t verify_candidates;    /* candidates LCP-scored in those launches             */
  uint64_t verify_quads;         /* quads read by those launches (gate evaluated)       */
  uint64_t verify_point_tests;   /* P-point distance tests (k-bar numerator), if enabled */
  uint64_t verify_queries;       /* point queries (candidates * n_Q)                    */
  uint64_t verify_l0_pass;       /* queries that passed the LDS coarse bitmap, if enabled */
  uint64_t verify_l1_pass;       /* queries that passed the reach bitmap, if enabled */
  uint64_t verify_l2_pass;       /* queries that passed the 4x4x4 sub-cell mask (go on to exact tests), if enabled */
  double   pairs_ms_total, quads_ms_total;
  uint64_t pairs_launches, quads_launches;
  double   host_octree_s;        /* host time in the pair-octree builds (loop 1 of IntersectionFunctor)   */
  double   host_wait_s;          /* host time blocked in stream synchronisation                           */
  uint64_t verify_pruned;        /* candidates abandoned because they could not exceed the best-count hint */
  uint64_t sweep_candidates;     /* candidates that went through the counting first pass (k_sweep: samples beyond LDS) ... */
  uint64_t sweep_survivors;      /* ... and those it handed on to the scoring pass                                         */
} s4p_profile;
/* enable_events: 0 off; 1 HIP events around every stage of a base (five records per base: verify_*, pairs_*, quads_*); 2 around
 * the LCP-verify kernel only (two records per base: what a throughput measurement that also wants the kernel's launch time
 * should use -- timing events are barriers in the lane's stream).  count_point_tests: the instrumented verify kernel. */
int32_t s4p_profile_enable(s4p_ctx* ctx, int32_t enable_events, int32_t count_point_tests);
int32_t s4p_profile_get(s4p_ctx* ctx, s4p_profile* out, int32_t reset);

/* IEEE self-test of the device float path (sqrt, divide, no-FMA mul/add) against the
 * values the host computed for the same inputs; returns number of mismatches in *n_bad. */
int32_t s4p_selftest_ieee(s4p_ctx* ctx, const float* a, const float* b, int64_t n,
                          float* out_sqrt, float* out_div, float* out_muladd);

#ifdef __cplusplus
}
#endif
#endif /* S4P_CAPI_H_ */
