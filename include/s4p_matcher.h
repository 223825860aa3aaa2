/*
 * s4p_matcher.h -- C ABI of the host RANSAC driver (C++ engine inside
 * libsuper4pcs_amd.so) that stands behind the reference's public surface
 *     GlobalRegistration::Match4PCSBase::ComputeTransformation()
 *         (src/super4pcs/algorithms/match4pcsBase.h:108-115, match4pcsBase.hpp:61-86)
 * and its protected steps init / Perform_N_steps / TryOneBase / SelectQuadrilateral
 * (match4pcsBase.hpp:90-360, match4pcsBase.cc:185-351).  The header-only facade in
 * include/super4pcs/ forwards its templates (Sampler, Visitor) to these functions;
 * the device work goes through include/s4p_capi.h.
 *
 * Point clouds cross this ABI as SoA float32 arrays; normals / rgb are optional (NULL).
 */
#ifndef S4P_MATCHER_H_
#define S4P_MATCHER_H_

#include <stdint.h>

#include "s4p_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct s4p_matcher s4p_matcher;

typedef struct {
  const float* x; const float* y; const float* z;
  const float* nx; const float* ny; const float* nz;   /* nullable (all three) */
  const float* r; const float* g; const float* b;      /* nullable (all three) */
  int64_t n;
} s4p_cloud_view;

typedef struct {
  int32_t number_of_trials;   /* number_of_trials_            */
  int32_t current_trial;      /* current_trial_               */
  int32_t n_sampled_p;        /* |sampled_P_3D_|              */
  int32_t n_sampled_q;        /* |sampled_Q_3D_|              */
  float   best_lcp;           /* best_LCP_                    */
  uint32_t best_count;        /* integer inlier count behind best_LCP_ */
  float   p_diameter;         /* P_diameter_ (measured on sampled Q: reference quirk) */
  float   centroid_p[3], centroid_q[3];
  float   transform[16];      /* transform_ (row-major, centred frame) */
  float   qcentroid1[3], qcentroid2[3];
  int32_t base[4], congruent[4];
  uint64_t candidates_verified;   /* visitor calls with fraction == -1 in the reference */
  uint64_t quads_total, pairs_total;
  uint64_t bases_tried;
  double  seconds_select, seconds_device;
} s4p_matcher_info;

/* Visitor concept of match4pcsBase.h:73-76: v(fraction, best_lcp, transformation[16 row-major]). */
typedef void (*s4p_visitor_fn)(void* user, float fraction, float best_lcp, float* transformation);

int32_t s4p_matcher_create(const s4p_options* opt, const s4p_limits* limits, int32_t device, s4p_matcher** out);
void    s4p_matcher_destroy(s4p_matcher* m);
const char* s4p_matcher_last_error(const s4p_matcher* m);
s4p_ctx* s4p_matcher_ctx(s4p_matcher* m);   /* the device context, for profiling */

/* UniformDistSampler::operator() (src/super4pcs/sampling.h:104-121): keeps the first point of
 * every delta-voxel in input order.  out_index receives the kept input indices (capacity n).
 * Returns the number kept, 0 on bad arguments, -1 if the device sampler hit a HIP error (printed to stderr;
 * there is no silent host fallback for errors -- the host hash only serves small clouds and machines without a device). */
int64_t s4p_uniform_dist_sample(const float* x, const float* y, const float* z, int64_t n, float delta,
                                int64_t* out_index);

/* Match4PCSBase::init (match4pcsBase.hpp:90-203) after the Sampler ran.
 *   p : sampled P (or the whole P when |P| <= sample_size)
 *   q : uniformly sampled Q before shuffle/truncation (q_needs_shuffle = 1), or the whole Q
 *       when |Q| <= sample_size (q_needs_shuffle = 0). */
int32_t s4p_matcher_init(s4p_matcher* m, const s4p_cloud_view* p, const s4p_cloud_view* q, int32_t q_needs_shuffle);

/* Convenience: default sampler on full clouds + s4p_matcher_init. */
int32_t s4p_matcher_init_full(s4p_matcher* m, const s4p_cloud_view* P, const s4p_cloud_view* Q);

int32_t s4p_matcher_get_info(s4p_matcher* m, s4p_matcher_info* out);
/* getFirstSampled (which=0) / getSecondSampled (which=1), match4pcsBase.h:88-95; centred coordinates. */
int32_t s4p_matcher_get_sampled(s4p_matcher* m, int32_t which, float* x, float* y, float* z);
/* Normals / colours of the same points, in the same (engine) order -- for Q that is the order after the shuffle and the
 * truncation to sample_size of match4pcsBase.hpp:129-138.  Any of the six output pointers may be NULL.  *has_normals /
 * *has_rgb (nullable) tell whether the cloud carried them at init (otherwise zeros / -1 are written, as Point3D defaults). */
int32_t s4p_matcher_get_sampled_attrs(s4p_matcher* m, int32_t which, float* nx, float* ny, float* nz,
                                      float* r, float* g, float* b, int32_t* has_normals, int32_t* has_rgb);

/* Match4PCSBase::SelectQuadrilateral (match4pcsBase.cc:279-351).  base_xyz: 12 floats (ordered base). */
int32_t s4p_matcher_select_quadrilateral(s4p_matcher* m, int32_t* found, float* invariant1, float* invariant2,
                                         int32_t* base_ids, float* base_xyz);

/* Match4PCSBase::TryOneBase (match4pcsBase.hpp:281-360): *ok = its boolean result. */
int32_t s4p_matcher_try_one_base(s4p_matcher* m, int32_t* ok, s4p_base_result* last /*nullable*/);

/* The two halves of TryOneBase, for sharding bases over several GPUs (one process per GPU):
 *   next_base : SelectQuadrilateral + (run_device ? the fused device pass : only the host-side state
 *               advance of s4p_skip_base).  Every rank calls it for every trial so RNG and pair-octree
 *               state stay identical; only the owner passes run_device = 1.
 *   commit    : the "if (lcp > best_LCP_)" update of TryCongruentSet (match4pcsBase.hpp:467-484) applied
 *               to a result that may have been produced on another rank; *ok = TryOneBase's return value. */
int32_t s4p_matcher_next_base(s4p_matcher* m, int32_t run_device, int32_t* found, int32_t* base_ids, s4p_base_result* result);
/* Declares this matcher rank `rank` of `world` (trial t is owned by rank t mod world, counted from this call) and says
 * whether base selection and octree staging move onto helper threads that run ahead of the caller (both are sequential by
 * nature -- RNG stream; persistent octree permutation -- but never read results): producer_threads = 0 never, 1 always,
 * 2 where they pay = with 4 or more ranks, or when SelectQuadrilateral's searches run on the device (sampled P >= 2^20
 * points); measured in DESIGN.md 5.1.  A matcher this was never called on behaves as (0, 1, 2).  The caller keeps calling
 * next_base / next_base_async / perform_n_steps as before; state is rewound exactly when a trial loop stops. */
int32_t s4p_matcher_set_sharding(s4p_matcher* m, int32_t rank, int32_t world, int32_t producer_threads);

/* Pipelined next_base: the owner's device pass is only enqueued (at most two in flight); wait_base
 * returns the results in submission order.  Lets a rank overlap its GPU pass with the host-side work of
 * the following trials. */
int32_t s4p_matcher_next_base_async(s4p_matcher* m, int32_t run_device, int32_t* found, int32_t* base_ids);
int32_t s4p_matcher_wait_base(s4p_matcher* m, s4p_base_result* result);
int32_t s4p_matcher_commit(s4p_matcher* m, int32_t found, const int32_t* base_ids, const s4p_base_result* result, int32_t* ok);

/* Match4PCSBase::Perform_N_steps (match4pcsBase.hpp:208-274) without the final apply to Q:
 * runs up to n trials, calling visitor (nullable) as the reference does; on return
 * *improved = (best_LCP_ > LCP at entry) and transformation = the global transform if
 * improved (match4pcsBase.hpp:259-262), else transform_.  *done = its boolean result. */
int32_t s4p_matcher_perform_n_steps(s4p_matcher* m, int32_t n, s4p_visitor_fn visitor, void* user,
                                    int32_t visitor_needs_global, float* transformation,
                                    int32_t* improved, int32_t* done);

/* Per-candidate visitor calls: with enable != 0, s4p_matcher_perform_n_steps also calls visitor(user, -1, lcp, T) for
 * every verified candidate of every trial, in the reference's order (match4pcsBase.hpp:458-465).  lcp is the full
 * inlier fraction (the reference's Verify may have stopped early against its running best, so its value can be
 * lower for candidates that cannot win).  Costs a device read-back per trial; off by default. */
int32_t s4p_matcher_visit_candidates(s4p_matcher* m, int32_t enable);

/* Early exit: inside the trial loops (s4p_matcher_perform_n_steps / _compute_transformation, the sharded loop) the device
 * abandons candidates that can no longer EXCEED the best inlier count committed so far (s4p_set_best_hint) -- the
 * reference's Verify does the same sequentially (match4pcsBase.cc:520,558-560).  Results are unaffected: best LCP, winner,
 * transform, candidates_verified.  On by default; off automatically while a per-candidate visitor listens
 * (s4p_matcher_visit_candidates) and never active for s4p_matcher_try_one_base / next_base outside a loop;
 * s4p_matcher_set_early_exit(m, 0) or S4P_EARLY_EXIT=0 in the environment restore full counts everywhere.
 * s4p_matcher_loop_begin / _end bracket a driver's own trial loop (the sharded loop uses them). */
int32_t s4p_matcher_set_early_exit(s4p_matcher* m, int32_t enable);
int32_t s4p_matcher_loop_begin(s4p_matcher* m);
int32_t s4p_matcher_loop_end(s4p_matcher* m);

/* Where SelectQuadrilateral's two searches run (match4pcsBase.cc:185-218 wide triangle, :321-338 4th point): mode 1 =
 * device reductions over the sampled P resident in HBM (s4p_select_base_points), 0 = the host search structures, -1
 * (default) = by size: device from 2^20 sampled P points (the S4P_DEVICE_SELECT environment variable overrides the
 * default).  Same draws, same bases either way.  Takes effect at the next init; s4p_matcher_device_selection reports
 * what the initialised matcher uses. */
int32_t s4p_matcher_set_device_selection(s4p_matcher* m, int32_t mode);
int32_t s4p_matcher_device_selection(const s4p_matcher* m);

/* Device buffer capacities (s4p_limits): when a base has more pairs than a lane's buffers hold, that lane grows them to
 * what the base's own counters ask for and runs the base again inside the wait; a base with more congruent quads than fit
 * is processed in chunks (s4p_capi.h, "Device buffer capacities follow the data") -- same trials and results as with
 * limits that were large enough from the start (the reference's std::vector simply grows).  On by default, in the
 * sequential and in the sharded loops; with enable == 0, or when growth is refused (more than 60 % of the device memory),
 * the call fails with S4P_ERR_CAPACITY as the stage-level entry points always do.  s4p_matcher_capacity_growths counts
 * the regrowths. */
int32_t s4p_matcher_grow_on_overflow(s4p_matcher* m, int32_t enable);
int32_t s4p_matcher_capacity_growths(const s4p_matcher* m);

/* current_trial_ += n (match4pcsBase.hpp:258), for a driver that runs the trial loop itself through SelectQuadrilateral /
 * ExtractPairs / FindCongruentQuadrilaterals / TryCongruentSet (the facade does when a subclass overrides the hooks). */
int32_t s4p_matcher_advance_trials(s4p_matcher* m, int32_t n);

/* getGlobalTransform (match4pcsBase.hpp:224-229). */
int32_t s4p_matcher_global_transform(s4p_matcher* m, float* transformation);

/* Whole ComputeTransformation with the default sampler and no visitor:
 * Q (x,y,z arrays) is transformed in place when the LCP improved.  Returns the LCP in *lcp
 * (1e9 on empty input, match4pcsBase.hpp:69-70). */
int32_t s4p_matcher_compute_transformation(s4p_matcher* m, const s4p_cloud_view* P, const s4p_cloud_view* Q,
                                           float* qx_out, float* qy_out, float* qz_out,
                                           float* transformation, float* lcp);

/* opt.terminate_threshold of the matcher (the sharded loop needs it to rank "crossed the threshold" outcomes). */
float s4p_matcher_terminate_threshold(const s4p_matcher* m);
int32_t s4p_matcher_max_time_seconds(const s4p_matcher* m);
/* Number of s4p_matcher_init / _init_full calls so far: drivers that keep per-registration state (the sharded loop's
 * "terminated" flag, trial counters) reset it when this changes. */
int64_t s4p_matcher_init_generation(const s4p_matcher* m);

/* ---- multi-GPU: bases sharded over the GPUs of one node, one process per GPU ------------------
 * Every rank walks the same base sequence; rank (t mod world) runs the device pass of trial t; after each window of
 * `world` trials ONE 8-byte all-reduce(MAX) of a packed key picks the winner exactly as the sequential loop of
 * Match4PCSBase::Perform_N_steps would (match4pcsBase.hpp:236-256, 467-484), and the winner's record is broadcast only
 * when the window improved the best LCP.  Implemented in C++ (super4pcs_amd/csrc/s4p_shard.cpp); the built-in
 * collective calls RCCL (rccl.h) over xGMI, a caller-supplied one (MPI, gloo, ...) plugs into the same loop. */
typedef struct s4p_shard s4p_shard;
typedef struct {
  void* user;
  int32_t (*allreduce_max_u64)(void* user, uint64_t* key);                     /* in place, MAX over all ranks; 0 = ok */
  int32_t (*broadcast)(void* user, void* buf, int64_t bytes, int32_t root);    /* root's bytes to every rank; 0 = ok   */
} s4p_collective;

/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks by any means (file, MPI, env, ...). */
int32_t s4p_rccl_unique_id(uint8_t* out128);
/* Declares matcher m rank `rank` of `world` (as s4p_matcher_set_sharding does) and creates the sharded driver. */
int32_t s4p_shard_create(s4p_matcher* m, int32_t rank, int32_t world, int32_t producer_threads, s4p_shard** out);
void    s4p_shard_destroy(s4p_shard* s);
const char* s4p_shard_last_error(const s4p_shard* s);
/* ncclCommInitRank on `device` (the matcher's GPU); collective = ncclAllReduce(uint64, max) / ncclBroadcast on a private stream. */
int32_t s4p_shard_use_rccl(s4p_shard* s, int32_t device, const uint8_t* unique_id128);
int32_t s4p_shard_use_collective(s4p_shard* s, const s4p_collective* coll);
/* What the shard's RCCL communicator itself reports (ncclCommCount / ncclCommUserRank); -1 / -1 when the collective is not
 * the library's own RCCL communicator.  s4p_shard_use_rccl fails if they disagree with the rank / world the shard was made with. */
int32_t s4p_shard_comm_info(const s4p_shard* s, int32_t* n_ranks, int32_t* rank);
/* Measurement aid: the shard plays one rank of its world alone (reduction = own key, broadcast = no-op), so that the
 * per-window cost of a rank at a given world size can be measured on one GPU (tools/sim_world.py). */
int32_t s4p_shard_use_null_collective(s4p_shard* s);
/* How the job is spread over the ranks.  0 (default): trials sharded by base -- rank (t mod world) runs trial t, one
 * all-reduce per window of `world` trials.  1: EVERY base over all ranks (SURVEY.md 8e level 2) -- each rank runs every
 * trial on its share of the base's second pair set (s4p_set_quad_slice), two 8-byte all-reduce(MAX) per trial pick the
 * base's first maximum among the shares, one broadcast carries the winner when it improves the best LCP.  For bases that
 * take seconds each (the 20 000-point sample: ~10^9 quads per base) or jobs with fewer trials than GPUs; in this mode the
 * "windows" of s4p_shard_run_windows are single trials.  Call after s4p_shard_create, before the matcher is initialised. */
int32_t s4p_shard_set_mode(s4p_shard* s, int32_t mode);
/* n_windows windows (n_windows * world trials of the common sequence) through the pipelined loop; *candidates_local = candidates
 * this rank verified, *terminated = the terminate threshold was crossed (later windows are drained, not committed). */
int32_t s4p_shard_run_windows(s4p_shard* s, int32_t n_windows, uint64_t* candidates_local, int32_t* terminated);
/* Match4PCSBase::ComputeTransformation over `world` GPUs: same arguments and result as s4p_matcher_compute_transformation,
 * called by every rank with the same clouds. */
int32_t s4p_shard_compute_transformation(s4p_shard* s, const s4p_cloud_view* P, const s4p_cloud_view* Q,
                                         float* qx_out, float* qy_out, float* qz_out, float* transformation, float* lcp);
/* Host-only self-check of the window loop on recorded outcomes (no matcher, no GPU): the CPU tests drive it over gloo.
 * A recorded result with n_quads == UINT64_MAX stands for a device pass that FAILED on this rank: the rank returns the
 * error after posting the error key, and every other rank must leave the loop with S4P_ERR_STATE in the same window. */
int32_t s4p_shard_replay(int32_t rank, int32_t world, const s4p_collective* coll, int32_t n_windows, int32_t depth,
                         uint32_t threshold_count, uint32_t start_best_count, const int32_t* found,
                         const s4p_base_result* results, int32_t* commit_trials, uint32_t* commit_counts, int32_t commit_cap,
                         int32_t* n_commits, int32_t* terminated, uint64_t* trials_done);

/* The same self-check for the split-base mode: results[t] = THIS rank's share of trial t (best_rank = its order tag). */
int32_t s4p_shard_replay_split(int32_t rank, int32_t world, const s4p_collective* coll, int32_t n_trials, int32_t depth,
                               uint32_t threshold_count, uint32_t start_best_count, const int32_t* found,
                               const s4p_base_result* results, int32_t* commit_trials, uint32_t* commit_counts, uint64_t* commit_tags,
                               int32_t commit_cap, int32_t* n_commits, int32_t* terminated, uint64_t* trials_done);

#ifdef __cplusplus
}
#endif
#endif /* S4P_MATCHER_H_ */
