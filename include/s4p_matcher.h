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
/* Declares this matcher rank `rank` of `world` (trial t is owned by rank t mod world, counted from this call) and,
 * with producer_threads != 0, moves base selection and octree staging onto two helper threads that run ahead of the
 * caller: both are sequential by nature (RNG stream; persistent octree permutation) but never read results.  The
 * caller keeps calling next_base / next_base_async / perform_n_steps as before; state is rewound exactly when a
 * trial loop stops. */
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

/* getGlobalTransform (match4pcsBase.hpp:224-229). */
int32_t s4p_matcher_global_transform(s4p_matcher* m, float* transformation);

/* Whole ComputeTransformation with the default sampler and no visitor:
 * Q (x,y,z arrays) is transformed in place when the LCP improved.  Returns the LCP in *lcp
 * (1e9 on empty input, match4pcsBase.hpp:69-70). */
int32_t s4p_matcher_compute_transformation(s4p_matcher* m, const s4p_cloud_view* P, const s4p_cloud_view* Q,
                                           float* qx_out, float* qy_out, float* qz_out,
                                           float* transformation, float* lcp);

#ifdef __cplusplus
}
#endif
#endif /* S4P_MATCHER_H_ */
