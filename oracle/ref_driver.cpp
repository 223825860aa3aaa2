// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE.
// C entry points around the REFERENCE's own classes (GlobalRegistration::MatchSuper4PCS), whose unmodified
// sources are compiled where they lie under /root/reference against oracle/eigen_shim (the image has no Eigen).
// Built by `make -C oracle ref` into oracle/_ref/libs4p_ref.so; used only by tests to pin the restatement in
// oracle/s4p_oracle.cpp (control flow, float/double mixes, RNG use, ordering) against the real code.
// The subclass re-exposes protected steps exactly like the reference's Testing::TestMatcher (tests/testing.h:71-154).
#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>

#include "super4pcs/algorithms/super4pcs.h"
#include "super4pcs/shared4pcs.h"
#include "super4pcs/utils/logger.h"

using namespace GlobalRegistration;

namespace {
struct CandidateVisitor {
  std::vector<float>* lcps;
  inline void operator()(float fraction, float lcp, Eigen::Ref<Match4PCSBase::MatrixType>) const {
    if (fraction < 0 && lcps) lcps->push_back(lcp);          // "one candidate verified", match4pcsBase.hpp:458-465
  }
  constexpr bool needsGlobalTransformation() const { return false; }
};

// bench.py cpu_baseline: counts verified candidates and aborts the run (by exception, from the per-candidate
// visitor call) once a wall-time budget is spent.  The clock starts at the visitor's first call, i.e. after init.
struct BudgetExceeded {};
struct BudgetVisitor {
  mutable uint64_t* n; mutable double* elapsed; double budget;
  mutable std::chrono::steady_clock::time_point t0; mutable bool started;
  inline void operator()(float fraction, float, Eigen::Ref<Match4PCSBase::MatrixType>) const {
    if (!started) { t0 = std::chrono::steady_clock::now(); started = true; }
    if (fraction < 0) {
      ++*n;
      if ((*n & 15u) == 0) {
        *elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (*elapsed > budget) throw BudgetExceeded();
      }
    }
  }
  constexpr bool needsGlobalTransformation() const { return false; }
};

class RefMatcher : public MatchSuper4PCS {
 public:
  RefMatcher(const Match4PCSOptions& o, const Utils::Logger& l) : MatchSuper4PCS(o, l) {}
  using MatchSuper4PCS::ExtractPairs;
  using MatchSuper4PCS::FindCongruentQuadrilaterals;
  using Match4PCSBase::SelectQuadrilateral;
  using Match4PCSBase::Verify;
  using Match4PCSBase::base3D;
  using Match4PCSBase::best_LCP_;
  using Match4PCSBase::number_of_trials_;
  using Match4PCSBase::P_diameter_;
  using Match4PCSBase::sampled_P_3D_;
  using Match4PCSBase::sampled_Q_3D_;
  using Match4PCSBase::transform_;
  using Match4PCSBase::base_;
  using Match4PCSBase::current_congruent_;
  using Match4PCSBase::base_3D_;
  void do_init(const std::vector<Point3D>& P, const std::vector<Point3D>& Q) { init(P, Q, Sampling::UniformDistSampler()); }
  bool try_set(int b1, int b2, int b3, int b4, const std::vector<Quadrilateral>& q, const CandidateVisitor& v, size_t& nb) {
    return TryCongruentSet(b1, b2, b3, b4, q, v, nb);
  }
  bool one_base(const CandidateVisitor& v) { return TryOneBase(v); }
};

struct Handle {
  Utils::Logger logger{Utils::NoLog};
  RefMatcher* m = nullptr;
  std::vector<float> lcps;
  ~Handle() { delete m; }
};

std::vector<Point3D> cloud(const float* xyz, const float* nrm, const float* rgb, uint64_t n) {
  std::vector<Point3D> c;
  c.reserve(n);
  for (uint64_t i = 0; i < n; ++i) {
    Point3D p(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    if (nrm) p.set_normal(Point3D::VectorType(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
    if (rgb) p.set_rgb(Point3D::VectorType(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
    c.push_back(p);
  }
  return c;
}
}  // namespace

extern "C" {

struct s4pr_options {
  float delta, max_normal_difference, max_translation_distance, max_angle, max_color_distance;
  uint64_t sample_size;
  int32_t max_time_seconds;
  uint32_t random_seed;
  float terminate_threshold, overlap_estimation;
};

void* s4pr_create(const s4pr_options* o) {
  Match4PCSOptions opt;
  opt.delta = o->delta; opt.max_normal_difference = o->max_normal_difference;
  opt.max_translation_distance = o->max_translation_distance; opt.max_angle = o->max_angle;
  opt.max_color_distance = o->max_color_distance; opt.sample_size = o->sample_size;
  opt.max_time_seconds = o->max_time_seconds; opt.randomSeed = o->random_seed;
  if (!opt.configureOverlap(o->overlap_estimation, o->terminate_threshold)) return nullptr;
  Handle* h = new Handle();
  h->m = new RefMatcher(opt, h->logger);
  return h;
}
void s4pr_destroy(void* h) { delete static_cast<Handle*>(h); }

void s4pr_init(void* hh, const float* P, uint64_t nP, const float* Q, uint64_t nQ) {
  Handle* h = static_cast<Handle*>(hh);
  h->m->do_init(cloud(P, nullptr, nullptr, nP), cloud(Q, nullptr, nullptr, nQ));
}
// with per-point normals / colours (nullable) -- exercises the pair filters of pairCreationFunctor.h:166-200
void s4pr_init_attr(void* hh, const float* P, const float* Pn, const float* Pc, uint64_t nP,
                    const float* Q, const float* Qn, const float* Qc, uint64_t nQ) {
  Handle* h = static_cast<Handle*>(hh);
  h->m->do_init(cloud(P, Pn, Pc, nP), cloud(Q, Qn, Qc, nQ));
}
float s4pr_compute_transformation_attr(void* hh, const float* P, const float* Pn, const float* Pc, uint64_t nP,
                                       float* Q, const float* Qn, const float* Qc, uint64_t nQ, float* M_rowmajor, int64_t* n_candidates) {
  Handle* h = static_cast<Handle*>(hh);
  std::vector<Point3D> p = cloud(P, Pn, Pc, nP), q = cloud(Q, Qn, Qc, nQ);
  Match4PCSBase::MatrixType M = Match4PCSBase::MatrixType::Identity();
  h->lcps.clear();
  CandidateVisitor v{&h->lcps};
  const float r = h->m->ComputeTransformation(p, &q, M, Sampling::UniformDistSampler(), v);
  for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) M_rowmajor[4 * a + c] = M(a, c);
  for (uint64_t i = 0; i < nQ; ++i) { Q[3 * i] = q[i].x(); Q[3 * i + 1] = q[i].y(); Q[3 * i + 2] = q[i].z(); }
  *n_candidates = int64_t(h->lcps.size());
  return r;
}
void s4pr_get_stats(void* hh, int32_t* trials, int32_t* nP, int32_t* nQ, float* best_lcp, float* p_diameter) {
  Handle* h = static_cast<Handle*>(hh);
  *trials = h->m->number_of_trials_; *nP = int32_t(h->m->sampled_P_3D_.size()); *nQ = int32_t(h->m->sampled_Q_3D_.size());
  *best_lcp = h->m->best_LCP_; *p_diameter = h->m->P_diameter_;
}
void s4pr_get_cloud(void* hh, int which, float* xyz) {
  Handle* h = static_cast<Handle*>(hh);
  const std::vector<Point3D>& c = which == 0 ? h->m->sampled_P_3D_ : h->m->sampled_Q_3D_;
  for (size_t i = 0; i < c.size(); ++i) { xyz[3 * i] = c[i].x(); xyz[3 * i + 1] = c[i].y(); xyz[3 * i + 2] = c[i].z(); }
}
int32_t s4pr_select_quadrilateral(void* hh, float* inv1, float* inv2, int32_t* base, float* base_xyz) {
  Handle* h = static_cast<Handle*>(hh);
  int b1, b2, b3, b4;
  const bool ok = h->m->SelectQuadrilateral(*inv1, *inv2, b1, b2, b3, b4);
  base[0] = b1; base[1] = b2; base[2] = b3; base[3] = b4;
  for (int i = 0; i < 4; ++i) { base_xyz[3 * i] = h->m->base3D()[i].x(); base_xyz[3 * i + 1] = h->m->base3D()[i].y(); base_xyz[3 * i + 2] = h->m->base3D()[i].z(); }
  return ok;
}
int64_t s4pr_extract_pairs(void* hh, float d, float na, float eps, int32_t b1, int32_t b2, int32_t* out, int64_t cap) {
  Handle* h = static_cast<Handle*>(hh);
  Match4PCSBase::PairsVector pairs;
  h->m->ExtractPairs(d, na, eps, b1, b2, &pairs);
  for (size_t i = 0; i < pairs.size() && int64_t(i) < cap; ++i) { out[2 * i] = pairs[i].first; out[2 * i + 1] = pairs[i].second; }
  return int64_t(pairs.size());
}
int64_t s4pr_find_congruent(void* hh, float inv1, float inv2, float thr, const int32_t* p1, int64_t m1, const int32_t* p2, int64_t m2,
                            int32_t* out, int64_t cap) {
  Handle* h = static_cast<Handle*>(hh);
  Match4PCSBase::PairsVector a(m1), b(m2);
  for (int64_t i = 0; i < m1; ++i) a[i] = {p1[2 * i], p1[2 * i + 1]};
  for (int64_t i = 0; i < m2; ++i) b[i] = {p2[2 * i], p2[2 * i + 1]};
  std::vector<Quadrilateral> quads;
  h->m->FindCongruentQuadrilaterals(inv1, inv2, thr, thr, a, b, &quads);
  for (size_t i = 0; i < quads.size() && int64_t(i) < cap; ++i) for (int k = 0; k < 4; ++k) out[4 * i + k] = quads[i][k];
  return int64_t(quads.size());
}
// TryCongruentSet on explicit quads; lcps_out gets the LCP of every verified candidate in visiting order.
int64_t s4pr_try_congruent_set(void* hh, const int32_t* base, const int32_t* quads, int64_t K, float* lcps_out, int64_t cap, int64_t* n_lcps) {
  Handle* h = static_cast<Handle*>(hh);
  std::vector<Quadrilateral> q;
  for (int64_t i = 0; i < K; ++i) q.emplace_back(quads[4 * i], quads[4 * i + 1], quads[4 * i + 2], quads[4 * i + 3]);
  h->lcps.clear();
  CandidateVisitor v{&h->lcps};
  size_t nb = 0;
  h->m->try_set(base[0], base[1], base[2], base[3], q, v, nb);
  *n_lcps = int64_t(h->lcps.size());
  for (size_t i = 0; i < h->lcps.size() && int64_t(i) < cap; ++i) lcps_out[i] = h->lcps[i];
  return int64_t(nb);
}
float s4pr_verify(void* hh, const float* T_rowmajor) {
  Handle* h = static_cast<Handle*>(hh);
  Match4PCSBase::MatrixType M;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) M(r, c) = T_rowmajor[4 * r + c];
  return h->m->Verify(M);
}
void s4pr_get_best(void* hh, float* T_rowmajor, float* lcp, int32_t* base, int32_t* congruent) {
  Handle* h = static_cast<Handle*>(hh);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T_rowmajor[4 * r + c] = h->m->transform_(r, c);
  *lcp = h->m->best_LCP_;
  for (int i = 0; i < 4; ++i) { base[i] = h->m->base_[i]; congruent[i] = h->m->current_congruent_[i]; }
}
// The public entry point, ComputeTransformation (match4pcsBase.h:108-115).  Q is transformed in place.
float s4pr_compute_transformation(void* hh, const float* P, uint64_t nP, float* Q, uint64_t nQ, float* M_rowmajor, int64_t* n_candidates) {
  Handle* h = static_cast<Handle*>(hh);
  std::vector<Point3D> p = cloud(P, nullptr, nullptr, nP), q = cloud(Q, nullptr, nullptr, nQ);
  Match4PCSBase::MatrixType M = Match4PCSBase::MatrixType::Identity();
  h->lcps.clear();
  CandidateVisitor v{&h->lcps};
  const float r = h->m->ComputeTransformation(p, &q, M, Sampling::UniformDistSampler(), v);
  for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) M_rowmajor[4 * a + c] = M(a, c);
  for (uint64_t i = 0; i < nQ; ++i) { Q[3 * i] = q[i].x(); Q[3 * i + 1] = q[i].y(); Q[3 * i + 2] = q[i].z(); }
  *n_candidates = int64_t(h->lcps.size());
  return r;
}

// Bounded CPU-baseline sample: the reference's ComputeTransformation, cut after `budget` seconds of RANSAC time.
int32_t s4pr_bench(void* hh, const float* P, uint64_t nP, const float* Q, uint64_t nQ, double budget, uint64_t* n_candidates, double* seconds) {
  Handle* h = static_cast<Handle*>(hh);
  std::vector<Point3D> p = cloud(P, nullptr, nullptr, nP), q = cloud(Q, nullptr, nullptr, nQ);
  Match4PCSBase::MatrixType M = Match4PCSBase::MatrixType::Identity();
  *n_candidates = 0; *seconds = 0;
  BudgetVisitor v{n_candidates, seconds, budget, std::chrono::steady_clock::now(), false};
  try {
    h->m->ComputeTransformation(p, &q, M, Sampling::UniformDistSampler(), v);
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - v.t0).count();
    return 0;      // finished within the budget
  } catch (const BudgetExceeded&) {
    return 1;
  }
}

}  // extern "C"
