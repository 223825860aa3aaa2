// Drop-in for src/super4pcs/algorithms/match4pcsBase.{h,hpp}: class GlobalRegistration::Match4PCSBase with the
// public surface of match4pcsBase.h:66-115 (typedefs, constants, getFirstSampled/getSecondSampled,
// ComputeTransformation<Sampler,Visitor>) and the protected hooks the reference's tests reach through
// Testing::TestMatcher (tests/testing.h:71-154).  Bodies forward to the C ABI of libsuper4pcs_amd.so
// (s4p_matcher.h / s4p_capi.h); nothing is computed on the CPU except the user's Sampler/Visitor templates.
#ifndef S4P_FACADE_MATCH4PCSBASE_H_
#define S4P_FACADE_MATCH4PCSBASE_H_

#include <array>
#include <stdexcept>
#include <type_traits>
#include <string>
#include <utility>
#include <vector>

#include "s4p_matcher.h"
#include "super4pcs/sampling.h"
#include "super4pcs/shared4pcs.h"
#include "super4pcs/utils/logger.h"

#ifdef S4P_HAVE_EIGEN
#include <Eigen/Geometry>
#endif

namespace GlobalRegistration {

class Match4PCSBase {
 public:
  using PairsVector = std::vector<std::pair<int, int>>;
  using Scalar = typename Point3D::Scalar;
  using VectorType = typename Point3D::VectorType;
#ifdef S4P_HAVE_EIGEN
  using MatrixType = Eigen::Matrix<Scalar, 4, 4>;
  using MatrixRef = Eigen::Ref<MatrixType>;
#else
  using MatrixType = compat::Matrix4f;
  using MatrixRef = compat::Matrix4f&;
#endif
  using LogLevel = Utils::LogLevel;
  struct DummyTransformVisitor {
    inline void operator()(float, float, MatrixRef) const {}
    constexpr bool needsGlobalTransformation() const { return false; }
  };
  using DefaultSampler = Sampling::UniformDistSampler;

  static constexpr int kNumberOfDiameterTrials = 1000;
  static constexpr Scalar kLargeNumber = 1e9;
  static constexpr Scalar distance_factor = 2.0;

  virtual ~Match4PCSBase() { s4p_matcher_destroy(engine_); }
  Match4PCSBase(const Match4PCSBase&) = delete;
  Match4PCSBase& operator=(const Match4PCSBase&) = delete;

  inline const std::vector<Point3D>& getFirstSampled() const { return sampled_P_3D_; }
  inline const std::vector<Point3D>& getSecondSampled() const { return sampled_Q_3D_; }

  // match4pcsBase.hpp:61-86
  template <typename Sampler = DefaultSampler, typename Visitor = DummyTransformVisitor>
  Scalar ComputeTransformation(const std::vector<Point3D>& P, std::vector<Point3D>* Q, MatrixRef transformation,
                               const Sampler& sampler = Sampler(), const Visitor& v = Visitor()) {
    if (Q == nullptr) return kLargeNumber;
    if (P.empty() || Q->empty()) return kLargeNumber;
    init(P, *Q, sampler);
    if (best_LCP_ != Scalar(1.)) Perform_N_steps(number_of_trials_, transformation, Q, v);
    return best_LCP_;
  }

 protected:
  Match4PCSBase(const Match4PCSOptions& options, const Utils::Logger& logger, int device = 0)
      : options_(options), logger_(logger) {
    s4p_options o{};
    o.delta = options.delta; o.max_normal_difference = options.max_normal_difference;
    o.max_translation_distance = options.max_translation_distance; o.max_angle = options.max_angle;
    o.max_color_distance = options.max_color_distance; o.sample_size = options.sample_size;
    o.max_time_seconds = options.max_time_seconds; o.random_seed = options.randomSeed;
    o.terminate_threshold = options.getTerminateThreshold(); o.overlap_estimation = options.getOverlapEstimation();
    const int32_t rc = s4p_matcher_create(&o, nullptr, device, &engine_);
    if (rc != S4P_OK) {   // no GPU / unsupported option: fail loudly, there is no CPU path behind this class
      const std::string msg = std::string("Match4PCSBase (MI355X): ") + s4p_matcher_last_error(nullptr);
      logger_.Log<Utils::ErrorReport>(msg);
      throw std::runtime_error(msg);
    }
  }

  template <Utils::LogLevel level, typename... Args>
  inline void Log(Args... args) const { logger_.Log<level>(args...); }

  // ---- match4pcsBase.hpp:90-203 ------------------------------------------------------------------
  template <typename Sampler>
  void init(const std::vector<Point3D>& P, const std::vector<Point3D>& Q, const Sampler& sampler) {
    std::vector<Point3D> ps, qu;
    const bool sample_q = Q.size() > options_.sample_size;
    if (P.size() > options_.sample_size) sampler(P, options_, ps);
    else { Log<LogLevel::ErrorReport>("(P) More samples requested than available: use whole cloud"); ps = P; }
    if (sample_q) sampler(Q, options_, qu);
    else { Log<LogLevel::ErrorReport>("(Q) More samples requested than available: use whole cloud"); qu = Q; }
    Soa sp(ps), sq(qu);
    const s4p_cloud_view vp = sp.view(), vq = sq.view();
    check(s4p_matcher_init(engine_, &vp, &vq, sample_q ? 1 : 0));
    Q_copy_ = Q;
    refresh();
    // sampled clouds as the engine holds them (centred; Q shuffled and truncated)
    pull_sampled(0, ps, sampled_P_3D_);
    pull_sampled(1, qu, sampled_Q_3D_);
    Log<LogLevel::Verbose>("norm_max_dist: ", options_.delta);
    Log<LogLevel::Verbose>("Initial LCP: ", best_LCP_);
  }

  // ---- match4pcsBase.hpp:208-274 -----------------------------------------------------------------
  template <typename Visitor>
  bool Perform_N_steps(int n, MatrixRef transformation, std::vector<Point3D>* Q, const Visitor& v) {
    if (Q == nullptr) return false;
    float M[16];
    to_rowmajor(transformation, M);
    VisitorThunk<Visitor> thunk{&v};
    int32_t improved = 0, done = 0;
    // the reference also calls v(-1, lcp, T) once per verified candidate (:458-465); only pay for it when someone listens
    check(s4p_matcher_visit_candidates(engine_, std::is_same<Visitor, DummyTransformVisitor>::value ? 0 : 1));
    check(s4p_matcher_perform_n_steps(engine_, n, &VisitorThunk<Visitor>::call, &thunk, v.needsGlobalTransformation() ? 1 : 0,
                                      M, &improved, &done));
    from_rowmajor(M, transformation);
    refresh();
    if (improved) {                                   // :259-268 -- the final apply runs on the GPU (k_apply)
      *Q = Q_copy_;
      const int64_t nq = int64_t(Q->size());
      std::vector<float> x(nq), y(nq), z(nq);
      for (int64_t i = 0; i < nq; ++i) { x[i] = (*Q)[i].x(); y[i] = (*Q)[i].y(); z[i] = (*Q)[i].z(); }
      check(s4p_transform_points(s4p_matcher_ctx(engine_), M, x.data(), y.data(), z.data(), nq));
      for (int64_t i = 0; i < nq; ++i) { (*Q)[i].x() = x[i]; (*Q)[i].y() = y[i]; (*Q)[i].z() = z[i]; }
    }
    return done != 0;
  }

  // ---- match4pcsBase.hpp:281-360 -----------------------------------------------------------------
  template <typename Visitor>
  bool TryOneBase(const Visitor&) {
    int32_t ok = 0;
    check(s4p_matcher_try_one_base(engine_, &ok, nullptr));
    refresh();
    return ok != 0;
  }

  // ---- match4pcsBase.cc:279-351 ------------------------------------------------------------------
  bool SelectQuadrilateral(Scalar& invariant1, Scalar& invariant2, int& base1, int& base2, int& base3, int& base4) {
    int32_t found = 0, ids[4] = {0, 0, 0, 0};
    float bx[12];
    check(s4p_matcher_select_quadrilateral(engine_, &found, &invariant1, &invariant2, ids, bx));
    base1 = ids[0]; base2 = ids[1]; base3 = ids[2]; base4 = ids[3];
    if (found) {
      base_3D_.resize(4);
      for (int t = 0; t < 4; ++t) base_3D_[t] = sampled_P_3D_[size_t(ids[t])];
      // base normals / colours feed the pair filters of the ExtractPairs hook (pairCreationFunctor.h:166-192)
      float bn[12], bc[12];
      for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 3; ++k) { bn[3 * t + k] = base_3D_[t].normal()(k); bc[3 * t + k] = base_3D_[t].rgb()(k); }
      check(s4p_set_base(s4p_matcher_ctx(engine_), bx, bn, bc));
    }
    return found != 0;
  }
  const std::vector<Point3D>& base3D() const { return base_3D_; }

  // ---- virtual hooks (match4pcsBase.h:270-326); MatchSuper4PCS implements them on the GPU ------------
  virtual void Initialize(const std::vector<Point3D>& P, const std::vector<Point3D>& Q) = 0;
  virtual void ExtractPairs(Scalar pair_distance, Scalar pair_normals_angle, Scalar pair_distance_epsilon, int base_point1,
                            int base_point2, PairsVector* pairs) const = 0;
  virtual bool FindCongruentQuadrilaterals(Scalar invariant1, Scalar invariant2, Scalar distance_threshold1,
                                           Scalar distance_threshold2, const PairsVector& P_pairs, const PairsVector& Q_pairs,
                                           std::vector<Quadrilateral>* quadrilaterals) const = 0;

  // ---- match4pcsBase.hpp:363-497 -----------------------------------------------------------------
  template <typename Visitor>
  bool TryCongruentSet(int base_id1, int base_id2, int base_id3, int base_id4, const std::vector<Quadrilateral>& congruent_quads,
                       const Visitor&, size_t& nbCongruent) {
    const int32_t ids[4] = {base_id1, base_id2, base_id3, base_id4};
    std::vector<int32_t> q(congruent_quads.size() * 4);
    for (size_t i = 0; i < congruent_quads.size(); ++i) for (int k = 0; k < 4; ++k) q[4 * i + size_t(k)] = congruent_quads[i][k];
    s4p_base_result r;
    check(s4p_try_congruent_set(s4p_matcher_ctx(engine_), ids, q.data(), int64_t(congruent_quads.size()), nullptr, &r));
    nbCongruent = size_t(r.n_verified);
    r.n_pairs1 = r.n_pairs2 = 1; r.n_quads = congruent_quads.size();
    int32_t ok = 0;
    check(s4p_matcher_commit(engine_, 1, ids, &r, &ok));
    refresh();
    return ok != 0;
  }

  // state mirrored from the engine after every call (names as in match4pcsBase.h:118-170)
  int number_of_trials_ = 0;
  Scalar P_diameter_ = 0;
  Scalar best_LCP_ = 0;
  int current_trial_ = 0;
  int base_[4] = {0, 0, 0, 0};
  int current_congruent_[4] = {0, 0, 0, 0};
  std::vector<Point3D> sampled_P_3D_, sampled_Q_3D_, base_3D_, Q_copy_;
  VectorType centroid_P_, centroid_Q_;
  const Match4PCSOptions options_;
  const Utils::Logger& logger_;
  s4p_matcher* engine_ = nullptr;

  void check(int32_t rc) const {
    if (rc == S4P_OK) return;
    const std::string msg = std::string("super4pcs_amd: ") + s4p_matcher_last_error(engine_) + " / " +
                            s4p_last_error(s4p_matcher_ctx(engine_));
    logger_.Log<Utils::ErrorReport>(msg);
    throw std::runtime_error(msg);
  }

 private:
  struct Soa {
    std::vector<float> a[9];
    bool has_n = false, has_c = false;
    explicit Soa(const std::vector<Point3D>& pts) {
      const size_t n = pts.size();
      for (auto& v : a) v.resize(n);
      for (size_t i = 0; i < n; ++i) {
        a[0][i] = pts[i].x(); a[1][i] = pts[i].y(); a[2][i] = pts[i].z();
        for (int k = 0; k < 3; ++k) { a[3 + k][i] = pts[i].normal()(k); a[6 + k][i] = pts[i].rgb()(k); }
        has_n = has_n || pts[i].normal().squaredNorm() > 0.f;
        has_c = has_c || pts[i].rgb()(0) >= 0.f;
      }
    }
    s4p_cloud_view view() const {
      return s4p_cloud_view{a[0].data(), a[1].data(), a[2].data(),
                            has_n ? a[3].data() : nullptr, has_n ? a[4].data() : nullptr, has_n ? a[5].data() : nullptr,
                            has_c ? a[6].data() : nullptr, has_c ? a[7].data() : nullptr, has_c ? a[8].data() : nullptr,
                            int64_t(a[0].size())};
    }
  };
  template <typename Visitor>
  struct VisitorThunk {
    const Visitor* v;
    static void call(void* user, float fraction, float lcp, float* M) {
      MatrixType T;
      from_rowmajor(M, T);
      (*static_cast<VisitorThunk*>(user)->v)(fraction, lcp, T);
    }
  };
  template <class Mat> static void to_rowmajor(const Mat& T, float* M) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) M[4 * r + c] = T(r, c); }
  template <class Mat> static void from_rowmajor(const float* M, Mat&& T) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T(r, c) = M[4 * r + c]; }

  void refresh() {
    s4p_matcher_info i;
    check(s4p_matcher_get_info(engine_, &i));
    number_of_trials_ = i.number_of_trials; current_trial_ = i.current_trial; best_LCP_ = i.best_lcp; P_diameter_ = i.p_diameter;
    for (int k = 0; k < 3; ++k) { centroid_P_(k) = i.centroid_p[k]; centroid_Q_(k) = i.centroid_q[k]; }
    for (int k = 0; k < 4; ++k) { base_[k] = i.base[k]; current_congruent_[k] = i.congruent[k]; }
  }
  // Sampled clouds as the engine holds them: centred positions plus the normals / colours that travelled with each
  // point through sampling, shuffle and truncation (match4pcsBase.hpp:112-138).
  void pull_sampled(int which, const std::vector<Point3D>&, std::vector<Point3D>& out) {
    s4p_matcher_info i;
    check(s4p_matcher_get_info(engine_, &i));
    const size_t n = size_t(which == 0 ? i.n_sampled_p : i.n_sampled_q);
    std::vector<float> v[9];
    for (auto& a : v) a.resize(n);
    int32_t has_n = 0, has_c = 0;
    check(s4p_matcher_get_sampled(engine_, which, v[0].data(), v[1].data(), v[2].data()));
    check(s4p_matcher_get_sampled_attrs(engine_, which, v[3].data(), v[4].data(), v[5].data(), v[6].data(), v[7].data(), v[8].data(), &has_n, &has_c));
    out.assign(n, Point3D());
    for (size_t k = 0; k < n; ++k) {
      out[k].x() = v[0][k]; out[k].y() = v[1][k]; out[k].z() = v[2][k];
      if (has_n) out[k].set_normal(typename Point3D::VectorType(v[3][k], v[4][k], v[5][k]));
      if (has_c) out[k].set_rgb(typename Point3D::VectorType(v[6][k], v[7][k], v[8][k]));
    }
  }
};

}  // namespace GlobalRegistration
#endif
