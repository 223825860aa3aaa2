// Drop-in for src/super4pcs/algorithms/match4pcsBase.{h,hpp}: class GlobalRegistration::Match4PCSBase with the
// public surface of match4pcsBase.h:66-115 (typedefs, constants, getFirstSampled/getSecondSampled,
// ComputeTransformation<Sampler,Visitor>) and the protected hooks the reference's tests reach through
// Testing::TestMatcher (tests/testing.h:71-154).  Bodies forward to the C ABI of libsuper4pcs_amd.so
// (s4p_matcher.h / s4p_capi.h); nothing is computed on the CPU except the user's Sampler/Visitor templates.
#ifndef S4P_FACADE_MATCH4PCSBASE_H_
#define S4P_FACADE_MATCH4PCSBASE_H_

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include <string>
#include <system_error>
#include <utility>
#include <vector>

#include "s4p_matcher.h"
#include "super4pcs/sampling.h"
#include "super4pcs/shared4pcs.h"
#include "super4pcs/utils/logger.h"

#ifdef S4P_HAVE_EIGEN
#include <Eigen/Geometry>
#endif

// -DS4P_FACADE_TRACE: where a ComputeTransformation call spends its time on the host side of the C ABI (stderr); lab aid
#ifdef S4P_FACADE_TRACE
#include <cstdio>
#define S4P_FACADE_LAP(what)                                                                                     \
  do {                                                                                                           \
    const auto s4p_now_ = std::chrono::steady_clock::now();                                                      \
    std::fprintf(stderr, "[facade] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(s4p_now_ - s4p_lap_).count()); \
    s4p_lap_ = s4p_now_;                                                                                         \
  } while (0)
#define S4P_FACADE_LAP_BEGIN() auto s4p_lap_ = std::chrono::steady_clock::now()
#else
#define S4P_FACADE_LAP(what) do {} while (0)
#define S4P_FACADE_LAP_BEGIN() do {} while (0)
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
    S4P_FACADE_LAP_BEGIN();
    const bool sample_q = Q.size() > options_.sample_size;
    // Q_copy_ = Q (match4pcsBase.hpp:191) is a whole-cloud copy nothing below reads: it runs beside the sampling and the
    // engine's init and is complete before Initialize() -- the first code of a subclass that could look at it -- is called
    Async copy_of_q([this, &Q] { Q_copy_ = Q; });
    S4P_FACADE_LAP("start copy of Q");
    init_engine(P, Q, sampler, sample_q, std::is_same<Sampler, DefaultSampler>());
    refresh();
    // sampled clouds as the engine holds them (centred; Q shuffled and truncated)
    pull_sampled(0, sampled_P_3D_);
    pull_sampled(1, sampled_Q_3D_);
    Log<LogLevel::Verbose>("norm_max_dist: ", options_.delta);
    // The virtual handler, "called once the internal state of the Base class has been set" (match4pcsBase.h:262-272,
    // match4pcsBase.hpp:197-198): with the caller's P and Q, after sampling / centring / the trial count, and with
    // best_LCP_ still 0 -- the initial LCP (= Verify(transform_), :200) is assigned after it, as in the reference.
    S4P_FACADE_LAP("pull sampled clouds");
    copy_of_q.wait();
    S4P_FACADE_LAP("wait for the copy of Q");
    const Scalar initial_lcp = best_LCP_;
    best_LCP_ = Scalar(0);
    Initialize(P, Q);
    best_LCP_ = initial_lcp;
    Log<LogLevel::Verbose>("Initial LCP: ", best_LCP_);
  }

  // Sampling (match4pcsBase.hpp:112-127) + the engine's init.  A user-supplied Sampler is host code and runs here on the
  // std::vector<Point3D>, its output handed to the engine as SoA arrays.  The stock UniformDistSampler is the engine's own
  // device sampler (sampling.h forwards to it), so for it the whole clouds go down once as SoA views and the sampled
  // vectors are never built on this side of the ABI: same voxel rule, same kept points, same shuffle, one pass less over
  // half a million Point3D.
  template <typename Sampler>
  void init_engine(const std::vector<Point3D>& P, const std::vector<Point3D>& Q, const Sampler& sampler, bool sample_q, std::false_type) {
    S4P_FACADE_LAP_BEGIN();
    std::vector<Point3D> ps, qu;
    if (P.size() > options_.sample_size) sampler(P, options_, ps);
    else { Log<LogLevel::ErrorReport>("(P) More samples requested than available: use whole cloud"); ps = P; }
    S4P_FACADE_LAP("sampler(P)");
    if (sample_q) sampler(Q, options_, qu);
    else { Log<LogLevel::ErrorReport>("(Q) More samples requested than available: use whole cloud"); qu = Q; }
    S4P_FACADE_LAP("sampler(Q)");
    Soa sp(ps), sq(qu);
    const s4p_cloud_view vp = sp.view(), vq = sq.view();
    S4P_FACADE_LAP("SoA of the samples");
    check(s4p_matcher_init(engine_, &vp, &vq, sample_q ? 1 : 0));
    S4P_FACADE_LAP("s4p_matcher_init");
  }
  template <typename Sampler>
  void init_engine(const std::vector<Point3D>& P, const std::vector<Point3D>& Q, const Sampler&, bool sample_q, std::true_type) {
    S4P_FACADE_LAP_BEGIN();
    if (!(P.size() > options_.sample_size)) Log<LogLevel::ErrorReport>("(P) More samples requested than available: use whole cloud");
    if (!sample_q) Log<LogLevel::ErrorReport>("(Q) More samples requested than available: use whole cloud");
    std::unique_ptr<Soa> sp, sq;
    {
      Async soa_of_p([&sp, &P] { sp.reset(new Soa(P)); });
      sq.reset(new Soa(Q));
      soa_of_p.wait();
    }
    const s4p_cloud_view vp = sp->view(), vq = sq->view();
    S4P_FACADE_LAP("SoA of P and Q");
    check(s4p_matcher_init_full(engine_, &vp, &vq));
    S4P_FACADE_LAP("s4p_matcher_init_full");
  }

  // ---- match4pcsBase.hpp:208-274 -----------------------------------------------------------------
  // Two routes, same trials and results:
  //  * the dynamic type uses the stock hooks (uses_stock_hooks()): the whole loop runs inside the engine, bases
  //    pipelined, pairs -> quads -> candidates fused on the device (s4p_matcher_perform_n_steps);
  //  * a subclass overrides ExtractPairs / FindCongruentQuadrilaterals (the plugin points of match4pcsBase.h:300-326, the
  //    pattern of tests/testing.h:71-154): the loop below is the reference's own -- TryOneBase calls the VIRTUAL hooks
  //    stage by stage, so the overrides see, and may change, every pair and quad list; the device still does the work
  //    behind the stock implementations and behind TryCongruentSet.
  template <typename Visitor>
  bool Perform_N_steps(int n, MatrixRef transformation, std::vector<Point3D>* Q, const Visitor& v) {
    if (Q == nullptr) return false;
    S4P_FACADE_LAP_BEGIN();
    float M[16];
    to_rowmajor(transformation, M);
    int32_t improved = 0, done = 0;
    if (uses_stock_hooks()) {
      VisitorThunk<Visitor> thunk{&v};
      // the reference also calls v(-1, lcp, T) once per verified candidate (:458-465); only pay for it when someone listens
      check(s4p_matcher_visit_candidates(engine_, std::is_same<Visitor, DummyTransformVisitor>::value ? 0 : 1));
      check(s4p_matcher_perform_n_steps(engine_, n, &VisitorThunk<Visitor>::call, &thunk, v.needsGlobalTransformation() ? 1 : 0,
                                        M, &improved, &done));
      from_rowmajor(M, transformation);
      refresh();
    } else {
      using sclock = std::chrono::system_clock;
      const Scalar last_best = best_LCP_;
      v(0, best_LCP_, transformation);                                       // :232
      bool ok = false;
      const auto t0 = sclock::now();
      for (int i = current_trial_; i < current_trial_ + n; ++i) {            // :236-256
        ok = TryOneBase(v);
        const Scalar fraction_try = Scalar(i) / Scalar(number_of_trials_);
        const Scalar fraction_time = Scalar(std::chrono::duration_cast<std::chrono::seconds>(sclock::now() - t0).count() /
                                            (long)options_.max_time_seconds);   // integer division: reference quirk, :240-243
        const Scalar fraction = std::max(fraction_time, fraction_try);
        current_transform(v.needsGlobalTransformation(), M);
        from_rowmajor(M, transformation);
        v(fraction, best_LCP_, transformation);
        if (ok || i > number_of_trials_ || fraction >= 0.99 || best_LCP_ == 1.0) break;
      }
      check(s4p_matcher_advance_trials(engine_, n));                         // current_trial_ += n, :258
      refresh();
      improved = best_LCP_ > last_best ? 1 : 0;
      if (improved) { current_transform(true, M); from_rowmajor(M, transformation); }      // getGlobalTransform, :259-262
      done = (ok || current_trial_ >= number_of_trials_) ? 1 : 0;
    }
    S4P_FACADE_LAP("trial loop");
    if (improved) {                                   // :259-268 -- the final apply runs on the GPU (k_apply)
      // *Q = Q_copy_ with the transformed positions: one pass out of the AoS copy, one pass back into *Q
      const size_t nq = Q_copy_.size();
      const std::unique_ptr<float[]> x(new float[nq]), y(new float[nq]), z(new float[nq]);
      const std::vector<Point3D>& src = Q_copy_;
      // Always out of Q_copy_ itself, as the reference does (`*Q = Q_copy_`, :264): protected state a subclass may have
      // rewritten in Initialize() or from a visitor; a threaded AoS -> SoA pass (the SoA copy kept from init was
      // validated on 32 points only and held 12 bytes per point for the life of the matcher -- removed)
      detail::for_ranges(nq, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) { x[i] = src[i].x(); y[i] = src[i].y(); z[i] = src[i].z(); }
      });
      S4P_FACADE_LAP("positions out of Q_copy_");
      check(s4p_transform_points(s4p_matcher_ctx(engine_), M, x.get(), y.get(), z.get(), int64_t(nq)));
      S4P_FACADE_LAP("s4p_transform_points");
      Q->resize(nq);
      std::vector<Point3D>& dst = *Q;
      detail::for_ranges(nq, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) { Point3D p = src[i]; p.x() = x[i]; p.y() = y[i]; p.z() = z[i]; dst[i] = p; }
      });
      S4P_FACADE_LAP("*Q = Q_copy_, transformed");
    }
    return done != 0;
  }

  // ---- match4pcsBase.hpp:281-360 -----------------------------------------------------------------
  template <typename Visitor>
  bool TryOneBase(const Visitor& v) {
    if (uses_stock_hooks()) {
      int32_t ok = 0;
      check(s4p_matcher_try_one_base(engine_, &ok, nullptr));
      refresh();
      return ok != 0;
    }
    Scalar invariant1, invariant2;
    int base_id1, base_id2, base_id3, base_id4;
    if (!SelectQuadrilateral(invariant1, invariant2, base_id1, base_id2, base_id3, base_id4)) return false;      // :313-316
    const Scalar distance1 = diff_norm(base_3D_[0].pos(), base_3D_[1].pos());                                     // :318-321
    const Scalar distance2 = diff_norm(base_3D_[2].pos(), base_3D_[3].pos());
    const Scalar normal_angle1 = diff_norm(base_3D_[0].normal(), base_3D_[1].normal());                           // :326-327
    const Scalar normal_angle2 = diff_norm(base_3D_[2].normal(), base_3D_[3].normal());
    PairsVector pairs1, pairs2;
    std::vector<Quadrilateral> congruent_quads;
    ExtractPairs(distance1, normal_angle1, distance_factor * options_.delta, 0, 1, &pairs1);                      // virtual
    ExtractPairs(distance2, normal_angle2, distance_factor * options_.delta, 2, 3, &pairs2);
    if (pairs1.size() == 0 || pairs2.size() == 0) return false;                                                   // :335-337
    if (!FindCongruentQuadrilaterals(invariant1, invariant2, distance_factor * options_.delta, distance_factor * options_.delta,
                                     pairs1, pairs2, &congruent_quads)) return false;                             // virtual, :340-347
    size_t nb = 0;
    return TryCongruentSet(base_id1, base_id2, base_id3, base_id4, congruent_quads, v, nb);
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

  // True iff the object's ExtractPairs / FindCongruentQuadrilaterals are the stock device implementations, so that the
  // trial loop may run fused inside the engine.  MatchSuper4PCS answers by dynamic type; a subclass that does not touch
  // the hooks may override this to return true and keep the fused loop.
  virtual bool uses_stock_hooks() const { return false; }

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
                       const Visitor& v, size_t& nbCongruent) {
    const int32_t ids[4] = {base_id1, base_id2, base_id3, base_id4};
    // (an empty set -- a hook that answered "found" with nothing in the list: the reference's loop does nothing, :363-497)
    if (congruent_quads.empty()) { nbCongruent = 0; return best_LCP_ > options_.getTerminateThreshold(); }
    std::vector<int32_t> q(congruent_quads.size() * 4);
    for (size_t i = 0; i < congruent_quads.size(); ++i) for (int k = 0; k < 4; ++k) q[4 * i + size_t(k)] = congruent_quads[i][k];
    s4p_base_result r;
    check(s4p_try_congruent_set(s4p_matcher_ctx(engine_), ids, q.data(), int64_t(congruent_quads.size()), nullptr, &r));
    nbCongruent = size_t(r.n_verified);
    if (!std::is_same<Visitor, DummyTransformVisitor>::value && r.n_verified) {       // v(-1, lcp, T) per candidate, :458-465
      std::vector<uint32_t> cnt(size_t(r.n_verified)); std::vector<float> Ts(size_t(r.n_verified) * 16);
      int64_t nv = 0;
      check(s4p_last_verified(s4p_matcher_ctx(engine_), cnt.data(), Ts.data(), int64_t(r.n_verified), &nv));
      for (int64_t k = 0; k < nv; ++k) {
        float* T = Ts.data() + 16 * size_t(k);
        if (v.needsGlobalTransformation())                                            // getGlobalTransform, :446-456
          for (int a = 0; a < 3; ++a) {
            const float rq = T[4 * a] * centroid_Q_(0) + (T[4 * a + 1] * centroid_Q_(1) + T[4 * a + 2] * centroid_Q_(2));
            T[4 * a + 3] = (T[4 * a + 3] + centroid_P_(a)) - rq;
          }
        MatrixType Tm;
        from_rowmajor(T, Tm);
        v(-1, float(cnt[size_t(k)]) / float(sampled_Q_3D_.size()), Tm);
      }
    }
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
  // a callable on a thread of its own (or inline when none can be started); wait() joins and rethrows what it threw
  struct Async {
    std::exception_ptr failed;
    std::thread t;
    template <class F> explicit Async(F f) {
      auto body = [this, f] { try { f(); } catch (...) { failed = std::current_exception(); } };
      try { t = std::thread(body); } catch (const std::system_error&) { body(); }       // no thread to be had: run it here
    }
    void wait() { if (t.joinable()) t.join(); if (failed) { std::exception_ptr f = failed; failed = nullptr; std::rethrow_exception(f); } }
    ~Async() { if (t.joinable()) t.join(); }
    Async(const Async&) = delete;
    Async& operator=(const Async&) = delete;
  };
  // SoA view of a cloud for the C ABI: positions always; normals / colours only when some point carries them (the flags are
  // found in a first pass, so that a plain xyz cloud costs three arrays, not nine); whole-cloud sizes on a few threads
  struct Soa {
    std::unique_ptr<float[]> a[9];
    size_t n = 0;
    bool has_n = false, has_c = false;
    explicit Soa(const std::vector<Point3D>& pts) : n(pts.size()) {
      for (int k = 0; k < 3; ++k) a[k].reset(new float[n ? n : 1]);
      std::atomic<unsigned> flags{0u};
      detail::for_ranges(n, [&](size_t b, size_t e) {                          // positions, and whether anything else is there
        unsigned f = 0;
        for (size_t i = b; i < e; ++i) {
          a[0][i] = pts[i].x(); a[1][i] = pts[i].y(); a[2][i] = pts[i].z();
          f |= (pts[i].normal().squaredNorm() > 0.f ? 1u : 0u) | (pts[i].rgb()(0) >= 0.f ? 2u : 0u);
        }
        flags.fetch_or(f, std::memory_order_relaxed);
      });
      has_n = (flags.load() & 1u) != 0; has_c = (flags.load() & 2u) != 0;
      if (!has_n && !has_c) return;
      if (has_n) for (int k = 3; k < 6; ++k) a[k].reset(new float[n ? n : 1]);
      if (has_c) for (int k = 6; k < 9; ++k) a[k].reset(new float[n ? n : 1]);
      detail::for_ranges(n, [&](size_t b, size_t e) {
        if (has_n) for (size_t i = b; i < e; ++i) for (int k = 0; k < 3; ++k) a[3 + k][i] = pts[i].normal()(k);
        if (has_c) for (size_t i = b; i < e; ++i) for (int k = 0; k < 3; ++k) a[6 + k][i] = pts[i].rgb()(k);
      });
    }
    s4p_cloud_view view() const {
      return s4p_cloud_view{a[0].get(), a[1].get(), a[2].get(),
                            has_n ? a[3].get() : nullptr, has_n ? a[4].get() : nullptr, has_n ? a[5].get() : nullptr,
                            has_c ? a[6].get() : nullptr, has_c ? a[7].get() : nullptr, has_c ? a[8].get() : nullptr, int64_t(n)};
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

  // (a - b).norm() of two 3-vectors in Eigen's fixed-size evaluation order x + (y + z); written on coefficients so that it
  // also serves the Eigen-free build
  template <class V> static Scalar diff_norm(const V& a, const V& b) {
    const Scalar d0 = a(0) - b(0), d1 = a(1) - b(1), d2 = a(2) - b(2);
    return std::sqrt(d0 * d0 + (d1 * d1 + d2 * d2));
  }
  // transform_ (centred frame) or the global transform of the current best (match4pcsBase.hpp:224-229, 246-252), row-major
  void current_transform(bool global, float* M) {
    if (global) { check(s4p_matcher_global_transform(engine_, M)); return; }
    s4p_matcher_info i;
    check(s4p_matcher_get_info(engine_, &i));
    for (int k = 0; k < 16; ++k) M[k] = i.transform[k];
  }
  void refresh() {
    s4p_matcher_info i;
    check(s4p_matcher_get_info(engine_, &i));
    number_of_trials_ = i.number_of_trials; current_trial_ = i.current_trial; best_LCP_ = i.best_lcp; P_diameter_ = i.p_diameter;
    for (int k = 0; k < 3; ++k) { centroid_P_(k) = i.centroid_p[k]; centroid_Q_(k) = i.centroid_q[k]; }
    for (int k = 0; k < 4; ++k) { base_[k] = i.base[k]; current_congruent_[k] = i.congruent[k]; }
  }
  // Sampled clouds as the engine holds them: centred positions plus the normals / colours that travelled with each
  // point through sampling, shuffle and truncation (match4pcsBase.hpp:112-138).
  void pull_sampled(int which, std::vector<Point3D>& out) {
    s4p_matcher_info i;
    check(s4p_matcher_get_info(engine_, &i));
    const size_t n = size_t(which == 0 ? i.n_sampled_p : i.n_sampled_q);
    int32_t has_n = 0, has_c = 0;
    check(s4p_matcher_get_sampled_attrs(engine_, which, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &has_n, &has_c));   // the flags only
    std::unique_ptr<float[]> v[9];
    for (int k = 0; k < 9; ++k)
      if (k < 3 || (k < 6 ? has_n : has_c)) v[k].reset(new float[n ? n : 1]);
    check(s4p_matcher_get_sampled(engine_, which, v[0].get(), v[1].get(), v[2].get()));
    if (has_n || has_c)
      check(s4p_matcher_get_sampled_attrs(engine_, which, v[3].get(), v[4].get(), v[5].get(), v[6].get(), v[7].get(), v[8].get(), nullptr, nullptr));
    out.assign(n, Point3D());
    detail::for_ranges(n, [&](size_t b, size_t e) {
      for (size_t k = b; k < e; ++k) {
        out[k].x() = v[0][k]; out[k].y() = v[1][k]; out[k].z() = v[2][k];
        if (has_n) out[k].set_normal(typename Point3D::VectorType(v[3][k], v[4][k], v[5][k]));
        if (has_c) out[k].set_rgb(typename Point3D::VectorType(v[6][k], v[7][k], v[8][k]));
      }
    });
  }
};

}  // namespace GlobalRegistration
#endif
