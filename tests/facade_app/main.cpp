// Mirrors the reference's tests/externalAppTest/main.cpp (an external application compiled against the
// installed headers) plus the Testing::TestMatcher pattern of tests/testing.h:71-154 that re-exposes the
// protected steps.  With a GPU it runs a registration; without one the constructor must throw.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "super4pcs/algorithms/super4pcs.h"
#include "super4pcs/shared4pcs.h"
#include "super4pcs/utils/logger.h"

using namespace GlobalRegistration;

struct CountingVisitor {
  mutable int calls = 0;
  inline void operator()(float, float, Match4PCSBase::MatrixRef) const { ++calls; }
  constexpr bool needsGlobalTransformation() const { return false; }
};

template <class BaseMatcher>
class TestMatcher : public BaseMatcher {
 public:
  using BaseMatcher::BaseMatcher;
  using PairsVector = typename BaseMatcher::PairsVector;
  using Scalar = typename BaseMatcher::Scalar;
  template <class S = typename BaseMatcher::DefaultSampler>
  void init(const std::vector<Point3D>& P, const std::vector<Point3D>& Q, const S& s = S()) { BaseMatcher::init(P, Q, s); }
  bool SelectQuadrilateral(Scalar& a, Scalar& b, int& b1, int& b2, int& b3, int& b4) {
    return BaseMatcher::SelectQuadrilateral(a, b, b1, b2, b3, b4);
  }
  void ExtractPairs(Scalar d, Scalar na, Scalar e, int p1, int p2, PairsVector* out) const override {
    BaseMatcher::ExtractPairs(d, na, e, p1, p2, out);
  }
  bool FindCongruentQuadrilaterals(Scalar i1, Scalar i2, Scalar t1, Scalar t2, const PairsVector& a, const PairsVector& b,
                                   std::vector<Quadrilateral>* q) const override {
    return BaseMatcher::FindCongruentQuadrilaterals(i1, i2, t1, t2, a, b, q);
  }
};

// A subclass that USES the plugin points (match4pcsBase.h:300-326): its ExtractPairs drops some of the pairs the stock
// implementation found, its FindCongruentQuadrilaterals records what it is handed.  The trial loop must route through
// both (match4pcsBase.hpp:328-347) -- the fused device loop would silently ignore them.
class FilteringMatcher : public MatchSuper4PCS {
 public:
  using MatchSuper4PCS::MatchSuper4PCS;
  mutable long extract_calls = 0, pairs_found = 0, pairs_kept = 0, find_calls = 0, pairs_seen_by_find = 0;
 protected:
  void ExtractPairs(Scalar d, Scalar na, Scalar e, int p1, int p2, PairsVector* out) const override {
    MatchSuper4PCS::ExtractPairs(d, na, e, p1, p2, out);
    ++extract_calls; pairs_found += long(out->size());
    PairsVector kept;
    for (const auto& pr : *out) if ((pr.first + pr.second) % 4 != 0) kept.push_back(pr);
    out->swap(kept);
    pairs_kept += long(out->size());
  }
  bool FindCongruentQuadrilaterals(Scalar i1, Scalar i2, Scalar t1, Scalar t2, const PairsVector& a, const PairsVector& b,
                                   std::vector<Quadrilateral>* q) const override {
    ++find_calls; pairs_seen_by_find += long(a.size() + b.size());
    return MatchSuper4PCS::FindCongruentQuadrilaterals(i1, i2, t1, t2, a, b, q);
  }
};

// A subclass that overrides the third plugin point, Initialize (match4pcsBase.h:262-272): the reference calls it from init()
// once per ComputeTransformation, with the caller's clouds, after the sampled clouds exist and before the initial LCP is
// assigned (match4pcsBase.hpp:197-200).  It keeps the stock pair / quad hooks, and says so, so the fused loop still runs.
class InitCountingMatcher : public MatchSuper4PCS {
 public:
  using MatchSuper4PCS::MatchSuper4PCS;
  int init_calls = 0;
  size_t seen_p = 0, seen_q = 0, sampled_p_at_call = 0, sampled_q_at_call = 0;
  float lcp_at_call = -1.f;
 protected:
  void Initialize(const std::vector<Point3D>& P, const std::vector<Point3D>& Q) override {
    ++init_calls; seen_p = P.size(); seen_q = Q.size();
    sampled_p_at_call = sampled_P_3D_.size(); sampled_q_at_call = sampled_Q_3D_.size();
    lcp_at_call = best_LCP_;
    MatchSuper4PCS::Initialize(P, Q);
  }
  bool uses_stock_hooks() const override { return true; }
};

static bool same_matrix(const Match4PCSBase::MatrixType& a, const Match4PCSBase::MatrixType& b) {
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (!(a(r, c) == b(r, c))) return false;
  return true;
}

int main(int argc, char** argv) {
  const bool expect_gpu = argc > 1 && std::atoi(argv[1]) != 0;
  Match4PCSOptions opt;
  opt.delta = 0.05f;
  opt.sample_size = 150;
  if (!opt.configureOverlap(0.7f)) return 2;
  Utils::Logger logger(Utils::NoLog);
  std::mt19937 g(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<Point3D> P, Q;
  for (int i = 0; i < 3000; ++i) {
    float x = nd(g), y = nd(g), z = nd(g);
    const float n = std::sqrt(x * x + y * y + z * z);
    x /= n; y /= n; z = z / n * 0.5f;
    P.emplace_back(x, y, z);
    Q.emplace_back(0.8f * x - 0.6f * y + 0.3f, 0.6f * x + 0.8f * y - 0.2f, z + 0.1f);   // rotated about z + shifted
  }
  try {
    TestMatcher<MatchSuper4PCS> matcher(opt, logger);
    if (!expect_gpu) { std::puts("constructed without a GPU: unexpected"); return 3; }
    Match4PCSBase::MatrixType mat = Match4PCSBase::MatrixType::Identity();
    CountingVisitor vis;
    std::vector<Point3D> Q2 = Q;
    const float score = matcher.ComputeTransformation(P, &Q2, mat, Sampling::UniformDistSampler(), vis);
    std::printf("score %.4f visitor calls %d sampled %zu/%zu\n", score, vis.calls, matcher.getFirstSampled().size(),
                matcher.getSecondSampled().size());
    if (!(score > 0.5f) || vis.calls < 2) return 4;
    // The same registration through the stock class: TestMatcher overrides the hooks (by forwarding), so it ran the
    // staged loop of the facade; MatchSuper4PCS itself runs the fused device loop.  Same trials, same candidates: the
    // score, the 4x4 and the transformed cloud must agree bit for bit, and so must the number of visitor calls.
    {
      MatchSuper4PCS stock(opt, logger);
      Match4PCSBase::MatrixType mat_s = Match4PCSBase::MatrixType::Identity();
      CountingVisitor vis_s;
      std::vector<Point3D> Q3 = Q;
      const float score_s = stock.ComputeTransformation(P, &Q3, mat_s, Sampling::UniformDistSampler(), vis_s);
      bool same_cloud = Q3.size() == Q2.size();
      for (size_t i = 0; same_cloud && i < Q3.size(); ++i)
        same_cloud = Q3[i].x() == Q2[i].x() && Q3[i].y() == Q2[i].y() && Q3[i].z() == Q2[i].z();
      std::printf("routes: staged %.6f (%d visitor calls) fused %.6f (%d)\n", score, vis.calls, score_s, vis_s.calls);
      if (score_s != score || !same_matrix(mat, mat_s) || !same_cloud || vis.calls != vis_s.calls) return 9;
    }
    // a user-supplied Sampler (any type that is not the stock one) runs on this side of the ABI and its output goes down as
    // the sampled clouds; the stock sampler is handed to the engine together with the whole clouds.  A sampler that merely
    // forwards to the stock one must therefore give the same registration through the other route.
    {
      struct ForwardingSampler {
        mutable int calls = 0;
        void operator()(const std::vector<Point3D>& in, const Match4PCSOptions& o, std::vector<Point3D>& out) const {
          ++calls;
          Sampling::UniformDistSampler()(in, o, out);
        }
      } fwd;
      MatchSuper4PCS user(opt, logger), stock(opt, logger);
      Match4PCSBase::MatrixType mat_u = Match4PCSBase::MatrixType::Identity(), mat_s = Match4PCSBase::MatrixType::Identity();
      std::vector<Point3D> Qu = Q, Qs = Q;
      const float score_u = user.ComputeTransformation(P, &Qu, mat_u, fwd);
      const float score_s = stock.ComputeTransformation(P, &Qs, mat_s);
      bool same = fwd.calls == 2 && score_u == score_s && same_matrix(mat_u, mat_s) && Qu.size() == Qs.size() &&
                  user.getFirstSampled().size() == stock.getFirstSampled().size() &&
                  user.getSecondSampled().size() == stock.getSecondSampled().size();
      for (size_t i = 0; same && i < Qu.size(); ++i) same = Qu[i].x() == Qs[i].x() && Qu[i].y() == Qs[i].y() && Qu[i].z() == Qs[i].z();
      for (size_t i = 0; same && i < user.getFirstSampled().size(); ++i)
        same = user.getFirstSampled()[i].x() == stock.getFirstSampled()[i].x() && user.getFirstSampled()[i].z() == stock.getFirstSampled()[i].z();
      for (size_t i = 0; same && i < user.getSecondSampled().size(); ++i)
        same = user.getSecondSampled()[i].y() == stock.getSecondSampled()[i].y();
      std::printf("samplers: user-supplied %.6f (%d calls) stock %.6f, sampled %zu / %zu\n", score_u, fwd.calls, score_s,
                  user.getFirstSampled().size(), user.getSecondSampled().size());
      if (!same) return 17;
    }
    // a subclass that filters pairs: its hooks must be called by the trial loop and must shape the result's inputs
    {
      FilteringMatcher fm(opt, logger);
      Match4PCSBase::MatrixType mat_f = Match4PCSBase::MatrixType::Identity();
      std::vector<Point3D> Q4 = Q;
      const float score_f = fm.ComputeTransformation(P, &Q4, mat_f);
      std::printf("filtering subclass: score %.4f extract calls %ld pairs %ld -> %ld, find calls %ld saw %ld pairs\n", score_f,
                  fm.extract_calls, fm.pairs_found, fm.pairs_kept, fm.find_calls, fm.pairs_seen_by_find);
      if (fm.extract_calls < 2 || fm.extract_calls % 2 != 0) return 10;          // two calls per trial (match4pcsBase.hpp:328-331)
      if (!(fm.pairs_kept < fm.pairs_found) || fm.find_calls < 1) return 11;
      if (fm.pairs_seen_by_find > fm.pairs_kept) return 12;                       // FindCongruentQuadrilaterals got the FILTERED lists
    }
    // the Initialize plugin point: called exactly once per ComputeTransformation, at the point the reference calls it
    {
      InitCountingMatcher im(opt, logger);
      Match4PCSBase::MatrixType mat_i = Match4PCSBase::MatrixType::Identity();
      std::vector<Point3D> Q5 = Q;
      const float score_i = im.ComputeTransformation(P, &Q5, mat_i);
      std::printf("Initialize hook: %d call(s), saw %zu / %zu points, %zu / %zu sampled, best_LCP_ %.3f at the call; score %.4f\n",
                  im.init_calls, im.seen_p, im.seen_q, im.sampled_p_at_call, im.sampled_q_at_call, im.lcp_at_call, score_i);
      if (im.init_calls != 1 || im.seen_p != P.size() || im.seen_q != Q.size()) return 13;
      if (im.sampled_p_at_call == 0 || im.sampled_q_at_call == 0 || im.lcp_at_call != 0.f) return 14;
      if (score_i != score || !same_matrix(mat, mat_i)) return 15;                 // same registration as the stock class
      std::vector<Point3D> Q6 = Q;
      (void)im.ComputeTransformation(P, &Q6, mat_i);
      if (im.init_calls != 2) return 16;
    }
    // empty sets (tests/externalAppTest/main.cpp): kLargeNumber
    std::vector<Point3D> e1, e2;
    if (matcher.ComputeTransformation(e1, &e2, mat) != Match4PCSBase::kLargeNumber) return 5;
    // protected steps: pairs / quads through the virtuals
    TestMatcher<MatchSuper4PCS> m2(opt, logger);
    m2.init(P, Q);
    float i1, i2; int b1, b2, b3, b4;
    if (!m2.SelectQuadrilateral(i1, i2, b1, b2, b3, b4)) return 6;
    Match4PCSBase::PairsVector p1, p2;
    m2.ExtractPairs(0.8f, 0.f, Match4PCSBase::distance_factor * opt.delta, 0, 1, &p1);
    m2.ExtractPairs(0.6f, 0.f, Match4PCSBase::distance_factor * opt.delta, 2, 3, &p2);
    std::vector<Quadrilateral> quads;
    m2.FindCongruentQuadrilaterals(i1, i2, 0.1f, 0.1f, p1, p2, &quads);
    std::printf("pairs %zu %zu quads %zu\n", p1.size(), p2.size(), quads.size());
    if (p1.empty() || p2.empty()) return 7;
    return 0;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return expect_gpu ? 8 : 0;
  }
}
