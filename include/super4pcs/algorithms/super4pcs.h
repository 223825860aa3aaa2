// Drop-in for src/super4pcs/algorithms/super4pcs.{h,cc}: class GlobalRegistration::MatchSuper4PCS with the
// constructor of super4pcs.h:62-63 and the three overrides; ExtractPairs / FindCongruentQuadrilaterals run the
// gfx950 kernels through s4p_extract_pairs / s4p_find_congruent (include/s4p_capi.h).
#ifndef S4P_FACADE_SUPER4PCS_H_
#define S4P_FACADE_SUPER4PCS_H_

#include <typeinfo>

#include "super4pcs/algorithms/match4pcsBase.h"

namespace GlobalRegistration {

class MatchSuper4PCS : public Match4PCSBase {
 public:
  using Base = Match4PCSBase;
  using Scalar = typename Base::Scalar;
  using PairsVector = typename Base::PairsVector;

  explicit MatchSuper4PCS(const Match4PCSOptions& options, const Utils::Logger& logger) : Base(options, logger) {}
  ~MatchSuper4PCS() {}

 protected:
  // The fused device loop only when the object IS a MatchSuper4PCS: any subclass may have overridden the hooks below
  // (tests/testing.h:71-154 does), and then the trial loop must call them (match4pcsBase.hpp:328-347).
  bool uses_stock_hooks() const override { return typeid(*this) == typeid(MatchSuper4PCS); }

  // super4pcs.cc:183-224
  void ExtractPairs(Scalar pair_distance, Scalar pair_normals_angle, Scalar pair_distance_epsilon, int base_point1,
                    int base_point2, PairsVector* pairs) const override {
    pairs->clear();
    s4p_ctx* ctx = s4p_matcher_ctx(engine_);
    // The call advances the persistent octree permutation, so it cannot simply be repeated with a larger buffer: the
    // state is saved first and restored if the first (modest) buffer turns out too small (*n_out tells the need).
    std::vector<uint32_t> state(size_t(s4p_pair_state_words(ctx)));
    check(s4p_pair_state_save(ctx, state.data()));
    std::vector<int32_t> buf(2 * (size_t(1) << 16));
    int64_t m = 0;
    int32_t rc = s4p_extract_pairs(ctx, pair_distance, pair_normals_angle, pair_distance_epsilon, base_point1, base_point2, buf.data(),
                                   int64_t(buf.size() / 2), &m);
    if (rc == S4P_ERR_CAPACITY && m > int64_t(buf.size() / 2)) {
      check(s4p_pair_state_restore(ctx, state.data()));
      buf.resize(2 * size_t(m));
      rc = s4p_extract_pairs(ctx, pair_distance, pair_normals_angle, pair_distance_epsilon, base_point1, base_point2, buf.data(),
                             int64_t(buf.size() / 2), &m);
    }
    check(rc);
    pairs->reserve(size_t(m));
    for (int64_t i = 0; i < m; ++i) pairs->emplace_back(buf[size_t(2 * i)], buf[size_t(2 * i + 1)]);
  }

  // super4pcs.cc:80-177
  bool FindCongruentQuadrilaterals(Scalar invariant1, Scalar invariant2, Scalar distance_threshold1, Scalar distance_threshold2,
                                   const PairsVector& P_pairs, const PairsVector& Q_pairs,
                                   std::vector<Quadrilateral>* quadrilaterals) const override {
    if (quadrilaterals == nullptr) return false;
    quadrilaterals->clear();
    std::vector<int32_t> p1(2 * P_pairs.size()), p2(2 * Q_pairs.size());
    for (size_t i = 0; i < P_pairs.size(); ++i) { p1[2 * i] = P_pairs[i].first; p1[2 * i + 1] = P_pairs[i].second; }
    for (size_t i = 0; i < Q_pairs.size(); ++i) { p2[2 * i] = Q_pairs[i].first; p2[2 * i + 1] = Q_pairs[i].second; }
    int64_t cap = 1 << 20, K = 0;
    std::vector<int32_t> out;
    while (true) {
      out.resize(size_t(4 * cap));
      const int32_t rc = s4p_find_congruent(s4p_matcher_ctx(engine_), invariant1, invariant2, distance_threshold1, distance_threshold2,
                                            p1.data(), int64_t(P_pairs.size()), p2.data(), int64_t(Q_pairs.size()), out.data(), cap, &K);
      if (rc == S4P_ERR_CAPACITY && K > cap) { cap = K; continue; }
      check(rc);
      break;
    }
    quadrilaterals->reserve(size_t(K));
    for (int64_t i = 0; i < K; ++i) quadrilaterals->emplace_back(out[size_t(4 * i)], out[size_t(4 * i + 1)], out[size_t(4 * i + 2)], out[size_t(4 * i + 3)]);
    return !quadrilaterals->empty();
  }

  // super4pcs.cc:230-234: pcfunctor_.synch3DContent() -- done inside s4p_matcher_init (s4p_set_clouds)
  void Initialize(const std::vector<Point3D>&, const std::vector<Point3D>&) override {}
};

}  // namespace GlobalRegistration
#endif
