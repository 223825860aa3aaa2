// Drop-in for src/super4pcs/algorithms/4pcs.h: class GlobalRegistration::Match4PCS with the constructor of 4pcs.h:59-60,
// so that sources written against the reference (the Meshlab plugin includes this header and instantiates the class
// behind its "useSuper4PCS = false" switch, demos/MeshlabPlugin/.../globalregistration.cpp:26,167) still compile.
//
// The legacy 4PCS matcher itself (O(n^2) pair extraction, 4pcs.cc) is OUT OF SCOPE of the MI355X hot path (SURVEY.md
// section 2): constructing it fails loudly instead of silently running something else.
#ifndef S4P_FACADE_4PCS_H_
#define S4P_FACADE_4PCS_H_

#include <stdexcept>

#include "super4pcs/algorithms/match4pcsBase.h"

namespace GlobalRegistration {

class Match4PCS : public Match4PCSBase {
 public:
  using Base = Match4PCSBase;
  using Scalar = typename Base::Scalar;
  using PairsVector = typename Base::PairsVector;

  explicit Match4PCS(const Match4PCSOptions& options, const Utils::Logger logger) : Base(options, logger) {
    const char* msg = "GlobalRegistration::Match4PCS (legacy 4PCS, algorithms/4pcs.cc) is not provided by the MI355X build: "
                      "use GlobalRegistration::MatchSuper4PCS";
    logger.Log<Utils::ErrorReport>(msg);
    throw std::runtime_error(msg);
  }
  ~Match4PCS() {}

 protected:   // never reached: the constructor throws
  void Initialize(const std::vector<Point3D>&, const std::vector<Point3D>&) override {}
  void ExtractPairs(Scalar, Scalar, Scalar, int, int, PairsVector* pairs) const override { if (pairs) pairs->clear(); }
  bool FindCongruentQuadrilaterals(Scalar, Scalar, Scalar, Scalar, const PairsVector&, const PairsVector&,
                                   std::vector<Quadrilateral>* quadrilaterals) const override {
    if (quadrilaterals) quadrilaterals->clear();
    return false;
  }
};

}  // namespace GlobalRegistration
#endif
