// Implementation of pcl::Super4PCS::computeTransformation (reference: demos/PCLWrapper/pcl/registration/impl/super4pcs.hpp:66-109)
// over the MI355X facade: target_ plays P, input_ plays Q, one MatchSuper4PCS::ComputeTransformation call.
#ifndef PCL_REGISTRATION_IMPL_SUPER4PCS_H_
#define PCL_REGISTRATION_IMPL_SUPER4PCS_H_

#include <cstdio>
#include <vector>

#include <pcl/common/transforms.h>
#include <pcl/console/print.h>

#include <super4pcs/algorithms/super4pcs.h>
#include <super4pcs/sampling.h>
#include <super4pcs/utils/logger.h>

namespace pcl {
namespace super4pcs_detail {

// Progress line on stdout once per RANSAC trial; per-candidate calls (fraction < 0) are ignored, so the engine does not
// have to read the candidates back (match4pcsBase.h:73-76 visitor concept).
struct ProgressVisitor {
  template <class Matrix>
  void operator()(float fraction, float best_lcp, Matrix&&) const {
    if (fraction < 0.f) return;
    std::printf("done: %d%% best: %f                  \r", int(fraction * 100.f), best_lcp);
    std::fflush(stdout);
  }
  constexpr bool needsGlobalTransformation() const { return false; }
};

template <class Cloud>
void to_point_set(const Cloud& cloud, std::vector<GlobalRegistration::Point3D>& out) {
  out.clear();
  out.reserve(cloud.size());
  for (std::size_t i = 0; i < cloud.size(); ++i) out.emplace_back(cloud[i].x, cloud[i].y, cloud[i].z);
}

}  // namespace super4pcs_detail

template <typename PointSource, typename PointTarget>
void Super4PCS<PointSource, PointTarget>::computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess) {
  namespace GR = GlobalRegistration;
  final_transformation_ = guess;

  GR::Utils::Logger logger(GR::Utils::Verbose);
  GR::MatchSuper4PCS matcher(options_, logger);          // throws if no gfx950 device is visible: there is no CPU path
  GR::Sampling::UniformDistSampler sampler;
  super4pcs_detail::ProgressVisitor visitor;

  std::vector<GR::Point3D> reference_set, moving_set;
  super4pcs_detail::to_point_set(*target_, reference_set);
  super4pcs_detail::to_point_set(*input_, moving_set);

  const float score = matcher.ComputeTransformation(reference_set, &moving_set, final_transformation_, sampler, visitor);

  transformPointCloud(*input_, output, final_transformation_);
  pcl::console::print_highlight("Final score: %f\n", score);
  converged_ = true;
}

}  // namespace pcl

#endif  // PCL_REGISTRATION_IMPL_SUPER4PCS_H_
