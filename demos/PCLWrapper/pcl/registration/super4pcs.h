// pcl::Super4PCS -- the PCL registration wrapper of the reference (demos/PCLWrapper/pcl/registration/super4pcs.h:64-110),
// on top of the MI355X facade (include/super4pcs/**).  Same class name, template parameters, public member
// (`options_`), base class and protected override, so PCL programs written against the reference wrapper
// (demos/PCLWrapper/pcl_super4pcs.cc usage: `pcl::Super4PCS<PointNT, PointNT> align; align.options_...; align.align(out)`)
// build unchanged.  The congruent-set search and the LCP verification behind it run on the GPU.
#ifndef PCL_REGISTRATION_SUPER4PCS_H_
#define PCL_REGISTRATION_SUPER4PCS_H_

#include <pcl/registration/registration.h>
#include <pcl/registration/transformation_estimation_svd.h>

#include <super4pcs/shared4pcs.h>

namespace pcl {

template <typename PointSource, typename PointTarget>
class Super4PCS : public Registration<PointSource, PointTarget> {
  using Reg = Registration<PointSource, PointTarget>;

 public:
  typedef typename Reg::Matrix4 Matrix4;
  typedef typename Reg::PointCloudSource PointCloudSource;
  typedef typename PointCloudSource::Ptr PointCloudSourcePtr;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef typename Reg::PointCloudTarget PointCloudTarget;
  typedef PointIndices::Ptr PointIndicesPtr;
  typedef PointIndices::ConstPtr PointIndicesConstPtr;

  using Reg::converged_;
  using Reg::final_transformation_;
  using Reg::getClassName;
  using Reg::input_;
  using Reg::reg_name_;
  using Reg::target_;
  using Reg::transformation_estimation_;

  /** Parameters of the matcher (delta, overlap, sample size, ...): set them before align(). */
  GlobalRegistration::Match4PCSOptions options_;

  Super4PCS() {
    reg_name_ = "Super4PCS";
    transformation_estimation_.reset(new pcl::registration::TransformationEstimationSVD<PointSource, PointTarget>);
  }
  virtual ~Super4PCS() {}

 protected:
  /** Registration::align() lands here: `output` receives the input cloud moved by the transformation found; `guess`
   *  is the initial value of final_transformation_. */
  void computeTransformation(PointCloudSource& output, const Eigen::Matrix4f& guess);
};

}  // namespace pcl

#include <pcl/registration/impl/super4pcs.hpp>

#endif  // PCL_REGISTRATION_SUPER4PCS_H_
