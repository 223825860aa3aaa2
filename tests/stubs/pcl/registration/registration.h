// TEST STUB (not PCL): the subset of pcl::Registration / PointCloud / PointIndices that demos/PCLWrapper touches, so that the
// wrapper can be syntax-checked in an image without PCL (tests/test_wrappers_compile.py).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include <Eigen/Core>
namespace pcl {
struct PointXYZ { float x, y, z; };
struct PointNormal { float x, y, z, normal_x, normal_y, normal_z; };
struct PointIndices { typedef std::shared_ptr<PointIndices> Ptr; typedef std::shared_ptr<const PointIndices> ConstPtr; std::vector<int> indices; };
template <class PointT> struct PointCloud {
  typedef std::shared_ptr<PointCloud> Ptr; typedef std::shared_ptr<const PointCloud> ConstPtr;
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
  const PointT& operator[](std::size_t i) const { return points[i]; }
  PointT& operator[](std::size_t i) { return points[i]; }
};
namespace registration { template <class S, class T> struct TransformationEstimation { virtual ~TransformationEstimation() {} }; }
template <class PointSource, class PointTarget>
class Registration {
 public:
  typedef Eigen::Matrix4f Matrix4;
  typedef pcl::PointCloud<PointSource> PointCloudSource;
  typedef pcl::PointCloud<PointTarget> PointCloudTarget;
  virtual ~Registration() {}
  void setInputSource(const typename PointCloudSource::ConstPtr& c) { input_ = c; }
  void setInputTarget(const typename PointCloudTarget::ConstPtr& c) { target_ = c; }
  void align(PointCloudSource& out) { computeTransformation(out, Matrix4::Identity()); }
  bool hasConverged() const { return converged_; }
  Matrix4 getFinalTransformation() const { return final_transformation_; }
  const std::string& getClassName() const { return reg_name_; }
 protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  std::string reg_name_;
  typename PointCloudSource::ConstPtr input_;
  typename PointCloudTarget::ConstPtr target_;
  std::shared_ptr<registration::TransformationEstimation<PointSource, PointTarget>> transformation_estimation_;
  Matrix4 final_transformation_;
  bool converged_ = false;
};
}  // namespace pcl
