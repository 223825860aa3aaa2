// TEST STUB (not PCL), see registration/registration.h
#pragma once
#include <pcl/registration/registration.h>
namespace pcl {
template <class PointT>
void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix4f& M) {
  out = in;
  for (auto& p : out.points) {
    const float x = p.x, y = p.y, z = p.z;
    p.x = M(0, 0) * x + M(0, 1) * y + M(0, 2) * z + M(0, 3);
    p.y = M(1, 0) * x + M(1, 1) * y + M(1, 2) * z + M(1, 3);
    p.z = M(2, 0) * x + M(2, 1) * y + M(2, 2) * z + M(2, 3);
  }
}
}  // namespace pcl
