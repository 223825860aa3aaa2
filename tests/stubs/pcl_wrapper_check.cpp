// Instantiates the PCL wrapper (demos/PCLWrapper) against the stub PCL headers in this directory and the facade:
// tests/test_wrappers_compile.py compiles it with -fsyntax-only (and, on a GPU box, can link and run it).
#include <pcl/registration/super4pcs.h>
int main() {
  typedef pcl::PointNormal PointNT;
  pcl::PointCloud<PointNT>::Ptr object(new pcl::PointCloud<PointNT>), scene(new pcl::PointCloud<PointNT>);
  pcl::PointCloud<PointNT> aligned;
  pcl::Super4PCS<PointNT, PointNT> align;
  align.setInputSource(object);
  align.setInputTarget(scene);
  align.options_.sample_size = 200;
  align.options_.delta = 0.01f;
  align.options_.configureOverlap(0.7f);
  align.align(aligned);
  return align.hasConverged() ? 0 : 1;
}
