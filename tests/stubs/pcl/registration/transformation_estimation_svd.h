// TEST STUB (not PCL), see registration.h
#pragma once
#include <pcl/registration/registration.h>
namespace pcl { namespace registration { template <class S, class T> struct TransformationEstimationSVD : TransformationEstimation<S, T> {}; } }
