// Drop-in for src/super4pcs/sampling.h:59-122 (GlobalRegistration::Sampling::UniformDistSampler):
// the Sampler concept is  void operator()(const std::vector<Point>&, const Match4PCSOptions&, std::vector<Point>&) const.
#ifndef S4P_FACADE_SAMPLING_H_
#define S4P_FACADE_SAMPLING_H_
#include <cstdint>
#include <vector>

#include "s4p_matcher.h"
#include "super4pcs/shared4pcs.h"

namespace GlobalRegistration {
namespace Sampling {

struct UniformDistSampler {
  template <typename Point>
  inline void operator()(const std::vector<Point>& inputset, const Match4PCSOptions& options, std::vector<Point>& output) const {
    const int64_t n = int64_t(inputset.size());
    output.clear();
    if (n == 0) return;
    std::vector<float> x(n), y(n), z(n);
    for (int64_t i = 0; i < n; ++i) { x[i] = inputset[i].x(); y[i] = inputset[i].y(); z[i] = inputset[i].z(); }
    std::vector<int64_t> keep(n);
    const int64_t k = s4p_uniform_dist_sample(x.data(), y.data(), z.data(), n, options.delta, keep.data());
    output.reserve(size_t(k));
    for (int64_t i = 0; i < k; ++i) output.push_back(inputset[size_t(keep[size_t(i)])]);
  }
};

}  // namespace Sampling
}  // namespace GlobalRegistration
#endif
