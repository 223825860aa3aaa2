// Drop-in for src/super4pcs/sampling.h:59-122 (GlobalRegistration::Sampling::UniformDistSampler):
// the Sampler concept is  void operator()(const std::vector<Point>&, const Match4PCSOptions&, std::vector<Point>&) const.
#ifndef S4P_FACADE_SAMPLING_H_
#define S4P_FACADE_SAMPLING_H_
#include <algorithm>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

#include "s4p_matcher.h"
#include "super4pcs/shared4pcs.h"

namespace GlobalRegistration {
namespace detail {

// fn(begin, end) over [0, n): on a few threads when n is a whole cloud (the AoS <-> SoA passes of the drop-in are strided
// reads plus first-touch page faults of freshly allocated arrays, and both scale with threads), inline otherwise.  fn must
// not throw.
template <class F>
inline void for_ranges(size_t n, F&& fn) {
  const size_t kMinPerThread = size_t(1) << 16;
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t parts = std::min<size_t>(std::min<size_t>(16, hw ? hw : 1), n / kMinPerThread);
  if (parts <= 1) { fn(size_t(0), n); return; }
  const size_t step = (n + parts - 1) / parts;
  std::vector<std::thread> helpers;
  helpers.reserve(parts - 1);
  size_t started = 1;                              // ranges [0, started) are taken care of: range 0 by this thread
  try {
    for (; started < parts; ++started) {
      const size_t p = started;
      helpers.emplace_back([&fn, p, step, n] { fn(std::min(n, p * step), std::min(n, (p + 1) * step)); });
    }
  } catch (...) {}                                 // no more threads to be had: the remaining ranges run here
  fn(size_t(0), std::min(n, step));
  for (size_t p = started; p < parts; ++p) fn(std::min(n, p * step), std::min(n, (p + 1) * step));
  for (auto& t : helpers) t.join();
}

}  // namespace detail

namespace Sampling {

struct UniformDistSampler {
  template <typename Point>
  inline void operator()(const std::vector<Point>& inputset, const Match4PCSOptions& options, std::vector<Point>& output) const {
    const int64_t n = int64_t(inputset.size());
    output.clear();
    if (n == 0) return;
    // uninitialised arrays (no serial zero-fill pass), first touched by the threads that fill them; `keep` receives only
    // the k indices of the points kept
    const std::unique_ptr<float[]> x(new float[size_t(n)]), y(new float[size_t(n)]), z(new float[size_t(n)]);
    detail::for_ranges(size_t(n), [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) { x[i] = inputset[i].x(); y[i] = inputset[i].y(); z[i] = inputset[i].z(); }
    });
    const std::unique_ptr<int64_t[]> keep(new int64_t[size_t(n)]);
    const int64_t k = s4p_uniform_dist_sample(x.get(), y.get(), z.get(), n, options.delta, keep.get());
    output.reserve(size_t(k));
    for (int64_t i = 0; i < k; ++i) output.push_back(inputset[size_t(keep[size_t(i)])]);
  }
};

}  // namespace Sampling
}  // namespace GlobalRegistration
#endif
