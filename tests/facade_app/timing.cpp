// Time-to-register through the DROP-IN (include/super4pcs/**, std::vector<Point3D> in and out) next to the C ABI on the same
// clouds: what the facade's AoS <-> SoA conversions cost a user of the reference API.
// usage: timing N delta sample overlap
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "s4p_matcher.h"
#include "super4pcs/algorithms/super4pcs.h"
#include "super4pcs/shared4pcs.h"
#include "super4pcs/utils/logger.h"

using namespace GlobalRegistration;
using clk = std::chrono::steady_clock;

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 1000000;
  const float delta = argc > 2 ? float(std::atof(argv[2])) : 0.004f;
  const int sample = argc > 3 ? std::atoi(argv[3]) : 2000;
  const float overlap = argc > 4 ? float(std::atof(argv[4])) : 0.5f;
  // a bumpy closed surface, two partial views (z >= -c / z <= c), Q moved rigidly + noise: the shape of configs[2]
  std::mt19937 g(20140814);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto surf = [&](float& x, float& y, float& z) {
    float a = nd(g), b = nd(g), c = nd(g);
    const float r = std::sqrt(a * a + b * b + c * c);
    a /= r; b /= r; c /= r;
    const float th = std::acos(c), ph = std::atan2(b, a);
    const float rad = 0.5f * (1.f + 0.06f * std::sin(3 * th + 0.3f) * std::sin(2 * ph + 1.f) + 0.05f * std::sin(5 * th + 2.f) * std::sin(4 * ph));
    x = a * rad; y = b * rad; z = c * rad;
  };
  const float cut = overlap / (2.f - overlap) * 0.5f;
  std::vector<Point3D> P, Q;
  P.reserve(size_t(n)); Q.reserve(size_t(n));
  while (int(P.size()) < n) { float x, y, z; surf(x, y, z); if (z >= -cut) P.emplace_back(x, y, z); }
  while (int(Q.size()) < n) {
    float x, y, z; surf(x, y, z);
    if (z > cut) continue;
    const float qx = 0.8f * x - 0.6f * y + 0.3f + delta * nd(g), qy = 0.6f * x + 0.8f * y - 0.2f + delta * nd(g), qz = z + 0.1f + delta * nd(g);
    Q.emplace_back(qx, qy, qz);
  }
  Match4PCSOptions opt;
  opt.delta = delta; opt.sample_size = size_t(sample);
  if (!opt.configureOverlap(overlap)) return 2;
  Utils::Logger logger(Utils::NoLog);
  double t_facade = 0, t_abi = 0; float s_facade = 0, s_abi = 0;
  for (int rep = 0; rep < 2; ++rep) {                       // second pass: allocator and page cache warm
    {
      MatchSuper4PCS m(opt, logger);
      Match4PCSBase::MatrixType mat = Match4PCSBase::MatrixType::Identity();
      std::vector<Point3D> Q2 = Q;
      const auto t0 = clk::now();
      s_facade = m.ComputeTransformation(P, &Q2, mat);
      t_facade = std::chrono::duration<double>(clk::now() - t0).count();
    }
    {
      std::vector<float> px(P.size()), py(P.size()), pz(P.size()), qx(Q.size()), qy(Q.size()), qz(Q.size());
      for (size_t i = 0; i < P.size(); ++i) { px[i] = P[i].x(); py[i] = P[i].y(); pz[i] = P[i].z(); }
      for (size_t i = 0; i < Q.size(); ++i) { qx[i] = Q[i].x(); qy[i] = Q[i].y(); qz[i] = Q[i].z(); }
      s4p_options o{};
      o.delta = opt.delta; o.max_normal_difference = opt.max_normal_difference; o.max_translation_distance = opt.max_translation_distance;
      o.max_angle = opt.max_angle; o.max_color_distance = opt.max_color_distance; o.sample_size = opt.sample_size;
      o.max_time_seconds = opt.max_time_seconds; o.random_seed = opt.randomSeed;
      o.terminate_threshold = opt.getTerminateThreshold(); o.overlap_estimation = opt.getOverlapEstimation();
      s4p_matcher* e = nullptr;
      if (s4p_matcher_create(&o, nullptr, 0, &e) != S4P_OK) return 3;
      const s4p_cloud_view vp{px.data(), py.data(), pz.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, int64_t(P.size())};
      const s4p_cloud_view vq{qx.data(), qy.data(), qz.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, int64_t(Q.size())};
      float M[16], lcp = 0;
      const auto t0 = clk::now();
      if (s4p_matcher_compute_transformation(e, &vp, &vq, qx.data(), qy.data(), qz.data(), M, &lcp) != S4P_OK) return 4;
      t_abi = std::chrono::duration<double>(clk::now() - t0).count();
      s_abi = lcp;
      s4p_matcher_destroy(e);
    }
  }
  std::printf("{\"n_points\": %d, \"sample\": %d, \"facade_time_to_register_s\": %.4f, \"c_abi_time_to_register_s\": %.4f, \"facade_over_abi\": %.3f, "
              "\"lcp_facade\": %.6f, \"lcp_abi\": %.6f}\n", n, sample, t_facade, t_abi, t_facade / t_abi, double(s_facade), double(s_abi));
  return s_facade == s_abi ? 0 : 5;
}
