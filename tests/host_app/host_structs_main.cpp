// Test harness (CPU, no GPU): drives the product's host-side structures (super4pcs_amd/csrc/s4p_host_structs.hpp)
// so that tests/test_host_structs.py can compare them with the oracle.
//   octree <file>   file: n, then n lines "x y z" (sampled, centred Q), then k, then k lines "distance epsilon".
//                   Replays PairOctree (split memo + distance shells) over the k calls; prints, per call,
//                   "n_seq n_leaf" followed by the permutation `ids` and the flattened sequence ids.
//   frame <file>    same file format (only the points are used): prints UnitFrame's centre and ratio as hex floats.
//   fourth <seed>   randomized check of FourthPointIndex against the literal loop of match4pcsBase.cc:324-338;
//                   prints "queries <q> mismatches <m>".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <random>
#include <vector>

#include "s4p_host_structs.hpp"

static int run_octree(const char* path) {
  std::ifstream f(path);
  size_t n = 0; f >> n;
  std::vector<float> qx(n), qy(n), qz(n), ux, uy, uz;
  for (size_t i = 0; i < n; ++i) f >> qx[i] >> qy[i] >> qz[i];
  size_t k = 0; f >> k;
  s4p::UnitFrame frame; frame.build(qx, qy, qz, ux, uy, uz);
  s4p::PairOctree tree; tree.reset(uint32_t(n));
  std::vector<uint32_t> sid(n), loff(n + 1); std::vector<s4p::Leaf> leaves(n);
  for (size_t c = 0; c < k; ++c) {
    float d = 0, eps = 0; f >> d >> eps;
    tree.build(ux.data(), uy.data(), uz.data(), uint32_t(n), d / frame.ratio, eps / frame.ratio, 50);
    tree.flatten(sid.data(), loff.data(), leaves.data());
    if (loff[tree.n_leaf()] != tree.n_seq() || loff[0] != 0) { std::printf("bad leaf offsets\n"); return 1; }
    std::printf("%u %u\n", tree.n_seq(), tree.n_leaf());
    for (size_t i = 0; i < n; ++i) std::printf("%u ", tree.ids[i]);
    std::printf("\n");
    // a sequence word is id | leaf << 16: the leaf must be the one whose slot range holds the word
    for (uint32_t l = 0; l < tree.n_leaf(); ++l)
      for (uint32_t i = loff[l]; i < loff[l + 1]; ++i) if ((sid[i] >> 16) != l) { std::printf("bad leaf tag\n"); return 1; }
    for (uint32_t i = 0; i < tree.n_seq(); ++i) std::printf("%u ", sid[i] & 0xFFFFu);
    std::printf("\n");
  }
  return 0;
}

static int run_frame(const char* path) {
  std::ifstream f(path);
  size_t n = 0; f >> n;
  std::vector<float> qx(n), qy(n), qz(n), ux, uy, uz;
  for (size_t i = 0; i < n; ++i) f >> qx[i] >> qy[i] >> qz[i];
  s4p::UnitFrame frame; frame.build(qx, qy, qz, ux, uy, uz);
  std::printf("%a %a %a %a\n", double(frame.gcenter[0]), double(frame.gcenter[1]), double(frame.gcenter[2]), double(frame.ratio));
  return 0;
}

static int literal_fourth(const std::vector<float>& X, const std::vector<float>& Y, const std::vector<float>& Z, float pa, float pb,
                          float pc, const float* A, const float* B, const float* C, float too_small) {
  int b4 = -1; float best = std::numeric_limits<float>::max();
  auto far = [&](size_t i, const float* q) { const float dx = X[i] - q[0], dy = Y[i] - q[1], dz = Z[i] - q[2]; return dx * dx + (dy * dy + dz * dz) >= too_small; };
  for (size_t i = 0; i < X.size(); ++i) {
    const float v = (pa * X[i] + pb * Y[i]) + pc * Z[i];
    const float d = float(std::fabs(double(v) - 1.0));
    if (d < best && far(i, A) && far(i, B) && far(i, C)) { best = d; b4 = int(i); }
  }
  return b4;
}

static int run_fourth(unsigned seed) {
  std::mt19937 g(seed); std::normal_distribution<float> nd(0, 1);
  long bad = 0, queries = 0;
  for (int variant = 0; variant < 4; ++variant) {
    const size_t n = variant == 2 ? 1777 : (variant == 3 ? 5 : 20011);
    std::vector<float> x(n), y(n), z(n);
    for (size_t i = 0; i < n; ++i) {
      float a = nd(g), b = nd(g), c = nd(g); float r = std::sqrt(a * a + b * b + c * c);
      r /= (1.f + 0.2f * std::sin(5 * a / r) * std::cos(3 * b / r));
      x[i] = a / r; y[i] = b / r; z[i] = c / r;
      if (variant == 1) { x[i] = std::round(x[i] * 8) / 8; y[i] = std::round(y[i] * 8) / 8; z[i] = std::round(z[i] * 8) / 8; }   // many exact ties
    }
    s4p::FourthPointIndex idx; idx.build(x.data(), y.data(), z.data(), n);
    for (int r = 0; r < 1500; ++r) {
      const size_t i1 = g() % n, i2 = g() % n, i3 = g() % n;
      const double x1 = x[i1], y1 = y[i1], z1 = z[i1], x2 = x[i2], y2 = y[i2], z2 = z[i2], x3 = x[i3], y3 = y[i3], z3 = z[i3];
      const float denom = float(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
      if (denom == 0) continue;
      const float pa = float((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
      const float pb = float((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
      const float pc = float((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
      const float A[3] = {x[i1], y[i1], z[i1]}, B[3] = {x[i2], y[i2], z[i2]}, C[3] = {x[i3], y[i3], z[i3]};
      const float ts = r % 7 == 0 ? 4.0f : (r % 3 == 0 ? 0.25f : 0.01f);      // 4.0: nothing qualifies -> -1
      const int want = literal_fourth(x, y, z, pa, pb, pc, A, B, C, ts);
      static std::vector<float> scratch;
      const int got = idx.query(pa, pb, pc, A, B, C, ts, scratch);
      bad += want != got; ++queries;
    }
  }
  std::printf("queries %ld mismatches %ld\n", queries, bad);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 3 && !std::strcmp(argv[1], "octree")) return run_octree(argv[2]);
  if (argc == 3 && !std::strcmp(argv[1], "frame")) return run_frame(argv[2]);
  if (argc == 3 && !std::strcmp(argv[1], "fourth")) return run_fourth(unsigned(std::atoi(argv[2])));
  std::fprintf(stderr, "usage: %s octree <file> | frame <file> | fourth <seed>\n", argv[0]);
  return 2;
}
