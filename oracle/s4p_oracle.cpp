// ============================================================================
// oracle/s4p_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (plain C++17, no Eigen) of the Super4PCS hot path
//   sampling -> init -> base selection -> ExtractPairs -> FindCongruentQuadrilaterals
//   -> TryCongruentSet (ComputeRigidTransformation + Verify) -> global transform
// following the reference files in the order SURVEY.md §8c lists.  Every function
// cites the reference file:line it follows (paths relative to /root/reference/).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library.  The product (super4pcs_amd/) never links, imports or calls it.
//
// PARITY STATUS: pinned against the reference's own sources, up to Eigen's internal evaluation order.
// The reference cannot be built as shipped (Eigen is an un-vendored, un-pinned submodule and the image has no
// Eigen), and it stores no golden vectors for this path.  So (a) this file is checked against everything the
// reference publishes -- sampler counts on its bundled hippo assets (doc/Usage.md:80), the RANSAC trial-count
// formula, its pair_extraction test predicate (tests/testing.h:172-194) -- and (b) the reference's UNMODIFIED
// match4pcsBase.cc / super4pcs.cc (+ headers) are compiled where they lie against a minimal Eigen stand-in
// (oracle/eigen_shim, oracle/Makefile target `ref` -> oracle/_ref/libs4p_ref.so) and this restatement must
// reproduce them bit for bit: sampled clouds, std::shuffle, base selection, ordered pair lists across calls,
// ordered quad lists, gate decisions, per-candidate LCPs, winning transform (tests/test_oracle_vs_reference.py).
// What stays unpinned is the order in which real Eigen evaluates its fixed-size expressions; the shim and this
// file both use the orders derived in DESIGN.md "Numerics contract".
//
// Floating point: compile with  g++ -O2 -ffp-contract=off -fno-fast-math  (x86-64
// SSE2, no FMA contraction), matching a plain CMake Release build of the reference.
// Eigen 3.3 fixed-size evaluation orders that matter are written out explicitly:
//   * 3-vector sum/dot/squaredNorm reduce as  x + (y + z)     (redux_novec_unroller)
//   * Matrix3f*Vector3f and 3x3*3x3 coefficients:  a0*b0 + (a1*b1 + a2*b2)
//   * Matrix4f * v.homogeneous() (Verify):  ((m0*x + m1*y) + m2*z) + m3   (packet path)
// Documented deviations from the reference (see DESIGN.md "Deviations"):
//   D1  Quaternion::setFromTwoVectors near-opposite branch (JacobiSVD in Eigen) is
//       replaced by a closed-form perpendicular axis.
//   D2  IndexedNormalSet's dense egSize^3 pointer array is a hash map with identical
//       cell/bucket membership and visit order (the final quad list is a sorted set).
//   D3  linear cell indices are 64-bit (the reference's int overflows for egSize>=1291).
// ============================================================================
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <set>
#include <unordered_map>
#include <utility>
#include <vector>

namespace s4po {

// ----------------------------------------------------------------------------
// shared4pcs.h:61-111  Point3D ;  shared4pcs.h:148-190  Match4PCSOptions
// ----------------------------------------------------------------------------
struct P3 {
  float pos[3] = {0.f, 0.f, 0.f};
  float nrm[3] = {0.f, 0.f, 0.f};
  float rgb[3] = {-1.f, -1.f, -1.f};
};

struct Options {
  float delta = 5.0f;
  float max_normal_difference = -1.f;
  float max_translation_distance = -1.f;
  float max_angle = -1.f;
  float max_color_distance = -1.f;
  uint64_t sample_size = 200;
  int max_time_seconds = 60;
  unsigned int randomSeed = 5489u;  // std::mt19937::default_seed
  float terminate_threshold = 1.0f;
  float overlap_estimation = 0.2f;
};

// Eigen fixed-size 3-vector reductions: x + (y + z).
static inline float dot3(const float* a, const float* b) {
  return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
}
static inline float sqn3(const float* a) { return dot3(a, a); }
static inline float norm3(const float* a) { return std::sqrt(sqn3(a)); }
static inline void sub3(const float* a, const float* b, float* o) {
  o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2];
}
static inline void cross3(const float* a, const float* b, float* o) {
  // Eigen cross3: (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
  float x = a[1] * b[2] - a[2] * b[1];
  float y = a[2] * b[0] - a[0] * b[2];
  float z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
// MatrixBase::normalize()/normalized() (Eigen 3.3): z = squaredNorm; if z>0 v /= sqrt(z)
static inline void normalize3(float* v) {
  float z = sqn3(v);
  if (z > 0.f) { float s = std::sqrt(z); v[0] /= s; v[1] /= s; v[2] /= s; }
}

// ----------------------------------------------------------------------------
// sampling.h:59-122  UniformDistSampler : keep the first point (input order) of
// every delta-voxel; voxel = int(floor(coord * (1.0f/delta))).  The open-addressing
// table of the reference only decides *where* a voxel is stored, not which point
// wins, so any exact voxel->first-index map gives the same output sequence.
// ----------------------------------------------------------------------------
static void uniform_dist_sample(const std::vector<P3>& in, float delta, std::vector<P3>& out) {
  const uint64_t n = in.size();
  out.clear();
  if (n == 0) return;
  const float scale = 1.0f / delta;                       // sampling.h:76
  // The reference hashes a voxel with (M1 x + M2 y + M3 z) % n and probes linearly (:70-72, :88-99).  On a scene of planes that
  // hash clusters: 10 M points of configs[4] took 200 s here against 2 s with the table below.  Where a voxel is stored does not
  // matter to the output, so the voxel -> "seen" map is an open-addressing table under a mixing hash; the point kept for a
  // voxel is, as in the reference (:115-118), the first one of the input order.
  uint64_t cap = 16;
  while (cap < 2 * n) cap <<= 1;
  struct Slot { int c[3]; int used; };
  std::vector<Slot> table(cap, Slot{{0, 0, 0}, 0});
  auto mix = [](uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; };
  for (uint64_t i = 0; i < n; ++i) {
    const P3& p = in[i];
    const int c[3] = {int(std::floor(p.pos[0] * scale)), int(std::floor(p.pos[1] * scale)),
                      int(std::floor(p.pos[2] * scale))};                   // :84-86
    uint64_t key = mix(mix(uint64_t(uint32_t(c[0])) | (uint64_t(uint32_t(c[1])) << 32)) ^ uint64_t(uint32_t(c[2]))) & (cap - 1);
    while (table[key].used && !(table[key].c[0] == c[0] && table[key].c[1] == c[1] && table[key].c[2] == c[2])) key = (key + 1) & (cap - 1);
    if (!table[key].used) {                                                  // first point of this voxel: kept
      table[key] = Slot{{c[0], c[1], c[2]}, 1};
      out.push_back(p);
    }
  }
}

// ----------------------------------------------------------------------------
// kdtree.h  KdTree<float,int>: 64 pts/leaf, depth <= 32 (:60-63), build :554-635,
// split :516-531, restricted closest query :388-453.
// ----------------------------------------------------------------------------
struct KdTree {
  struct Node { float splitValue = 0; unsigned firstChildId = 0; unsigned dim = 0; unsigned leaf = 0;
                unsigned start = 0; unsigned size = 0; };
  std::vector<std::array<float, 3>> pts;
  std::vector<int> idx;
  std::vector<Node> nodes;
  static constexpr unsigned kCell = 64, kDepth = 32;

  void build(const std::vector<P3>& P) {
    pts.clear(); idx.clear(); nodes.clear();
    pts.reserve(P.size()); idx.reserve(P.size());
    for (size_t i = 0; i < P.size(); ++i) {                 // match4pcsBase.cc:359-361, kdtree.h add()
      pts.push_back({P[i].pos[0], P[i].pos[1], P[i].pos[2]});
      idx.push_back(int(i));
    }
    nodes.reserve(4 * pts.size() / kCell + 8);
    nodes.emplace_back();
    nodes.back().leaf = 0;
    if (!pts.empty()) create(0, 0, unsigned(pts.size()), 1);
  }
  unsigned split(int start, int end, unsigned dim, float sv) {   // kdtree.h:516-531
    int l = start, r = end - 1;
    for (; l < r; ++l, --r) {
      while (l < end && pts[l][dim] < sv) l++;
      while (r >= start && pts[r][dim] >= sv) r--;
      if (l > r) break;
      std::swap(pts[l], pts[r]);
      std::swap(idx[l], idx[r]);
    }
    if (l >= int(pts.size())) return unsigned(l);   // guard the reference's unguarded read
    return (pts[l][dim] < sv ? l + 1 : l);
  }
  void create(unsigned nodeId, unsigned start, unsigned end, unsigned level) {   // kdtree.h:554-635
    float mn[3], mx[3];
    // Eigen::AlignedBox default ctor is setEmpty(): min=+max(), max=-max(); only extend() matters.
    mn[0] = mn[1] = mn[2] = std::numeric_limits<float>::max();
    mx[0] = mx[1] = mx[2] = -std::numeric_limits<float>::max();
    for (unsigned i = start; i < end; ++i)
      for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], pts[i][k]); mx[k] = std::max(mx[k], pts[i][k]); }
    float diag[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    unsigned dim = 0;                                       // maxCoeff(&dim): first maximum
    if (diag[1] > diag[dim]) dim = 1;
    if (diag[2] > diag[dim]) dim = 2;
    nodes[nodeId].dim = dim;
    nodes[nodeId].splitValue = (mn[dim] + mx[dim]) / 2.f;  // AlignedBox::center() = (min+max)/2
    unsigned midId = split(int(start), int(end), dim, nodes[nodeId].splitValue);
    unsigned fc = unsigned(nodes.size());
    nodes[nodeId].firstChildId = fc;
    nodes.emplace_back(); nodes.emplace_back();
    {
      if (midId - start <= kCell || level >= kDepth) {
        nodes[fc].leaf = 1; nodes[fc].start = start; nodes[fc].size = midId - start;
      } else { nodes[fc].leaf = 0; create(fc, start, midId, level + 1); }
    }
    {
      if (end - midId <= kCell || level >= kDepth) {
        nodes[fc + 1].leaf = 1; nodes[fc + 1].start = midId; nodes[fc + 1].size = end - midId;
      } else { nodes[fc + 1].leaf = 0; create(fc + 1, midId, end, level + 1); }
    }
  }
  // kdtree.h:388-453  returns index or -1
  int closest(const float* q, float sqdist, int currentId = -1) const {
    struct QN { unsigned nodeId; float sq; };
    QN stack[64];
    int cl_id = -1;
    float cl_dist = sqdist;
    stack[0] = {0, 0.f};
    unsigned count = 1;
    while (count) {
      QN& qn = stack[count - 1];
      const Node& node = nodes[qn.nodeId];
      if (qn.sq < cl_dist) {
        if (node.leaf) {
          --count;
          const int end = int(node.start + node.size);
          for (int i = int(node.start); i < end; ++i) {
            float d[3] = {q[0] - pts[i][0], q[1] - pts[i][1], q[2] - pts[i][2]};
            const float sq = sqn3(d);
            if (sq <= cl_dist && idx[i] != currentId) { cl_dist = sq; cl_id = idx[i]; }
          }
        } else {
          const float new_off = q[node.dim] - node.splitValue;
          if (new_off < 0.) { stack[count].nodeId = node.firstChildId; qn.nodeId = node.firstChildId + 1; }
          else { stack[count].nodeId = node.firstChildId + 1; qn.nodeId = node.firstChildId; }
          stack[count].sq = qn.sq;
          qn.sq = new_off * new_off;
          ++count;
        }
      } else {
        --count;
      }
    }
    return cl_id;
  }
};

// ----------------------------------------------------------------------------
// match4pcsBase.cc:64-131  distSegmentToSegment, instantiated with Scalar=double
// (invariants are double&, match4pcsBase.cc:238-245), VectorType=Vector3f.
// double*Vector3f promotes the scalar to float (Eigen promote_scalar_arg).
// ----------------------------------------------------------------------------
static double dist_segment_to_segment(const float* p1, const float* p2, const float* q1,
                                      const float* q2, double& invariant1, double& invariant2) {
  const double kSmallNumber = 0.0001;
  float u[3], v[3], w[3];
  sub3(p2, p1, u); sub3(q2, q1, v); sub3(p1, q1, w);
  double a = dot3(u, u), b = dot3(u, v), c = dot3(v, v), d = dot3(u, w), e = dot3(v, w);
  double f = a * c - b * b;
  double s1 = 0.0, s2 = f, t1 = 0.0, t2 = f;
  if (f < kSmallNumber) {
    s1 = 0.0; s2 = 1.0; t1 = e; t2 = c;
  } else {
    s1 = (b * e - c * d);
    t1 = (a * e - b * d);
    if (s1 < 0.0) { s1 = 0.0; t1 = e; t2 = c; }
    else if (s1 > s2) { s1 = s2; t1 = e + b; t2 = c; }
  }
  if (t1 < 0.0) {
    t1 = 0.0;
    if (-d < 0.0) s1 = 0.0;
    else if (-d > a) s1 = s2;
    else { s1 = -d; s2 = a; }
  } else if (t1 > t2) {
    t1 = t2;
    if ((-d + b) < 0.0) s1 = 0;
    else if ((-d + b) > a) s1 = s2;
    else { s1 = (-d + b); s2 = a; }
  }
  invariant1 = (std::abs(s1) < kSmallNumber ? 0.0 : s1 / s2);
  invariant2 = (std::abs(t1) < kSmallNumber ? 0.0 : t1 / t2);
  const float i1 = float(invariant1), i2 = float(invariant2);
  float r[3];
  for (int k = 0; k < 3; ++k) r[k] = (w[k] + (i1 * u[k])) - (i2 * v[k]);
  return double(norm3(r));
}

// ----------------------------------------------------------------------------
// pairExtraction/intersectionPrimitive.h:117-157  HyperSphere::intersect / intersectPoint
// ----------------------------------------------------------------------------
static inline bool sphere_box_intersect(const float* c, float radius, const float* nodeCenter, float h) {
  float dmin_t[3], dmax_t[3];
  for (int k = 0; k < 3; ++k) {
    const float mn = nodeCenter[k] - h, mx = nodeCenter[k] + h;
    const float sqmin = (c[k] - mn) * (c[k] - mn);
    const float sqmax = (c[k] - mx) * (c[k] - mx);
    dmin_t[k] = (c[k] < mn) ? sqmin : ((c[k] > mx) ? sqmax : 0.f);
    dmax_t[k] = (sqmin < sqmax) ? sqmax : sqmin;
  }
  const float dmin = dmin_t[0] + (dmin_t[1] + dmin_t[2]);
  const float dmax = dmax_t[0] + (dmax_t[1] + dmax_t[2]);
  const float r2 = radius * radius;
  return (dmin < r2 && r2 < dmax);
}
static inline bool sphere_point_intersect(const float* c, float radius, const float* pos, float eps) {
  float d[3]; sub3(pos, c, d);
  const float t = norm3(d) - radius;
  return t * t < eps * eps;
}

// pairExtraction/intersectionFunctor.h:59-67
static float rounded_epsilon(float epsilon, int* lvl) {
  const int lvlMax = int(-std::log2(epsilon));
  if (lvl) *lvl = lvlMax;
  return float(1.f / std::pow(2, lvlMax));
}

// pairExtraction/intersectionNode.h:72-245  NdNode<Point,3,float>
struct NdNode { float c[3]; unsigned begin, end; int rangeLength() const { return int(end) - int(begin); } };

struct Trace {  // one record per TryOneBase call (parity harness)
  int ok_select; int base[4]; float inv1, inv2; int m1, m2, K, C; unsigned best_count; int best_index;
};

// ----------------------------------------------------------------------------
// The matcher: Match4PCSBase (+ MatchSuper4PCS overrides)
// ----------------------------------------------------------------------------
struct Matcher {
  Options opt;
  std::mt19937 rng;
  std::vector<P3> Ps, Qs;           // sampled_P_3D_, sampled_Q_3D_ (centred)
  std::vector<P3> base3D;           // base_3D_
  float centroidP[3], centroidQ[3];
  float P_diameter = 0, max_base_diameter = -1, P_mean_distance = 1;
  int number_of_trials = 0, current_trial = 0;
  float best_LCP = 0;
  float transform[16];              // row-major 4x4 (transform_)
  float qcentroid1[3], qcentroid2[3];
  int base_ids[4], current_congruent[4];
  KdTree kd;
  // PairCreationFunctor state (pairCreationFunctor.h)
  std::vector<unsigned> ids;        // persists across calls (:36,120)
  std::vector<std::array<float, 3>> upts;   // unit-cube points
  float gcenter[3]; float ratio = 1.f;
  // statistics
  uint64_t n_verified = 0;          // visitor calls with fraction == -1
  uint64_t n_quads = 0, n_pairs = 0, n_verify_queries = 0;
  double t_pairs = 0, t_quads = 0, t_verify = 0, t_select = 0;
  double budget_seconds = 0;        // >0: TryCongruentSet stops once this much wall time was spent (bench sample)
  std::chrono::steady_clock::time_point budget_t0;
  bool budget_hit = false;
  bool full_counts = false;         // if true Verify never exits early (parity mode)
  bool use_kdtree = true;           // false: brute-force predicate
  std::vector<Trace> trace;
  bool keep_trace = false;
  // last-call buffers
  std::vector<std::pair<int, int>> last_pairs;

  explicit Matcher(const Options& o) : opt(o), rng(o.randomSeed) {
    base3D.resize(4);
    for (int i = 0; i < 16; ++i) transform[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int k = 0; k < 3; ++k) centroidP[k] = centroidQ[k] = qcentroid1[k] = qcentroid2[k] = 0.f;
    for (int k = 0; k < 4; ++k) base_ids[k] = current_congruent[k] = 0;
  }

  // ---- match4pcsBase.hpp:90-203  init --------------------------------------
  void init(const std::vector<P3>& P, const std::vector<P3>& Q) {
    const float kSmallError = 0.00001f;
    const int kMinNumberOfTrials = 4;
    const float kDiameterFraction = 0.3f;
    for (int k = 0; k < 3; ++k) centroidP[k] = centroidQ[k] = 0.f;
    Ps.clear(); Qs.clear();
    if (P.size() > opt.sample_size) uniform_dist_sample(P, opt.delta, Ps);   // :112-119
    else Ps = P;
    if (Q.size() > opt.sample_size) {                                       // :124-138
      std::vector<P3> uq;
      uniform_dist_sample(Q, opt.delta, uq);
      std::vector<uint32_t> perm(uq.size());
      for (size_t i = 0; i < perm.size(); ++i) perm[i] = uint32_t(i);
      std::shuffle(perm.begin(), perm.end(), rng);   // same swaps as shuffling the points
      size_t nb = std::min<size_t>(uq.size(), opt.sample_size);
      for (size_t i = 0; i < nb; ++i) Qs.push_back(uq[perm[i]]);
    } else Qs = Q;
    auto centre = [](std::vector<P3>& c, float* cen) {                     // :142-149
      for (const auto& p : c) { cen[0] += p.pos[0]; cen[1] += p.pos[1]; cen[2] += p.pos[2]; }
      const float n = float(c.size());
      cen[0] /= n; cen[1] /= n; cen[2] /= n;
      for (auto& p : c) { p.pos[0] -= cen[0]; p.pos[1] -= cen[1]; p.pos[2] -= cen[2]; }
    };
    centre(Ps, centroidP);
    centre(Qs, centroidQ);
    kd.build(Ps);                                                            // initKdTree, .cc:353-363
    P_diameter = 0.f;                                                        // :155-164 (on sampled Q: quirk)
    for (int i = 0; i < 1000; ++i) {
      int at = int(rng() % Qs.size());
      int bt = int(rng() % Qs.size());
      float d[3]; sub3(Qs[bt].pos, Qs[at].pos, d);
      float l = norm3(d);
      if (l > P_diameter) P_diameter = l;
    }
    // MeanDistance() (:168, match4pcsBase.cc:158-182) fills P_mean_distance_ only: nothing reads it and it draws no random
    // numbers.  It is one restricted nearest-neighbour query per sampled P point -- minutes at n_P = 4.2 M -- so the tests of the
    // 5 M / 10 M-point configs switch it off (S4PO_SKIP_MEAN_DISTANCE=1); every result is the same with and without it.
    if (!std::getenv("S4PO_SKIP_MEAN_DISTANCE")) P_mean_distance = mean_distance();
    max_base_diameter = P_diameter;                                          // :172
    // :175-185.  std::log(float) -> logf; unqualified pow(float,float) -> ::pow(double,double)
    float first_estimation =
        float(std::log(kSmallError) / std::log(1.0 - std::pow(double(opt.overlap_estimation),
                                                              double(float(kMinNumberOfTrials)))));
    number_of_trials = int(first_estimation * (P_diameter / kDiameterFraction) / max_base_diameter);
    if (number_of_trials < kMinNumberOfTrials) number_of_trials = kMinNumberOfTrials;
    current_trial = 0;
    best_LCP = 0.f;
    for (int i = 0; i < 4; ++i) { base_ids[i] = 0; current_congruent[i] = 0; }
    for (int i = 0; i < 16; ++i) transform[i] = (i % 5 == 0) ? 1.f : 0.f;
    synch3DContent();                                                        // Initialize(), super4pcs.cc:230-234
    bool fc = full_counts; full_counts = true;
    unsigned good = 0;
    best_LCP = verify(transform, &good);                                     // :201
    full_counts = fc;
  }

  // Checker-only entry: take clouds that are ALREADY sampled and centred exactly as they are (no sampling, no shuffle, no
  // re-centring -- re-centring a centred cloud would move every coordinate by a rounding error) and build the kd-tree, so
  // that Verify can recount transforms scored elsewhere on the very same points.  Not a reference code path.
  void set_sampled(const std::vector<P3>& P, const std::vector<P3>& Q) {
    for (int k = 0; k < 3; ++k) centroidP[k] = centroidQ[k] = 0.f;
    Ps = P; Qs = Q;
    kd.build(Ps);
    best_LCP = 0.f;
  }

  // ---- match4pcsBase.cc:158-182  MeanDistance (result stored, otherwise unused) ----
  float mean_distance() {
    const float kDiameterFraction = 0.2f;
    int number_of_samples = 0;
    float distance = 0.f;
    for (size_t i = 0; i < Ps.size(); ++i) {
      int res = kd.closest(Ps[i].pos, P_diameter * kDiameterFraction, int(i));  // sqdist quirk :169
      if (res != -1) {
        float d[3]; sub3(Ps[i].pos, Ps[res].pos, d);
        distance += norm3(d);
        number_of_samples++;
      }
    }
    return distance / float(number_of_samples);
  }

  // ---- pairCreationFunctor.h:90-122  synch3DContent -------------------------
  void synch3DContent() {
    upts.clear();
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<float>::max() / 2; mx[k] = -mn[k]; }  // bbox.h:71-73
    const unsigned n = unsigned(Qs.size());
    for (unsigned i = 0; i < n; ++i) {
      upts.push_back({Qs[i].pos[0], Qs[i].pos[1], Qs[i].pos[2]});
      for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], Qs[i].pos[k]); mx[k] = std::max(mx[k], Qs[i].pos[k]); }
    }
    for (int k = 0; k < 3; ++k) gcenter[k] = (mn[k] + mx[k]) / 2.f;     // AlignedBox::center
    float dg[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    float mc = dg[0]; if (dg[1] > mc) mc = dg[1]; if (dg[2] > mc) mc = dg[2];
    ratio = float(double(mc) + 0.001);                                    // :111
    for (unsigned i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) upts[i][k] = (upts[i][k] - gcenter[k]) / ratio + 0.5f;   // worldToUnit :65-69
      ids.push_back(i);                                                    // :120 (never cleared)
    }
  }

  // ---- match4pcsBase.cc:185-218  SelectRandomTriangle ----------------------
  bool select_random_triangle(int& base1, int& base2, int& base3) {
    int number_of_points = int(Ps.size());
    base1 = base2 = base3 = -1;
    int first_point = int(rng() % (unsigned long)number_of_points);
    const float sq_max = max_base_diameter * max_base_diameter;
    float best_wide = 0.f;
    for (int i = 0; i < 1000; ++i) {
      const int second_point = int(rng() % (unsigned long)number_of_points);
      const int third_point = int(rng() % (unsigned long)number_of_points);
      float u[3], w[3], cr[3];
      sub3(Ps[second_point].pos, Ps[first_point].pos, u);
      sub3(Ps[third_point].pos, Ps[first_point].pos, w);
      cross3(u, w, cr);
      float how_wide = norm3(cr);
      if (how_wide > best_wide && sqn3(u) < sq_max && sqn3(w) < sq_max) {
        best_wide = how_wide; base1 = first_point; base2 = second_point; base3 = third_point;
      }
    }
    return base1 != -1 && base2 != -1 && base3 != -1;
  }

  // ---- match4pcsBase.cc:225-274  TryQuadrilateral ---------------------------
  bool try_quadrilateral(float& invariant1, float& invariant2, int& id1, int& id2, int& id3, int& id4) {
    float min_distance = std::numeric_limits<float>::max();
    int best1 = -1, best2 = -1, best3 = -1, best4 = -1;
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 4; ++j) {
        if (i == j) continue;
        int k = 0; while (k == i || k == j) k++;
        int l = 0; while (l == i || l == j || l == k) l++;
        double li1, li2;
        float segment_distance = float(dist_segment_to_segment(base3D[i].pos, base3D[j].pos,
                                                               base3D[k].pos, base3D[l].pos, li1, li2));
        if (segment_distance < min_distance) {
          min_distance = segment_distance;
          best1 = i; best2 = j; best3 = k; best4 = l;
          invariant1 = float(li1); invariant2 = float(li2);
        }
      }
    }
    if (best1 < 0 || best2 < 0 || best3 < 0 || best4 < 0) return false;
    std::vector<P3> tmp = base3D;
    base3D[0] = tmp[best1]; base3D[1] = tmp[best2]; base3D[2] = tmp[best3]; base3D[3] = tmp[best4];
    std::array<int, 4> tmpId = {id1, id2, id3, id4};
    id1 = tmpId[best1]; id2 = tmpId[best2]; id3 = tmpId[best3]; id4 = tmpId[best4];
    return true;
  }

  // ---- match4pcsBase.cc:279-351  SelectQuadrilateral ------------------------
  bool select_quadrilateral(float& invariant1, float& invariant2, int& base1, int& base2, int& base3, int& base4) {
    const float kBaseTooSmall = 0.2f;
    int current_trial_ = 0;
    while (current_trial_ < 1000) {
      if (!select_random_triangle(base1, base2, base3)) return false;
      base3D[0] = Ps[base1]; base3D[1] = Ps[base2]; base3D[2] = Ps[base3];
      const double x1 = base3D[0].pos[0], y1 = base3D[0].pos[1], z1 = base3D[0].pos[2];
      const double x2 = base3D[1].pos[0], y2 = base3D[1].pos[1], z2 = base3D[1].pos[2];
      const double x3 = base3D[2].pos[0], y3 = base3D[2].pos[1], z3 = base3D[2].pos[2];
      float denom = float(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
      if (denom != 0) {
        float A = float((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
        float B = float((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
        float C = float((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
        base4 = -1;
        float best_distance = std::numeric_limits<float>::max();
        const float too_small = float(std::pow(double(max_base_diameter * kBaseTooSmall), 2));   // std::pow(float,int) -> double
        for (unsigned i = 0; i < Ps.size(); ++i) {
          float d1[3], d2[3], d3[3];
          sub3(Ps[i].pos, Ps[base1].pos, d1); sub3(Ps[i].pos, Ps[base2].pos, d2); sub3(Ps[i].pos, Ps[base3].pos, d3);
          if (sqn3(d1) >= too_small && sqn3(d2) >= too_small && sqn3(d3) >= too_small) {
            const float distance =
                float(std::abs(double(A * Ps[i].pos[0] + B * Ps[i].pos[1] + C * Ps[i].pos[2]) - 1.0));
            if (distance < best_distance) { best_distance = distance; base4 = int(i); }
          }
        }
        if (base4 != -1) {
          base3D[3] = Ps[base4];
          if (try_quadrilateral(invariant1, invariant2, base1, base2, base3, base4)) return true;
        }
      }
      current_trial_++;
    }
    return false;
  }

  // ---- super4pcs.cc:183-224 ExtractPairs + intersectionFunctor.h:100-234 + pairCreationFunctor.h:151-218
  struct PairCtx {
    double pair_distance, pair_distance_epsilon, pair_normals_angle;
    float segment1[3]; int bp1, bp2;
  };
  void functor_process(const PairCtx& c, int i, int j, std::vector<std::pair<int, int>>& pairs) const {
    if (i > j) {
      const P3& p = Qs[j];
      const P3& q = Qs[i];
      float dv[3]; sub3(q.pos, p.pos, dv);
      const float distance = norm3(dv);
      if (std::abs(double(distance) - c.pair_distance) > c.pair_distance_epsilon) return;   // :162
      if (opt.max_normal_difference > 0 && sqn3(q.nrm) > 0 && sqn3(p.nrm) > 0) {           // :166-180
        const float norm_threshold = float(0.5 * double(opt.max_normal_difference) * M_PI / 180.0);
        float a[3], b[3];
        for (int k = 0; k < 3; ++k) { a[k] = q.nrm[k] - p.nrm[k]; b[k] = q.nrm[k] + p.nrm[k]; }
        const double first_normal_angle = norm3(a);
        const double second_normal_angle = norm3(b);
        const float first_norm_distance = float(std::min(std::abs(first_normal_angle - c.pair_normals_angle),
                                                         std::abs(second_normal_angle - c.pair_normals_angle)));
        if (first_norm_distance > norm_threshold) return;
      }
      if (opt.max_color_distance > 0) {                                                      // :182-192
        const bool use_rgb = (p.rgb[0] >= 0 && q.rgb[0] >= 0 && base3D[c.bp1].rgb[0] >= 0 && base3D[c.bp2].rgb[0] >= 0);
        float a[3], b[3];
        sub3(p.rgb, base3D[c.bp1].rgb, a); sub3(q.rgb, base3D[c.bp2].rgb, b);
        bool color_good = norm3(a) < opt.max_color_distance && norm3(b) < opt.max_color_distance;
        if (use_rgb && !color_good) return;
      }
      if (opt.max_translation_distance > 0) {                                                // :194-200
        float a[3], b[3];
        sub3(p.pos, base3D[c.bp1].pos, a); sub3(q.pos, base3D[c.bp2].pos, b);
        const bool dist_good = norm3(a) < opt.max_translation_distance && norm3(b) < opt.max_translation_distance;
        if (!dist_good) return;
      }
      if (opt.max_angle > 0) {                                                               // :203-212
        float s2[3]; sub3(q.pos, p.pos, s2); normalize3(s2);
        float ns2[3] = {-s2[0], -s2[1], -s2[2]};
        if (std::acos(dot3(c.segment1, s2)) <= double(opt.max_angle) * M_PI / 180.0) pairs.emplace_back(j, i);
        if (std::acos(dot3(c.segment1, ns2)) <= double(opt.max_angle) * M_PI / 180.0) pairs.emplace_back(i, j);
      } else {
        pairs.emplace_back(j, i);
        pairs.emplace_back(i, j);
      }
    }
  }

  unsigned node_split(unsigned start_, unsigned end_, unsigned dim, float sv) {   // intersectionNode.h:156-176
    int start = int(start_), end = int(end_);
    int l = start, r = end - 1;
    for (; l < r; ++l, --r) {
      while (l < end && upts[ids[l]][dim] < sv) l++;
      while (r >= start && upts[ids[r]][dim] >= sv) r--;
      if (l > r) break;
      std::swap(ids[l], ids[r]);
    }
    if (l >= end) return unsigned(end);
    return upts[ids[l]][dim] < sv ? unsigned(l + 1) : unsigned(l);
  }
  void node_split8(const NdNode& self, std::vector<NdNode>& childs, float rootEdgeHalfLength) {   // :187-245
    const int nbNode = 8;
    const int offset = int(childs.size());
    childs.resize(offset + nbNode, self);
    for (unsigned d = 0; d < 3; d++) {
      const unsigned nbInterval = 1u << (d + 1);
      const unsigned nbSplit = nbInterval / 2;
      const unsigned intervalNode = nbNode / nbSplit;
      const unsigned midNode = nbNode / nbInterval;
      for (unsigned s = 0; s != nbSplit; s++) {
        const unsigned beginNodeId = s * intervalNode + offset;
        const unsigned endNodeId = (s + 1) * intervalNode + offset;
        float currentCenterD = childs[beginNodeId].c[d];
        const unsigned splitId = node_split(childs[beginNodeId].begin, childs[endNodeId - 1].end, d, currentCenterD);
        const float beforeCenterD = currentCenterD - rootEdgeHalfLength / 2.f;
        const float afterCenterD = currentCenterD + rootEdgeHalfLength / 2.f;
        for (unsigned i = beginNodeId; i != beginNodeId + midNode; i++) { childs[i].c[d] = beforeCenterD; childs[i].end = splitId; }
        for (unsigned i = beginNodeId + midNode; i != endNodeId; i++) { childs[i].c[d] = afterCenterD; childs[i].begin = splitId; }
      }
    }
    childs.erase(std::remove_if(childs.begin(), childs.end(), [](const NdNode& c) { return c.rangeLength() == 0; }),
                 childs.end());
  }

  void extract_pairs(float pair_distance, float pair_normals_angle, float pair_distance_epsilon,
                     int base_point1, int base_point2, std::vector<std::pair<int, int>>& pairs) {
    pairs.clear();
    PairCtx c;
    c.pair_distance = pair_distance; c.pair_distance_epsilon = pair_distance_epsilon;
    c.pair_normals_angle = pair_normals_angle;
    c.bp1 = base_point1; c.bp2 = base_point2;
    const unsigned n = unsigned(upts.size());
    const float nRadius = pair_distance / ratio;                               // setRadius :124-129
    sub3(base3D[base_point2].pos, base3D[base_point1].pos, c.segment1);       // setBase :135-143
    normalize3(c.segment1);
    float epsilon = pair_distance_epsilon / ratio;                            // getNormalizedEpsilon :131-133
    // --- IntersectionFunctor::process ---
    int lvlMax = 0;
    epsilon = rounded_epsilon(epsilon, &lvlMax);                              // :126
    int clvl = 0;
    std::vector<NdNode> ping, pong;
    std::vector<NdNode>* nodes = &ping; std::vector<NdNode>* childNodes = &pong;
    std::vector<std::pair<NdNode, float>> earlyNodes;
    if (ids.size() != n) { ids.clear(); for (unsigned i = 0; i < n; i++) ids.push_back(i); }   // :139-144
    childNodes->push_back(NdNode{{0.5f, 0.5f, 0.5f}, 0u, unsigned(ids.size())});              // :148
    float edgeLength = 0.f, edgeHalfLength = 0.f;
    while (clvl != lvlMax - 1) {                                               // :154-191
      if (childNodes->empty()) break;
      edgeLength = float(1.f / std::pow(2, clvl));
      edgeHalfLength = edgeLength / 2.f;
      std::swap(nodes, childNodes);
      childNodes->clear();
      for (auto& nd : *nodes) {
        for (unsigned p = 0; p < n; ++p) {
          if (sphere_box_intersect(upts[p].data(), nRadius, nd.c, edgeHalfLength + epsilon)) {
            if (nd.rangeLength() > 50) node_split8(nd, *childNodes, edgeHalfLength);
            else earlyNodes.emplace_back(nd, edgeHalfLength + epsilon);
            break;
          }
        }
      }
      clvl++;
    }
    for (unsigned pId = 0; pId < n; ++pId) {                                   // :197-233
      const float* pc = upts[pId].data();
      for (const auto& nd : *childNodes) {
        if (sphere_box_intersect(pc, nRadius, nd.c, epsilon * 2.f)) {
          for (unsigned j = 0; j != unsigned(nd.rangeLength()); j++) {
            const unsigned id = ids[j + nd.begin];
            if (pId > id)
              if (sphere_point_intersect(pc, nRadius, upts[id].data(), epsilon)) functor_process(c, int(pId), int(id), pairs);
          }
        }
      }
      for (const auto& en : earlyNodes) {
        if (sphere_box_intersect(pc, nRadius, en.first.c, en.second)) {
          for (unsigned j = 0; j != unsigned(en.first.rangeLength()); j++) {
            const unsigned id = ids[j + en.first.begin];
            if (pId > id)
              if (sphere_point_intersect(pc, nRadius, upts[id].data(), epsilon)) functor_process(c, int(pId), int(id), pairs);
          }
        }
      }
    }
  }

  // ---- normalset.hpp:162-210 helpers ------------------------------------------
  static inline int index_normal(const float* n, float neps) {     // coordinatesNormal + UnrollIndexLoop
    int c0 = int((n[0] / 2.f + 0.5f) / neps);
    int c1 = int((n[1] / 2.f + 0.5f) / neps);
    int c2 = int((n[2] / 2.f + 0.5f) / neps);
    return c2 * 49 + c1 * 7 + c0;
  }
  // Eigen Quaternion::setFromTwoVectors(zhat, n) + deviation D1; q = (w, x, y, z)
  static inline void quat_from_z_to(const float* n, float* q) {
    float v1[3] = {n[0], n[1], n[2]};
    normalize3(v1);
    float c = 0.f * v1[0] + (0.f * v1[1] + 1.f * v1[2]);
    const float z0[3] = {0.f, 0.f, 1.f};
    if (c < -1.f + 1e-5f) {
      c = std::max(c, -1.f);
      float axis[3]; cross3(z0, v1, axis);
      float s = sqn3(axis);
      if (s > 0.f) { float r = std::sqrt(s); axis[0] /= r; axis[1] /= r; axis[2] /= r; }
      else { axis[0] = 1.f; axis[1] = 0.f; axis[2] = 0.f; }
      float w2 = (1.f + c) * 0.5f;
      q[0] = std::sqrt(w2);
      float sv = std::sqrt(1.f - w2);
      q[1] = axis[0] * sv; q[2] = axis[1] * sv; q[3] = axis[2] * sv;
      return;
    }
    float axis[3]; cross3(z0, v1, axis);
    float s = std::sqrt((1.f + c) * 2.f);
    float invs = 1.f / s;
    q[1] = axis[0] * invs; q[2] = axis[1] * invs; q[3] = axis[2] * invs;
    q[0] = s * 0.5f;
  }
  static inline void quat_rotate(const float* q, const float* v, float* o) {   // QuaternionBase::_transformVector
    float uv[3]; cross3(q + 1, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    float cr[3]; cross3(q + 1, uv, cr);
    for (int k = 0; k < 3; ++k) o[k] = (v[k] + q[0] * uv[k]) + cr[k];
  }

  // ---- super4pcs.cc:80-177  FindCongruentQuadrilaterals ----------------------
  bool find_congruent(float invariant1, float invariant2, float /*dt1*/, float distance_threshold2,
                      const std::vector<std::pair<int, int>>& P_pairs, const std::vector<std::pair<int, int>>& Q_pairs,
                      std::vector<std::array<int, 4>>& quads) {
    quads.clear();
    float s01[3], s23[3];
    sub3(base3D[1].pos, base3D[0].pos, s01); normalize3(s01);
    sub3(base3D[3].pos, base3D[2].pos, s23); normalize3(s23);
    const float alpha = dot3(s01, s23);                                         // :109-111 (it is cos(alpha))
    const float eps = distance_threshold2 / ratio;                              // :114
    // IndexedNormalSet ctor, normalset.h:114-124
    const float nepsilon = float(double(1.f / 7.f) + 0.00001);
    const int gridDepth = int(-std::log2(eps));
    const int egSize = int(std::pow(2, gridDepth));
    const float gepsilon = 1.f / float(egSize);
    std::unordered_map<uint64_t, std::vector<unsigned>> grid;   // key = cell*343 + bucket  (D2)
    std::unordered_map<uint64_t, char> cell_exists;
    auto index_pos = [&](const float* p) -> int64_t {
      int c0 = int(p[0] / gepsilon), c1 = int(p[1] / gepsilon), c2 = int(p[2] / gepsilon);
      return int64_t(c2) * egSize * egSize + int64_t(c1) * egSize + int64_t(c0);   // D3
    };
    for (size_t i = 0; i < P_pairs.size(); ++i) {                                 // :118-124
      const float* p1 = upts[P_pairs[i].first].data();
      const float* p2 = upts[P_pairs[i].second].data();
      float nrm[3]; sub3(p2, p1, nrm); normalize3(nrm);
      float pos[3]; for (int k = 0; k < 3; ++k) pos[k] = p1[k] + invariant1 * (p2[k] - p1[k]);
      const int64_t pId = index_pos(pos);
      const int nId = index_normal(nrm, nepsilon);
      grid[uint64_t(pId) * 343u + unsigned(nId)].push_back(unsigned(i));
      cell_exists[uint64_t(pId)] = 1;
    }
    std::set<std::pair<unsigned, unsigned>> comb;
    std::vector<unsigned> nei;
    // getNeighbors constants (normalset.hpp:174-179) depend only on cos(alpha)
    const float cosAlpha = alpha;
    const float ang = std::acos(cosAlpha);
    const float perimeter = float(double(2.f) * M_PI * double(std::atan(ang)));
    const unsigned nbSample = unsigned(2 * std::ceil(perimeter * 7.f / 2.f));
    const float angleStep = float(double(2.f) * M_PI / double(float(nbSample)));
    const float sinAlpha = std::sin(ang);
    for (unsigned i = 0; i < Q_pairs.size(); ++i) {                               // :132-164
      const float* p1 = upts[Q_pairs[i].first].data();
      const float* p2 = upts[Q_pairs[i].second].data();
      const float* pq1 = Qs[Q_pairs[i].first].pos;
      const float* pq2 = Qs[Q_pairs[i].second].pos;
      nei.clear();
      float query[3], queryQ[3], queryn[3];
      for (int k = 0; k < 3; ++k) query[k] = p1[k] + invariant2 * (p2[k] - p1[k]);
      for (int k = 0; k < 3; ++k) queryQ[k] = pq1[k] + invariant2 * (pq2[k] - pq1[k]);
      sub3(p2, p1, queryn); normalize3(queryn);
      // getNeighbors(query, queryn, alpha, nei)  normalset.hpp:162-210
      const int64_t cell = index_pos(query);
      if (cell_exists.find(uint64_t(cell)) != cell_exists.end()) {
        float q[4]; quat_from_z_to(queryn, q);
        std::set<unsigned> colored;
        for (unsigned a = 0; a != nbSample; a++) {
          float theta = float(a) * angleStep;
          float v[3] = {sinAlpha * std::cos(theta), sinAlpha * std::sin(theta), cosAlpha};
          float dir[3]; quat_rotate(q, v, dir); normalize3(dir);
          int id = index_normal(dir, nepsilon);
          auto it = grid.find(uint64_t(cell) * 343u + unsigned(id));
          if (it != grid.end() && !it->second.empty()) colored.insert(unsigned(id));
        }
        for (unsigned b : colored) {
          const auto& l = grid[uint64_t(cell) * 343u + b];
          nei.insert(nei.end(), l.begin(), l.end());
        }
      }
      for (unsigned k = 0; k != nei.size(); k++) {                               // :151-163
        const int id = int(nei[k]);
        const float* pp1 = Qs[P_pairs[id].first].pos;
        const float* pp2 = Qs[P_pairs[id].second].pos;
        float d[3];
        for (int t = 0; t < 3; ++t) { float inv = pp1[t] + (pp2[t] - pp1[t]) * invariant1; d[t] = queryQ[t] - inv; }
        if (sqn3(d) <= distance_threshold2) comb.emplace(unsigned(id), i);       // squared vs unsquared: quirk :160
      }
    }
    for (const auto& pr : comb)                                                    // :166-174
      quads.push_back({P_pairs[pr.first].first, P_pairs[pr.first].second, Q_pairs[pr.second].first, Q_pairs[pr.second].second});
    return !quads.empty();
  }

  // ---- the same enumeration, streaming: COUNTS and order-independent CHECKSUMS instead of the sorted list -------------
  // For sizes where FindCongruentQuadrilaterals' std::set cannot be built (a base of a 20 000-point sample has ~10^9
  // congruent quads): per set-2 pair the same invariant point, cell, cone rasterisation (normalset.hpp:162-210) and
  // distance test (super4pcs.cc:151-163) as find_congruent above -- checked against it on small cases
  // (tests/test_oracle.py) -- with the IndexedNormalSet held as a sorted (cell * 343 + bucket, pair) array and the loop
  // over set 2 under OpenMP.  Every quad found also goes through ComputeRigidTransformation + the rms gate of
  // TryCongruentSet (match4pcsBase.hpp:436-439) when a base is given.
  //   out[0] = #quads, out[1] = sum of quad_mix over them, out[2] = #quads that pass the gate, out[3] = their quad_mix sum
  // Gated quads whose quad_mix % sample_mod == 0 are returned (lexicographically sorted) as a deterministic subsample.
  static inline uint64_t quad_mix(int a, int b, int c, int d) {      // == s4p_quad_mix (include/s4p_capi.h)
    uint64_t x = (uint64_t(uint32_t(a)) << 32) | uint32_t(b);
    uint64_t y = (uint64_t(uint32_t(c)) << 32) | uint32_t(d);
    x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29;
    y *= 0xC2B2AE3D27D4EB4Full; y ^= y >> 31;
    const uint64_t h = (x + y) * 0xD6E8FEB86659FD93ull;
    return h ^ (h >> 32);
  }
  void count_congruent(float invariant1, float invariant2, float distance_threshold2,
                       const std::vector<std::pair<int, int>>& P_pairs, const std::vector<std::pair<int, int>>& Q_pairs,
                       const int* base /* 4 sampled-P ids or nullptr */, int threads, uint64_t out[4],
                       uint64_t sample_mod, std::vector<std::array<int, 4>>* sample,
                       uint64_t* best7 = nullptr /* {found, max inlier count, id (set-1 pair), i (set-2 pair), then the quad is id/i's} */) const {
    float s01[3], s23[3];
    sub3(base3D[1].pos, base3D[0].pos, s01); normalize3(s01);
    sub3(base3D[3].pos, base3D[2].pos, s23); normalize3(s23);
    const float cosAlpha = dot3(s01, s23);                                      // :109-111
    const float eps = distance_threshold2 / ratio;                              // :114
    const float nepsilon = float(double(1.f / 7.f) + 0.00001);
    const int gridDepth = int(-std::log2(eps));
    const int egSize = int(std::pow(2, gridDepth));
    const float gepsilon = 1.f / float(egSize);
    auto index_pos = [&](const float* p) -> int64_t {
      int c0 = int(p[0] / gepsilon), c1 = int(p[1] / gepsilon), c2 = int(p[2] / gepsilon);
      return int64_t(c2) * egSize * egSize + int64_t(c1) * egSize + int64_t(c0);
    };
    const size_t m1 = P_pairs.size();
    std::vector<std::pair<uint64_t, unsigned>> keyed(m1);                         // (cell * 343 + bucket, set-1 pair)
#pragma omp parallel for schedule(static) num_threads(threads)
    for (long long i = 0; i < (long long)m1; ++i) {                             // :118-124
      const float* p1 = upts[P_pairs[i].first].data();
      const float* p2 = upts[P_pairs[i].second].data();
      float nrm[3]; sub3(p2, p1, nrm); normalize3(nrm);
      float pos[3]; for (int k = 0; k < 3; ++k) pos[k] = p1[k] + invariant1 * (p2[k] - p1[k]);
      keyed[size_t(i)] = {uint64_t(index_pos(pos)) * 343u + unsigned(index_normal(nrm, nepsilon)), unsigned(i)};
    }
    std::sort(keyed.begin(), keyed.end());
    const float ang = std::acos(cosAlpha);                                       // normalset.hpp:174-179
    const float perimeter = float(double(2.f) * M_PI * double(std::atan(ang)));
    const unsigned nbSample = unsigned(2 * std::ceil(perimeter * 7.f / 2.f));
    const float angleStep = float(double(2.f) * M_PI / double(float(nbSample)));
    const float sinAlpha = std::sin(ang);
    P3 cbase[4]; float centroid1[3] = {0, 0, 0};
    if (base) {
      for (int k = 0; k < 4; ++k) cbase[k] = Ps[base[k]];
      for (int k = 0; k < 3; ++k) centroid1[k] = ((cbase[0].pos[k] + cbase[1].pos[k]) + cbase[2].pos[k]) / 3.f;
    }
    const double pi = std::acos(-1);
    const float max_angle_rad = float(double(opt.max_angle) * pi / 180.0);
    uint64_t K = 0, ksum = 0, C = 0, csum = 0;
    std::vector<std::array<int, 4>> picked;
    // (best7) the winner TryCongruentSet would keep: every gated candidate verified in full, greatest inlier count, then the
    // smallest (id, i) = the first such candidate in the std::set order of super4pcs.cc:127,166 (match4pcsBase.hpp:467-484)
    bool w_found = false; unsigned w_count = 0; uint64_t w_key = ~0ull;
    auto lower = [&](uint64_t key) { return std::lower_bound(keyed.begin(), keyed.end(), std::make_pair(key, 0u)); };
#pragma omp parallel num_threads(threads) reduction(+ : K, ksum, C, csum)
    {
      std::vector<std::array<int, 4>> mine;
      bool t_found = false; unsigned t_count = 0; uint64_t t_key = ~0ull;
#pragma omp for schedule(dynamic, 256)
      for (long long i = 0; i < (long long)Q_pairs.size(); ++i) {                  // :132-164
        const float* p1 = upts[Q_pairs[i].first].data();
        const float* p2 = upts[Q_pairs[i].second].data();
        const float* pq1 = Qs[Q_pairs[i].first].pos;
        const float* pq2 = Qs[Q_pairs[i].second].pos;
        float query[3], queryQ[3], queryn[3];
        for (int k = 0; k < 3; ++k) query[k] = p1[k] + invariant2 * (p2[k] - p1[k]);
        for (int k = 0; k < 3; ++k) queryQ[k] = pq1[k] + invariant2 * (pq2[k] - pq1[k]);
        sub3(p2, p1, queryn); normalize3(queryn);
        const uint64_t cell = uint64_t(index_pos(query));
        auto c_lo = lower(cell * 343u), c_hi = lower(cell * 343u + 343u);
        if (c_lo == c_hi) continue;                                               // no set-1 pair in this cell
        float q[4]; quat_from_z_to(queryn, q);
        bool colored[343] = {false};
        for (unsigned a = 0; a != nbSample; a++) {
          float theta = float(a) * angleStep;
          float v[3] = {sinAlpha * std::cos(theta), sinAlpha * std::sin(theta), cosAlpha};
          float dir[3]; quat_rotate(q, v, dir); normalize3(dir);
          const int id = index_normal(dir, nepsilon);
          if (id >= 0 && id < 343) colored[id] = true;
        }
        for (auto it = c_lo; it != c_hi; ++it) {
          if (!colored[it->first - cell * 343u]) continue;
          const int id = int(it->second);
          const float* pp1 = Qs[P_pairs[id].first].pos;
          const float* pp2 = Qs[P_pairs[id].second].pos;
          float d[3];
          for (int t = 0; t < 3; ++t) { float inv = pp1[t] + (pp2[t] - pp1[t]) * invariant1; d[t] = queryQ[t] - inv; }
          if (!(sqn3(d) <= distance_threshold2)) continue;                          // quirk :160
          const int a = P_pairs[id].first, b = P_pairs[id].second, cq = Q_pairs[i].first, dq = Q_pairs[i].second;
          const uint64_t mix = quad_mix(a, b, cq, dq);
          ++K; ksum += mix;
          if (!base) continue;
          const P3 cc[4] = {Qs[a], Qs[b], Qs[cq], Qs[dq]};
          float centroid2[3];
          for (int k = 0; k < 3; ++k) centroid2[k] = ((cc[0].pos[k] + cc[1].pos[k]) + cc[2].pos[k]) / 3.f;
          float rms = -1; float T[16];
          const bool ok = compute_rigid(cbase, cc, centroid1, centroid2, max_angle_rad, T, rms);
          if (ok && rms >= 0.f && rms < 2.0f * opt.delta) {
            ++C; csum += mix;
            if (sample && sample_mod && mix % sample_mod == 0) mine.push_back({a, b, cq, dq});
            if (best7) {
              unsigned good = 0; uint64_t nq_ = 0;
              const float* Tc = T;
              // full count (no early exit): verify_against with a best of 0 never leaves early
              (void)verify_against(Tc, 0.f, &good, &nq_);
              const uint64_t key = (uint64_t(unsigned(id)) << 32) | uint64_t(unsigned(i));
              if (!t_found || good > t_count || (good == t_count && key < t_key)) { t_found = true; t_count = good; t_key = key; }
            }
          }
        }
      }
#pragma omp critical
      {
        picked.insert(picked.end(), mine.begin(), mine.end());
        if (t_found && (!w_found || t_count > w_count || (t_count == w_count && t_key < w_key))) { w_found = true; w_count = t_count; w_key = t_key; }
      }
    }
    if (best7) { best7[0] = w_found ? 1 : 0; best7[1] = w_count; best7[2] = w_key >> 32; best7[3] = w_key & 0xFFFFFFFFull; }
    out[0] = K; out[1] = ksum; out[2] = C; out[3] = csum;
    if (sample) { std::sort(picked.begin(), picked.end()); *sample = std::move(picked); }
  }

  // ---- match4pcsBase.cc:365-500  ComputeRigidTransformation (computeScale=false) ----
  // returns: 0 = false, 1 = true ; rms output. T row-major.
  bool compute_rigid(const P3* ref, const P3* cand, const float* centroid1, const float* centroid2,
                     float max_angle, float* T, float& rms) const {
    rms = 1e9f;
    const float kSmallNumber = 1e-6f;
    const float* p0 = ref[0].pos; const float* p1 = ref[1].pos; const float* p2 = ref[2].pos;
    const float* q0 = cand[0].pos; const float* q1 = cand[1].pos; const float* q2 = cand[2].pos;
    auto frame = [](const float* a0, const float* a1, const float* a2, float* e1, float* e2, float* e3) -> bool {
      sub3(a1, a0, e1);
      if (sqn3(e1) == 0) return false;
      normalize3(e1);
      float t[3]; sub3(a2, a0, t);
      const float dd = dot3(t, e1);
      for (int k = 0; k < 3; ++k) e2[k] = t[k] - dd * e1[k];
      if (sqn3(e2) == 0) return false;
      normalize3(e2);
      cross3(e1, e2, e3);
      if (sqn3(e3) == 0) return false;
      normalize3(e3);
      return true;
    };
    float vp[3][3], vq[3][3];
    if (!frame(p0, p1, p2, vp[0], vp[1], vp[2])) return true;   // "return kLargeNumber" == true, rms=1e9 (quirk :417-433)
    if (!frame(q0, q1, q2, vq[0], vq[1], vq[2])) return true;
    float R[3][3];   // rotation = rotate_p^T * rotate_q ; rotate_p.row(i) = vp[i]
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[r][c] = vp[0][r] * vq[0][c] + (vp[1][r] * vq[1][c] + vp[2][r] * vq[2][c]);
    for (int i = 0; i < 3; ++i) {                                   // (R*R).diagonal() - 1 > 1e-6  :453
      float dg = R[i][0] * R[0][i] + (R[i][1] * R[1][i] + R[i][2] * R[2][i]);
      if (dg - 1.f > kSmallNumber) return false;
    }
    if (max_angle >= 0) {                                            // :457-472
      // atan2(float,float) -> atan2f; the middle term mixes float/double exactly as written in the reference
      if (!(std::abs(std::atan2(R[2][1], R[2][2])) <= max_angle &&
            std::abs(std::atan2(double(-R[2][0]),
                                std::sqrt(std::pow(double(R[2][1]), 2) + std::pow(double(R[2][2]), 2)))) <= double(max_angle) &&
            std::abs(::atan2(double(R[1][0]), double(R[0][0]))) <= double(max_angle)))
        return false;
    }
    rms = 0.f;
    for (int i = 0; i < 3; ++i) {                                    // :477-489
      float first[3], tr[3], df[3];
      for (int k = 0; k < 3; ++k) first[k] = 1.f * cand[i].pos[k] - centroid2[k];
      for (int r = 0; r < 3; ++r) tr[r] = R[r][0] * first[0] + (R[r][1] * first[1] + R[r][2] * first[2]);
      for (int k = 0; k < 3; ++k) df[k] = (tr[k] - ref[i].pos[k]) + centroid1[k];
      rms += norm3(df);
    }
    rms /= 4.f;
    // Transform chain :491-497 -> [R | c1 + R*(-c2)]
    for (int r = 0; r < 3; ++r) {
      const float rc = R[r][0] * (-centroid2[0]) + (R[r][1] * (-centroid2[1]) + R[r][2] * (-centroid2[2]));
      T[r * 4 + 0] = R[r][0]; T[r * 4 + 1] = R[r][1]; T[r * 4 + 2] = R[r][2];
      T[r * 4 + 3] = centroid1[r] + rc;
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    return true;
  }

  // ---- match4pcsBase.cc:508-567  Verify ---------------------------------------
  float verify(const float* T, unsigned* good_out = nullptr) {
    uint64_t q = 0;
    const float r = verify_against(T, best_LCP, good_out, &q);
    n_verify_queries += q;
    return r;
  }
  // The loop of Verify with the running best passed in (const: callable from the OpenMP candidate loop of baseline B)
  float verify_against(const float* T, const float best_so_far, unsigned* good_out, uint64_t* queries) const {
    const float epsilon = opt.delta;
    unsigned good_points = 0;
    const size_t number_of_points = Qs.size();
    const size_t terminate_value = size_t(best_so_far * float(number_of_points));
    const float sq_eps = epsilon * epsilon;
    for (size_t i = 0; i < number_of_points; ++i) {
      const float* q = Qs[i].pos;
      float t[3];
      for (int r = 0; r < 3; ++r)   // (mat * q.homogeneous()).head<3>() : ((m0*x + m1*y) + m2*z) + m3
        t[r] = ((T[r * 4 + 0] * q[0] + T[r * 4 + 1] * q[1]) + T[r * 4 + 2] * q[2]) + T[r * 4 + 3];
      bool hit;
      if (use_kdtree) hit = kd.closest(t, sq_eps) != -1;
      else hit = brute_hit(t, sq_eps);
      ++*queries;
      if (hit) good_points++;
      if (!full_counts && number_of_points - i + good_points < terminate_value) break;
    }
    if (good_out) *good_out = good_points;
    return float(good_points) / float(number_of_points);
  }
  bool brute_hit(const float* t, float sq_eps) const {
    for (const auto& p : Ps) {
      float d[3] = {t[0] - p.pos[0], t[1] - p.pos[1], t[2] - p.pos[2]};
      if (sqn3(d) <= sq_eps) return true;
    }
    return false;
  }

  // ---- match4pcsBase.hpp:363-497  TryCongruentSet ------------------------------
  // per_cand (optional): for each quad: -1 if not verified (gate failed), else inlier count
  bool try_congruent_set(int b1, int b2, int b3, int b4, const std::vector<std::array<int, 4>>& quads,
                         size_t& nbCongruent, std::vector<int>* per_cand, unsigned* best_count, int* best_index) {
    const double pi = std::acos(-1);
    const P3 cbase[4] = {Ps[b1], Ps[b2], Ps[b3], Ps[b4]};
    float centroid1[3];
    for (int k = 0; k < 3; ++k) centroid1[k] = ((cbase[0].pos[k] + cbase[1].pos[k]) + cbase[2].pos[k]) / 3.f;
    size_t nb = 0;
    if (per_cand) per_cand->assign(quads.size(), -1);
    if (omp_threads > 1) return try_congruent_set_omp(b1, b2, b3, b4, cbase, centroid1, quads, nbCongruent, per_cand, best_count, best_index);
    for (int i = 0; i < int(quads.size()); ++i) {
      if (budget_seconds > 0 && (i & 15) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - budget_t0).count() > budget_seconds) {
        budget_hit = true;   // bounded cpu_baseline sample: not a reference behaviour
        break;
      }
      const int a = quads[i][0], b = quads[i][1], c = quads[i][2], d = quads[i][3];
      const P3 cc[4] = {Qs[a], Qs[b], Qs[c], Qs[d]};
      float centroid2[3];
      for (int k = 0; k < 3; ++k) centroid2[k] = ((cc[0].pos[k] + cc[1].pos[k]) + cc[2].pos[k]) / 3.f;
      float rms = -1; float T[16];
      const bool ok = compute_rigid(cbase, cc, centroid1, centroid2, float(double(opt.max_angle) * pi / 180.0), T, rms);
      if (ok && rms >= 0.f) {
        if (rms < 2.0f * opt.delta) {
          nb++;
          unsigned good = 0;
          float lcp = verify(T, &good);
          n_verified++;
          if (per_cand) (*per_cand)[i] = int(good);
          if (lcp > best_LCP) {                                    // :467-484
            base_ids[0] = b1; base_ids[1] = b2; base_ids[2] = b3; base_ids[3] = b4;
            current_congruent[0] = a; current_congruent[1] = b; current_congruent[2] = c; current_congruent[3] = d;
            best_LCP = lcp;
            std::memcpy(transform, T, sizeof(T));
            std::memcpy(qcentroid1, centroid1, sizeof(centroid1));
            std::memcpy(qcentroid2, centroid2, sizeof(centroid2));
            if (best_count) *best_count = good;
            if (best_index) *best_index = i;
          }
        }
      }
    }
    nbCongruent = nb;
    return best_LCP > opt.terminate_threshold;
  }

  // Baseline B of BASELINE.md section 3 ("best-effort CPU"): the candidate loop of TryCongruentSet under
  // `#pragma omp parallel for`, as the reference's legacy Match4PCS does by default (match4pcsBase.h:190-192,
  // match4pcsBase.hpp:390-393).  Every thread scores its candidates with the early exit against the best LCP the
  // base STARTED with (a read-only snapshot), then one ordered pass applies "first strictly greater wins": an
  // abandoned candidate could not have won, so base, winner and transform equal the serial loop's.
  // BENCH/TEST INFRASTRUCTURE, like the rest of this file.
  int omp_threads = 1;
  bool try_congruent_set_omp(int b1, int b2, int b3, int b4, const P3* cbase, const float* centroid1,
                             const std::vector<std::array<int, 4>>& quads, size_t& nbCongruent, std::vector<int>* per_cand,
                             unsigned* best_count, int* best_index) {
    const double pi = std::acos(-1);
    const int K = int(quads.size());
    std::vector<int> good_of(size_t(K), -1);
    const float snapshot = best_LCP;
    uint64_t queries = 0; size_t nb = 0;
    bool stop = false;
#pragma omp parallel for schedule(dynamic, 16) num_threads(omp_threads) reduction(+ : queries, nb)
    for (int i = 0; i < K; ++i) {
      if (stop) continue;
      if (budget_seconds > 0 && (i & 15) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - budget_t0).count() > budget_seconds) {
        stop = true;          // benign race: a flag that only ever goes to true
        continue;
      }
      const P3 cc[4] = {Qs[quads[i][0]], Qs[quads[i][1]], Qs[quads[i][2]], Qs[quads[i][3]]};
      float centroid2[3];
      for (int k = 0; k < 3; ++k) centroid2[k] = ((cc[0].pos[k] + cc[1].pos[k]) + cc[2].pos[k]) / 3.f;
      float rms = -1; float T[16];
      const bool ok = compute_rigid(cbase, cc, centroid1, centroid2, float(double(opt.max_angle) * pi / 180.0), T, rms);
      if (ok && rms >= 0.f && rms < 2.0f * opt.delta) {
        unsigned good = 0;
        verify_against(T, snapshot, &good, &queries);
        good_of[size_t(i)] = int(good);
        nb++;
      }
    }
    if (stop) budget_hit = true;
    n_verify_queries += queries; n_verified += nb;
    for (int i = 0; i < K; ++i) {                                   // ordered selection, match4pcsBase.hpp:467-484
      if (good_of[size_t(i)] < 0) continue;
      if (per_cand) (*per_cand)[size_t(i)] = good_of[size_t(i)];
      const float lcp = float(unsigned(good_of[size_t(i)])) / float(Qs.size());
      if (lcp > best_LCP) {
        const P3 cc[4] = {Qs[quads[i][0]], Qs[quads[i][1]], Qs[quads[i][2]], Qs[quads[i][3]]};
        float centroid2[3];
        for (int k = 0; k < 3; ++k) centroid2[k] = ((cc[0].pos[k] + cc[1].pos[k]) + cc[2].pos[k]) / 3.f;
        float rms = -1; float T[16];
        compute_rigid(cbase, cc, centroid1, centroid2, float(double(opt.max_angle) * pi / 180.0), T, rms);
        base_ids[0] = b1; base_ids[1] = b2; base_ids[2] = b3; base_ids[3] = b4;
        for (int k = 0; k < 4; ++k) current_congruent[k] = quads[i][k];
        best_LCP = lcp;
        std::memcpy(transform, T, sizeof(T));
        std::memcpy(qcentroid1, centroid1, 3 * sizeof(float));
        std::memcpy(qcentroid2, centroid2, sizeof(centroid2));
        if (best_count) *best_count = unsigned(good_of[size_t(i)]);
        if (best_index) *best_index = i;
      }
    }
    nbCongruent = nb;
    return best_LCP > opt.terminate_threshold;
  }

  // ---- match4pcsBase.hpp:281-360  TryOneBase ------------------------------------
  bool try_one_base() {
    using clk = std::chrono::steady_clock;
    Trace tr; std::memset(&tr, 0, sizeof(tr)); tr.best_index = -1;
    float invariant1, invariant2; int b1, b2, b3, b4;
    auto t0 = clk::now();
    bool sel = select_quadrilateral(invariant1, invariant2, b1, b2, b3, b4);
    t_select += std::chrono::duration<double>(clk::now() - t0).count();
    tr.ok_select = sel;
    if (!sel) { if (keep_trace) trace.push_back(tr); return false; }
    tr.base[0] = b1; tr.base[1] = b2; tr.base[2] = b3; tr.base[3] = b4; tr.inv1 = invariant1; tr.inv2 = invariant2;
    float d01[3], d23[3], n01[3], n23[3];
    sub3(base3D[0].pos, base3D[1].pos, d01); sub3(base3D[2].pos, base3D[3].pos, d23);
    const float distance1 = norm3(d01), distance2 = norm3(d23);
    sub3(base3D[0].nrm, base3D[1].nrm, n01); sub3(base3D[2].nrm, base3D[3].nrm, n23);
    const float normal_angle1 = norm3(n01), normal_angle2 = norm3(n23);
    std::vector<std::pair<int, int>> pairs1, pairs2;
    std::vector<std::array<int, 4>> quads;
    t0 = clk::now();
    extract_pairs(distance1, normal_angle1, 2.0f * opt.delta, 0, 1, pairs1);
    extract_pairs(distance2, normal_angle2, 2.0f * opt.delta, 2, 3, pairs2);
    t_pairs += std::chrono::duration<double>(clk::now() - t0).count();
    tr.m1 = int(pairs1.size()); tr.m2 = int(pairs2.size());
    n_pairs += pairs1.size() + pairs2.size();
    if (pairs1.empty() || pairs2.empty()) { if (keep_trace) trace.push_back(tr); return false; }
    t0 = clk::now();
    bool fq = find_congruent(invariant1, invariant2, 2.0f * opt.delta, 2.0f * opt.delta, pairs1, pairs2, quads);
    t_quads += std::chrono::duration<double>(clk::now() - t0).count();
    tr.K = int(quads.size());
    n_quads += quads.size();
    if (!fq) { if (keep_trace) trace.push_back(tr); return false; }
    size_t nb = 0;
    t0 = clk::now();
    unsigned bc = 0; int bi = -1;
    bool match = try_congruent_set(b1, b2, b3, b4, quads, nb, nullptr, &bc, &bi);
    t_verify += std::chrono::duration<double>(clk::now() - t0).count();
    tr.C = int(nb); tr.best_count = bc; tr.best_index = bi;
    if (keep_trace) trace.push_back(tr);
    return match;
  }

  // ---- global transform: match4pcsBase.hpp:224-229.  computeRotationScaling of a
  // rigid [R|t] gives rot*scale == R up to SVD round-off; the restatement uses the
  // linear part directly (rot*scale is re-multiplied in the reference; difference is
  // O(1e-7) and only touches the translation column).
  void global_transform(float* M) const {
    std::memcpy(M, transform, 16 * sizeof(float));
    float a[3];
    for (int k = 0; k < 3; ++k) a[k] = qcentroid2[k] + centroidQ[k];
    for (int r = 0; r < 3; ++r) {
      float ra = transform[r * 4 + 0] * a[0] + (transform[r * 4 + 1] * a[1] + transform[r * 4 + 2] * a[2]);
      M[r * 4 + 3] = (qcentroid1[r] + centroidP[r]) - ra;
    }
    M[12] = 0; M[13] = 0; M[14] = 0; M[15] = 1;
  }

  // ---- match4pcsBase.hpp:61-86, 208-274 ---------------------------------------
  // Q (full res) is transformed in place if the LCP improved.  Returns best LCP.
  float compute_transformation(const std::vector<P3>& P, std::vector<P3>* Q, float* M) {
    if (Q == nullptr) return 1e9f;
    if (P.empty() || Q->empty()) return 1e9f;
    init(P, *Q);
    if (best_LCP != 1.f) perform_n_steps(number_of_trials, M, Q);
    return best_LCP;
  }
  bool perform_n_steps(int n, float* M, std::vector<P3>* Q) {
    using sclock = std::chrono::system_clock;
    float last_best = best_LCP;
    bool ok = false;
    auto t0 = sclock::now();
    for (int i = current_trial; i < current_trial + n; ++i) {
      ok = try_one_base();
      float fraction_try = float(i) / float(number_of_trials);
      float fraction_time = float(std::chrono::duration_cast<std::chrono::seconds>(sclock::now() - t0).count() /
                                  opt.max_time_seconds);                       // integer division: quirk :240-243
      float fraction = std::max(fraction_time, fraction_try);
      std::memcpy(M, transform, sizeof(transform));
      if (ok || i > number_of_trials || fraction >= 0.99 || best_LCP == 1.0) break;
    }
    current_trial += n;
    if (best_LCP > last_best) {
      global_transform(M);
      if (Q) {
        for (auto& p : *Q) {                                                  // :265-267 (4x4 * homogeneous, packet order)
          float x = p.pos[0], y = p.pos[1], z = p.pos[2];
          for (int r = 0; r < 3; ++r) p.pos[r] = ((M[r * 4 + 0] * x + M[r * 4 + 1] * y) + M[r * 4 + 2] * z) + M[r * 4 + 3];
        }
      }
    }
    return ok || current_trial >= number_of_trials;
  }
};

}  // namespace s4po

// ============================================================================
// C API for ctypes (tests / bench cpu_baseline / smoke only)
// ============================================================================
using namespace s4po;

static std::vector<P3> make_cloud(const float* xyz, const float* nrm, const float* rgb, uint64_t n) {
  std::vector<P3> c(n);
  for (uint64_t i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) c[i].pos[k] = xyz[3 * i + k];
    if (nrm) for (int k = 0; k < 3; ++k) c[i].nrm[k] = nrm[3 * i + k];
    if (rgb) for (int k = 0; k < 3; ++k) c[i].rgb[k] = rgb[3 * i + k];
  }
  return c;
}

extern "C" {

struct s4po_options {
  float delta, max_normal_difference, max_translation_distance, max_angle, max_color_distance;
  uint64_t sample_size;
  int32_t max_time_seconds;
  uint32_t random_seed;
  float terminate_threshold, overlap_estimation;
};

struct s4po_stats {
  uint64_t n_verified, n_quads, n_pairs, n_verify_queries;
  double t_pairs, t_quads, t_verify, t_select;
  int32_t number_of_trials, current_trial, n_P, n_Q;
  float best_lcp, p_diameter;
};

void* s4po_create(const s4po_options* o) {
  Options opt;
  opt.delta = o->delta; opt.max_normal_difference = o->max_normal_difference;
  opt.max_translation_distance = o->max_translation_distance; opt.max_angle = o->max_angle;
  opt.max_color_distance = o->max_color_distance; opt.sample_size = o->sample_size;
  opt.max_time_seconds = o->max_time_seconds; opt.randomSeed = o->random_seed;
  opt.terminate_threshold = o->terminate_threshold; opt.overlap_estimation = o->overlap_estimation;
  return new Matcher(opt);
}
void s4po_destroy(void* h) { delete static_cast<Matcher*>(h); }
// bench-only: bound the CPU sample; the clock starts now.  Returns nothing; s4po_budget_hit() tells if it tripped.
void s4po_set_budget(void* h, double seconds) {
  Matcher* m = static_cast<Matcher*>(h);
  m->budget_seconds = seconds; m->budget_t0 = std::chrono::steady_clock::now(); m->budget_hit = false;
}
int32_t s4po_budget_hit(void* h) { return static_cast<Matcher*>(h)->budget_hit ? 1 : 0; }
// bench-only: number of OpenMP threads of the candidate loop (1 = the reference's serial loop)
void s4po_set_threads(void* h, int n) { static_cast<Matcher*>(h)->omp_threads = n < 1 ? 1 : n; }
void s4po_set_mode(void* h, int full_counts, int use_kdtree, int keep_trace) {
  Matcher* m = static_cast<Matcher*>(h);
  m->full_counts = full_counts != 0; m->use_kdtree = use_kdtree != 0; m->keep_trace = keep_trace != 0;
}

// sampler known-answer entry: returns the number of kept points, writes their input indices if out != NULL
uint64_t s4po_sample(const float* xyz, uint64_t n, float delta, float* out_xyz) {
  std::vector<P3> c = make_cloud(xyz, nullptr, nullptr, n), o;
  uniform_dist_sample(c, delta, o);
  if (out_xyz) for (size_t i = 0; i < o.size(); ++i) for (int k = 0; k < 3; ++k) out_xyz[3 * i + k] = o[i].pos[k];
  return o.size();
}

void s4po_init(void* h, const float* Pxyz, const float* Pn, const float* Prgb, uint64_t nP,
               const float* Qxyz, const float* Qn, const float* Qrgb, uint64_t nQ) {
  Matcher* m = static_cast<Matcher*>(h);
  m->init(make_cloud(Pxyz, Pn, Prgb, nP), make_cloud(Qxyz, Qn, Qrgb, nQ));
}

void s4po_set_sampled(void* h, const float* Pxyz, uint64_t nP, const float* Qxyz, uint64_t nQ) {
  static_cast<Matcher*>(h)->set_sampled(make_cloud(Pxyz, nullptr, nullptr, nP), make_cloud(Qxyz, nullptr, nullptr, nQ));
}

void s4po_get_stats(void* h, s4po_stats* s) {
  Matcher* m = static_cast<Matcher*>(h);
  s->n_verified = m->n_verified; s->n_quads = m->n_quads; s->n_pairs = m->n_pairs; s->n_verify_queries = m->n_verify_queries;
  s->t_pairs = m->t_pairs; s->t_quads = m->t_quads; s->t_verify = m->t_verify; s->t_select = m->t_select;
  s->number_of_trials = m->number_of_trials; s->current_trial = m->current_trial;
  s->n_P = int32_t(m->Ps.size()); s->n_Q = int32_t(m->Qs.size());
  s->best_lcp = m->best_LCP; s->p_diameter = m->P_diameter;
}

// which: 0 = sampled P (centred), 1 = sampled Q (centred), 2 = unit-cube Q.  xyz/nrm/rgb may be NULL.
void s4po_get_cloud(void* h, int which, float* xyz, float* nrm, float* rgb) {
  Matcher* m = static_cast<Matcher*>(h);
  if (which == 2) { for (size_t i = 0; i < m->upts.size(); ++i) for (int k = 0; k < 3; ++k) xyz[3 * i + k] = m->upts[i][k]; return; }
  const std::vector<P3>& c = which == 0 ? m->Ps : m->Qs;
  for (size_t i = 0; i < c.size(); ++i) for (int k = 0; k < 3; ++k) {
    if (xyz) xyz[3 * i + k] = c[i].pos[k];
    if (nrm) nrm[3 * i + k] = c[i].nrm[k];
    if (rgb) rgb[3 * i + k] = c[i].rgb[k];
  }
}
void s4po_get_frame(void* h, float* centroidP, float* centroidQ, float* gcenter, float* ratio) {
  Matcher* m = static_cast<Matcher*>(h);
  for (int k = 0; k < 3; ++k) { centroidP[k] = m->centroidP[k]; centroidQ[k] = m->centroidQ[k]; gcenter[k] = m->gcenter[k]; }
  *ratio = m->ratio;
}

int32_t s4po_select_quadrilateral(void* h, float* inv1, float* inv2, int32_t* base, float* base_xyz) {
  Matcher* m = static_cast<Matcher*>(h);
  int b1, b2, b3, b4;
  bool ok = m->select_quadrilateral(*inv1, *inv2, b1, b2, b3, b4);
  base[0] = b1; base[1] = b2; base[2] = b3; base[3] = b4;
  if (base_xyz) for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) base_xyz[3 * i + k] = m->base3D[i].pos[k];
  return ok;
}
// sets base_3D_ directly from sampled-P indices (already ordered), for kernel-level tests
void s4po_set_base(void* h, const int32_t* base) {
  Matcher* m = static_cast<Matcher*>(h);
  for (int i = 0; i < 4; ++i) m->base3D[i] = m->Ps[base[i]];
}
void s4po_get_base(void* h, float* xyz, float* nrm, float* rgb) {
  Matcher* m = static_cast<Matcher*>(h);
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) {
    if (xyz) xyz[3 * i + k] = m->base3D[i].pos[k];
    if (nrm) nrm[3 * i + k] = m->base3D[i].nrm[k];
    if (rgb) rgb[3 * i + k] = m->base3D[i].rgb[k];
  }
}

// returns m (number of ordered pairs); writes at most cap pairs (2 ints each)
int64_t s4po_extract_pairs(void* h, float pair_distance, float pair_normals_angle, float pair_distance_epsilon,
                           int32_t bp1, int32_t bp2, int32_t* out_pairs, int64_t cap) {
  Matcher* m = static_cast<Matcher*>(h);
  m->extract_pairs(pair_distance, pair_normals_angle, pair_distance_epsilon, bp1, bp2, m->last_pairs);
  int64_t n = int64_t(m->last_pairs.size());
  for (int64_t i = 0; i < n && i < cap; ++i) { out_pairs[2 * i] = m->last_pairs[i].first; out_pairs[2 * i + 1] = m->last_pairs[i].second; }
  return n;
}
void s4po_get_ids(void* h, uint32_t* out) {
  Matcher* m = static_cast<Matcher*>(h);
  for (size_t i = 0; i < m->ids.size(); ++i) out[i] = m->ids[i];
}

int64_t s4po_find_congruent(void* h, float inv1, float inv2, float thr, const int32_t* pairs1, int64_t m1,
                            const int32_t* pairs2, int64_t m2, int32_t* out_quads, int64_t cap) {
  Matcher* m = static_cast<Matcher*>(h);
  std::vector<std::pair<int, int>> p1(m1), p2(m2);
  for (int64_t i = 0; i < m1; ++i) p1[i] = {pairs1[2 * i], pairs1[2 * i + 1]};
  for (int64_t i = 0; i < m2; ++i) p2[i] = {pairs2[2 * i], pairs2[2 * i + 1]};
  std::vector<std::array<int, 4>> quads;
  m->find_congruent(inv1, inv2, thr, thr, p1, p2, quads);
  int64_t K = int64_t(quads.size());
  for (int64_t i = 0; i < K && i < cap; ++i) for (int k = 0; k < 4; ++k) out_quads[4 * i + k] = quads[i][k];
  return K;
}

// Streaming counterpart of s4po_find_congruent (+ the rms gate of TryCongruentSet when base != NULL): counts and
// checksums only, OpenMP over set 2.  out4 = {quads, quad checksum, gated quads, their checksum}; gated quads with
// quad_mix % sample_mod == 0 are written to sample_quads (sorted), *n_sample = how many there are.
void s4po_count_congruent(void* h, float inv1, float inv2, float thr, const int32_t* pairs1, int64_t m1,
                          const int32_t* pairs2, int64_t m2, const int32_t* base, int32_t threads, uint64_t* out4,
                          uint64_t sample_mod, int32_t* sample_quads, int64_t sample_cap, int64_t* n_sample) {
  Matcher* m = static_cast<Matcher*>(h);
  std::vector<std::pair<int, int>> p1(m1), p2(m2);
  for (int64_t i = 0; i < m1; ++i) p1[i] = {pairs1[2 * i], pairs1[2 * i + 1]};
  for (int64_t i = 0; i < m2; ++i) p2[i] = {pairs2[2 * i], pairs2[2 * i + 1]};
  std::vector<std::array<int, 4>> sample;
  int b4[4] = {0, 0, 0, 0};
  if (base) for (int k = 0; k < 4; ++k) b4[k] = base[k];
  m->count_congruent(inv1, inv2, thr, p1, p2, base ? b4 : nullptr, threads < 1 ? 1 : threads, out4, sample_mod,
                     sample_quads ? &sample : nullptr);
  if (n_sample) *n_sample = int64_t(sample.size());
  for (int64_t i = 0; i < int64_t(sample.size()) && i < sample_cap; ++i) for (int k = 0; k < 4; ++k) sample_quads[4 * i + k] = sample[size_t(i)][k];
}
// The same streaming enumeration with every gated candidate VERIFIED in full: best4 = {found, greatest inlier count, index
// of the winner's set-1 pair, index of its set-2 pair} -- the candidate TryCongruentSet would keep (first maximum in the
// reference's candidate order).  Cost: candidates x n_Q kd-tree queries; for sizes the host can afford.
void s4po_count_congruent_best(void* h, float inv1, float inv2, float thr, const int32_t* pairs1, int64_t m1,
                               const int32_t* pairs2, int64_t m2, const int32_t* base, int32_t threads, uint64_t* out4, uint64_t* best4) {
  Matcher* m = static_cast<Matcher*>(h);
  std::vector<std::pair<int, int>> p1(m1), p2(m2);
  for (int64_t i = 0; i < m1; ++i) p1[i] = {pairs1[2 * i], pairs1[2 * i + 1]};
  for (int64_t i = 0; i < m2; ++i) p2[i] = {pairs2[2 * i], pairs2[2 * i + 1]};
  int b4[4] = {base[0], base[1], base[2], base[3]};
  m->count_congruent(inv1, inv2, thr, p1, p2, b4, threads < 1 ? 1 : threads, out4, 0, nullptr, best4);
}
uint64_t s4po_quad_mix(int32_t a, int32_t b, int32_t c, int32_t d) { return Matcher::quad_mix(a, b, c, d); }

// TryCongruentSet on explicit quads.  per_cand[i] = -1 (gate failed) or inlier count.  Updates best state.
int64_t s4po_try_congruent_set(void* h, const int32_t* base, const int32_t* quads, int64_t K, int32_t* per_cand,
                               uint32_t* best_count, int32_t* best_index) {
  Matcher* m = static_cast<Matcher*>(h);
  std::vector<std::array<int, 4>> q(K);
  for (int64_t i = 0; i < K; ++i) for (int k = 0; k < 4; ++k) q[i][k] = quads[4 * i + k];
  size_t nb = 0;
  std::vector<int> pc;
  unsigned bc = 0; int bi = -1;
  m->try_congruent_set(base[0], base[1], base[2], base[3], q, nb, per_cand ? &pc : nullptr, &bc, &bi);
  if (per_cand) for (int64_t i = 0; i < K; ++i) per_cand[i] = pc[i];
  if (best_count) *best_count = bc;
  if (best_index) *best_index = bi;
  return int64_t(nb);
}

// rigid transform of one candidate (row-major T), returns ok flag; rms out
int32_t s4po_compute_rigid(void* h, const int32_t* base, const int32_t* quad, float* T, float* rms) {
  Matcher* m = static_cast<Matcher*>(h);
  const P3 cb[4] = {m->Ps[base[0]], m->Ps[base[1]], m->Ps[base[2]], m->Ps[base[3]]};
  const P3 cc[4] = {m->Qs[quad[0]], m->Qs[quad[1]], m->Qs[quad[2]], m->Qs[quad[3]]};
  float c1[3], c2[3];
  for (int k = 0; k < 3; ++k) { c1[k] = ((cb[0].pos[k] + cb[1].pos[k]) + cb[2].pos[k]) / 3.f; c2[k] = ((cc[0].pos[k] + cc[1].pos[k]) + cc[2].pos[k]) / 3.f; }
  const double pi = std::acos(-1);
  return m->compute_rigid(cb, cc, c1, c2, float(double(m->opt.max_angle) * pi / 180.0), T, *rms);
}

// Verify(T) for B row-major transforms -> inlier counts (full counts, no early exit)
void s4po_verify_batch(void* h, const float* T, int64_t B, uint32_t* counts) {
  Matcher* m = static_cast<Matcher*>(h);
  bool fc = m->full_counts; m->full_counts = true;
  for (int64_t b = 0; b < B; ++b) { unsigned g = 0; m->verify(T + 16 * b, &g); counts[b] = g; }
  m->full_counts = fc;
}

int32_t s4po_try_one_base(void* h) { return static_cast<Matcher*>(h)->try_one_base(); }

int64_t s4po_get_trace(void* h, int32_t* out /* 11 ints per record */, float* out_inv /* 2 per record */, int64_t cap) {
  Matcher* m = static_cast<Matcher*>(h);
  int64_t n = int64_t(m->trace.size());
  for (int64_t i = 0; i < n && i < cap; ++i) {
    const Trace& t = m->trace[i];
    int32_t* o = out + 11 * i;
    o[0] = t.ok_select; o[1] = t.base[0]; o[2] = t.base[1]; o[3] = t.base[2]; o[4] = t.base[3];
    o[5] = t.m1; o[6] = t.m2; o[7] = t.K; o[8] = t.C; o[9] = int32_t(t.best_count); o[10] = t.best_index;
    out_inv[2 * i] = t.inv1; out_inv[2 * i + 1] = t.inv2;
  }
  return n;
}

void s4po_get_best(void* h, float* transform_rowmajor, float* lcp, int32_t* base, int32_t* congruent,
                   float* qcentroid1, float* qcentroid2) {
  Matcher* m = static_cast<Matcher*>(h);
  std::memcpy(transform_rowmajor, m->transform, 16 * sizeof(float));
  *lcp = m->best_LCP;
  for (int i = 0; i < 4; ++i) { base[i] = m->base_ids[i]; congruent[i] = m->current_congruent[i]; }
  for (int k = 0; k < 3; ++k) { qcentroid1[k] = m->qcentroid1[k]; qcentroid2[k] = m->qcentroid2[k]; }
}

// Full registration.  Qxyz is transformed in place when the LCP improved.  M = row-major 4x4.
float s4po_compute_transformation(void* h, const float* Pxyz, const float* Pn, const float* Prgb, uint64_t nP,
                                  float* Qxyz, const float* Qn, const float* Qrgb, uint64_t nQ, float* M) {
  Matcher* m = static_cast<Matcher*>(h);
  if (Qxyz == nullptr) return 1e9f;
  std::vector<P3> P = make_cloud(Pxyz, Pn, Prgb, nP);
  std::vector<P3> Q = make_cloud(Qxyz, Qn, Qrgb, nQ);
  for (int i = 0; i < 16; ++i) M[i] = (i % 5 == 0) ? 1.f : 0.f;
  float r = m->compute_transformation(P, &Q, M);
  for (uint64_t i = 0; i < nQ; ++i) for (int k = 0; k < 3; ++k) Qxyz[3 * i + k] = Q[i].pos[k];
  return r;
}

}  // extern "C"
