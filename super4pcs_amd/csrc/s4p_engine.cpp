// s4p_engine.cpp -- host RANSAC driver behind include/s4p_matcher.h.
//
// Host-side parts of the reference that fix the RNG stream and the inputs of the hot
// path and therefore have to be reproduced exactly:
//   UniformDistSampler                    src/super4pcs/sampling.h:59-122
//   Match4PCSBase::init                   src/super4pcs/algorithms/match4pcsBase.hpp:90-203
//   SelectRandomTriangle / TryQuadrilateral / SelectQuadrilateral / distSegmentToSegment
//                                         src/super4pcs/algorithms/match4pcsBase.cc:64-131,185-351
//   TryOneBase / Perform_N_steps          match4pcsBase.hpp:208-360
// Everything data-parallel (pairs, quads, rigid transform, LCP, final apply) is delegated to
// the gfx950 kernels through the C ABI of s4p_capi.h.  No CPU fallback exists for those.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <mutex>
#include <atomic>
#include <random>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "s4p_host_structs.hpp"
#include "s4p_matcher.h"

namespace s4p {
long long gpu_uniform_dist_sample(const float* x, const float* y, const float* z, long long n, float delta, long long* out_index, int device);
}

namespace {

struct Cloud {
  std::vector<float> x, y, z, nx, ny, nz, r, g, b;
  bool has_n = false, has_c = false;
  size_t size() const { return x.size(); }
  void reserve(size_t n) { x.reserve(n); y.reserve(n); z.reserve(n); }
  void push_from(const s4p_cloud_view& v, int64_t i) {
    x.push_back(v.x[i]); y.push_back(v.y[i]); z.push_back(v.z[i]);
    if (has_n) { nx.push_back(v.nx[i]); ny.push_back(v.ny[i]); nz.push_back(v.nz[i]); }
    if (has_c) { r.push_back(v.r[i]); g.push_back(v.g[i]); b.push_back(v.b[i]); }
  }
  void init_flags(const s4p_cloud_view& v) { has_n = v.nx && v.ny && v.nz; has_c = v.r && v.g && v.b; }
};

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }      // Eigen: x + (y + z)
inline float sqn(V3 a) { return dot(a, a); }
inline float len(V3 a) { return std::sqrt(sqn(a)); }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct VoxelKey { int32_t a, b, c; bool operator==(const VoxelKey& o) const { return a == o.a && b == o.b && c == o.c; } };
struct VoxelHash {
  size_t operator()(const VoxelKey& k) const {
    uint64_t h = uint64_t(uint32_t(k.a)) * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t(uint32_t(k.b)) + 0x7F4A7C15ull + (h << 6) + (h >> 2));
    h ^= (uint64_t(uint32_t(k.c)) * 0xC2B2AE3D27D4EB4Full + (h << 7) + (h >> 3));
    return size_t(h);
  }
};

constexpr int64_t kGpuSamplerMin = 32768;

// sampling.h:104-121: first point per delta-voxel, voxel = int(floor(coord * (1.0f / delta))).
// Large clouds go through the device sampler (s4p_sampler.hip) when a GPU is visible; the host hash below is the
// same function for small clouds, for voxel coordinates beyond +-2^20 and for facade users that sample on a machine
// without any HIP device (the Sampler concept of the reference is host code, sampling.h).  A HIP *error* of the device
// sampler is not papered over: it is returned as -1 (and was printed to stderr by the sampler).
int64_t voxel_first_hits(const float* x, const float* y, const float* z, int64_t n, float delta, int64_t* out, int device = -1) {
  if (n >= kGpuSamplerMin) {
    const char* force = std::getenv("S4P_SAMPLER");
    if (!(force && std::strcmp(force, "host") == 0)) {
      static_assert(sizeof(long long) == sizeof(int64_t), "index type");
      const long long k = s4p::gpu_uniform_dist_sample(x, y, z, n, delta, reinterpret_cast<long long*>(out), device);
      if (k >= 0) return int64_t(k);
      if (k == -2) return -1;
    }
  }
  const float scale = 1.0f / delta;
  std::unordered_map<VoxelKey, char, VoxelHash> seen;
  seen.reserve(size_t(n / 4 + 16));
  int64_t kept = 0;
  for (int64_t i = 0; i < n; ++i) {
    VoxelKey k{int32_t(std::floor(x[i] * scale)), int32_t(std::floor(y[i] * scale)), int32_t(std::floor(z[i] * scale))};
    if (seen.emplace(k, 1).second) out[kept++] = i;
  }
  return kept;
}

}  // namespace

struct s4p_matcher {
  s4p_options opt{};
  s4p_ctx* ctx = nullptr;
  int device = 0;
  std::string err;
  std::mt19937 rng;
  Cloud Ps, Qs;
  float centroid_p[3] = {0, 0, 0}, centroid_q[3] = {0, 0, 0};
  float p_diameter = 0.f, max_base_diameter = -1.f;
  std::vector<float> P4;                 // sampled P as (x, y, z, 0) records: one cache line per random draw of the base search
  s4p::FourthPointIndex fourth;          // block-pruned 4th-point search over the sampled P (host selection mode)
  // Where SelectQuadrilateral's two searches run: -1 = by size (device from kDeviceSelectMin sampled P points), 0 = host
  // structures above, 1 = device reductions over the resident P (s4p_select_base_points); fixed by init.
  int select_mode = -1; bool device_select = false;
  int number_of_trials = 0, current_trial = 0;
  float best_lcp = 0.f; uint32_t best_count = 0;
  float transform[16];
  float qc1[3] = {0, 0, 0}, qc2[3] = {0, 0, 0};
  int base[4] = {0, 0, 0, 0}, congruent[4] = {0, 0, 0, 0};
  // base_3D_
  int b3_id[4] = {0, 0, 0, 0};
  uint64_t candidates_verified = 0, quads_total = 0, pairs_total = 0, bases_tried = 0;
  double seconds_select = 0, seconds_device = 0;
  bool ready = false;
  int64_t init_generation = 0;           // counts s4p_matcher_init calls (drivers keep per-registration state)
  bool grow_on_overflow = true;          // a lane whose base overflows grows its buffers and redoes the base (s4p_set_auto_grow)
  std::atomic<bool> select_failed{false}; std::string select_err;   // a device selection attempt returned an error (possibly on the selector thread)
  bool visit_candidates = false;         // issue the reference's per-candidate visitor calls (fraction == -1)
  // Early exit (s4p_set_best_hint): inside the trial loops (Perform_N_steps, the sharded loop) the device may abandon
  // candidates that cannot exceed the best inlier count committed so far -- the reference's Verify early exit,
  // match4pcsBase.cc:520,558-560.  Never while a per-candidate visitor listens, never for the stage-level calls.
  bool early_exit = true;                // s4p_matcher_set_early_exit / S4P_EARLY_EXIT=0
  bool in_loop = false;                  // a trial loop is running: commits refresh the hint
  // pipelined trials: bases whose device pass is in flight (at most two)
  struct Prepared {
    bool found = false, device = false;
    int ids[4] = {0, 0, 0, 0};
    int slot = -1;                         // staging slot to recycle after the wait (producer mode)
    std::mt19937 rng_before;               // host state before this trial was prepared (for exact roll-back)
    std::vector<uint32_t> pair_state_before;
  };
  std::vector<Prepared> inflight;          // FIFO

  // ---- optional producer: base selection and octree staging on two helper threads ------------------------
  // Both are inherently sequential (RNG stream; persistent octree permutation) but independent of results and of
  // each other's resource, so they pipeline: selector -> qa -> octree/staging -> qb -> main thread (launch/commit).
  struct Trial {
    long index = 0;
    bool found = false, owned = false, staged = false;
    int ids[4] = {0, 0, 0, 0};
    float inv1 = 0, inv2 = 0;
    float bx[12], bn[12], bc[12];
    int slot = -1;
    std::mt19937 rng_before;
    std::vector<uint32_t> pair_before;
  };
  struct Producer {
    bool enabled = false, running = false, stop = false;
    int mode = 2;                          // helper threads: 0 never, 1 always, 2 where they pay (update_helper_threads)
    int rank = 0, world = 1;
    long next_index = 0;                   // next trial the selector will draw
    long consumed = 0;                     // trials handed to the main thread
    std::thread tree;
    // Base selection as a pipeline of its own: the DRAWER owns the random stream and draws the 2001 indices of one attempt
    // after the other (match4pcsBase.cc:185-218 draws them unconditionally, so the stream never depends on a result);
    // EVALUATORS (two on the host structures; one that batches attempts through s4p_select_base_points_batch when the
    // searches run on the device) find the widest triangle, the fourth point and the ordering of an attempt, several
    // attempts at a time; the ASSEMBLER consumes the attempts strictly in order and turns them into trials exactly as
    // the reference's loop does (a failed attempt is followed by the next one, :283-349) and feeds qa.  The serial part of
    // a trial drops from ~15 us (draw + evaluate) to the ~5 us of the draws, which is what bounds a rank that has to walk
    // the bases of all `world` ranks.
    struct Attempt { uint32_t idx[2001]; std::mt19937 rng_before; int status = -1; int ids[4] = {0, 0, 0, 0}; float inv1 = 0, inv2 = 0; };
    static constexpr uint32_t kRing = 64;
    std::unique_ptr<Attempt[]> ring;
    // state of a slot = 3 * lap + phase (lap = seq / kRing; phase 0 free, 1 drawn, 2 evaluated): every stage waits for the
    // exact (lap, phase) of ITS attempt.  With the bare phase an evaluator descheduled on attempt 5 for ~1 ms let the other one
    // work through 6..68 and claim 69, find slot 5 still "drawn" and evaluate attempt 5's draws as attempt 69.
    std::atomic<uint64_t> ring_state[kRing];
    static uint64_t tag(uint64_t seq, uint32_t phase) { return 3ull * (seq / kRing) + phase; }
    std::atomic<uint64_t> drawn{0}, claimed{0}, taken{0};
    std::thread drawer, evals[2], assembler;
    int n_eval = 2;
    std::atomic<uint64_t> select_ns{0};
    // S4P_TRACE_CHAIN=1 (lab aid): busy time of every stage of the host chain, printed when the helpers stop
    std::atomic<uint64_t> draw_ns{0}, tree_ns{0}, batches{0}, attempts_done{0}, trials_done{0};
    std::chrono::steady_clock::time_point started;
    bool partial = false; std::mt19937 partial_rng;     // the assembler stopped inside a trial: where that trial's draws began
    std::mutex mu;
    // one condition per wait reason: a hand-off wakes only the thread that can use it, and a full queue's
    // producer is woken at the low-water mark (half empty), not once per item
    std::condition_variable cv_a_space, cv_a_item, cv_b_space, cv_b_item, cv_slot;
    std::deque<Trial> qa, qb;
    // == qa.size() / qb.size(), readable without the lock: a starved stage spins on these for a few hundred
    // microseconds before it sleeps.  (When the launch thread waits on the helpers -- 8 trials per window at
    // world 8 -- every hand-off to a *sleeping* thread costs the notifier a futex wake + IPI; measured on the MI355X
    // host that quadrupled the per-trial cost of the chain.  Streaming stages that spin never sleep; when the GPU is
    // the bottleneck the queues fill up and the helpers sleep on "space", woken once per half queue.)
    std::atomic<size_t> qa_ready{0}, qb_ready{0};
    std::atomic<bool> stop_flag{false};
    static constexpr int kSpinMicros = 250;
    std::vector<int> free_slots;
    size_t cap_a = 24, cap_b = 8;          // set_sharding: >= 3 windows of trials, so the helpers run ahead of a whole window
    double select_s = 0;
    void wake_all() { cv_a_space.notify_all(); cv_a_item.notify_all(); cv_b_space.notify_all(); cv_b_item.notify_all(); cv_slot.notify_all(); }
  } prod;

  void set_identity() { for (int i = 0; i < 16; ++i) transform[i] = (i % 5 == 0) ? 1.f : 0.f; }
  V3 P(int i) const { return {Ps.x[i], Ps.y[i], Ps.z[i]}; }
  int32_t fail(int32_t code, const std::string& m) { err = m; return code; }
  int32_t ctx_fail(int32_t code) { err = s4p_last_error(ctx); return code; }
};

namespace {

void centre_cloud(Cloud& c, float* cen) {               // match4pcsBase.hpp:142-149
  cen[0] = cen[1] = cen[2] = 0.f;
  const size_t n = c.size();
  for (size_t i = 0; i < n; ++i) { cen[0] += c.x[i]; cen[1] += c.y[i]; cen[2] += c.z[i]; }
  const float fn = float(n);
  cen[0] /= fn; cen[1] /= fn; cen[2] /= fn;
  for (size_t i = 0; i < n; ++i) { c.x[i] -= cen[0]; c.y[i] -= cen[1]; c.z[i] -= cen[2]; }
}

// match4pcsBase.cc:64-131 with Scalar = double and float vectors (double * Vector3f promotes to float).
double segment_segment(V3 p1, V3 p2, V3 q1, V3 q2, double& inv1, double& inv2) {
  const double tiny = 0.0001;
  const V3 u = sub(p2, p1), v = sub(q2, q1), w = sub(p1, q1);
  const double a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  const double f = a * c - b * b;
  double s1 = 0.0, s2 = f, t1 = 0.0, t2 = f;
  if (f < tiny) { s1 = 0.0; s2 = 1.0; t1 = e; t2 = c; }
  else {
    s1 = b * e - c * d;
    t1 = a * e - b * d;
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
  inv1 = (std::abs(s1) < tiny ? 0.0 : s1 / s2);
  inv2 = (std::abs(t1) < tiny ? 0.0 : t1 / t2);
  const float f1 = float(inv1), f2 = float(inv2);
  const V3 r{(w.x + f1 * u.x) - f2 * v.x, (w.y + f1 * u.y) - f2 * v.y, (w.z + f1 * u.z) - f2 * v.z};
  return double(len(r));
}

// match4pcsBase.cc:185-218.  Same draws, same winner as the reference loop, arranged for the producer's serial chain:
//  * `rng() % n` through an exact multiply-high remainder (n is loop-invariant);
//  * the area test on the squared cross product first -- sqrt is monotone, so a candidate whose squared area does
//    not exceed the best one's cannot pass `wide > widest`; the square root is taken only for the few that can.
// the 2001 draws of one attempt: `first`, then 1000 x (second, third), each `rng() % n` (match4pcsBase.cc:192-199)
void draw_attempt(std::mt19937& rng, uint32_t n, uint32_t idx[2001]) {
  const uint64_t magic = ~0ull / n + 1ull;                       // exact for every 32-bit dividend
  for (int t = 0; t < 2001; ++t) {
    const uint32_t a = uint32_t(rng());                          // mt19937 yields 32-bit values
    idx[t] = uint32_t((static_cast<unsigned __int128>(magic * a) * n) >> 64);
  }
}
bool eval_triangle(const s4p_matcher* m, const uint32_t idx[2001], int& b1, int& b2, int& b3) {
  b1 = b2 = b3 = -1;
  const float* P = m->P4.data();
  for (int t = 0; t < 2001; ++t) __builtin_prefetch(P + 4 * size_t(idx[t]));       // the records the draws address
  const uint32_t first = idx[0];
  const float limit = m->max_base_diameter * m->max_base_diameter;
  float widest = 0.f, widest_sq = 0.f;
  const V3 o{P[4 * size_t(first)], P[4 * size_t(first) + 1], P[4 * size_t(first) + 2]};
  for (int t = 0; t < 1000; ++t) {
    const uint32_t second = idx[1 + 2 * t], third = idx[2 + 2 * t];
    const float* ps = P + 4 * size_t(second); const float* pt = P + 4 * size_t(third);
    const V3 u{ps[0] - o.x, ps[1] - o.y, ps[2] - o.z}, w{pt[0] - o.x, pt[1] - o.y, pt[2] - o.z};
    const float sq = sqn(cross(u, w));
    if (!(sq > widest_sq)) continue;
    if (!(sqn(u) < limit && sqn(w) < limit)) continue;
    const float wide = std::sqrt(sq);
    if (wide > widest) { widest = wide; widest_sq = sq; b1 = int(first); b2 = int(second); b3 = int(third); }
  }
  return b1 != -1 && b2 != -1 && b3 != -1;
}
bool pick_triangle(s4p_matcher* m, int& b1, int& b2, int& b3) {
  const uint32_t n = uint32_t(m->Ps.size());
  b1 = b2 = b3 = -1;
  if (n == 0) return false;
  uint32_t idx[2001];
  draw_attempt(m->rng, n, idx);
  return eval_triangle(m, idx, b1, b2, b3);
}

// match4pcsBase.cc:225-274: best of the 12 segment pairings; reorders ids.
bool order_quadrilateral(const V3 pts[4], int ids[4], float& inv1, float& inv2) {
  float best = std::numeric_limits<float>::max();
  int pick[4] = {-1, -1, -1, -1};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      if (i == j) continue;
      int k = 0; while (k == i || k == j) k++;
      int l = 0; while (l == i || l == j || l == k) l++;
      double a, b;
      const float dist = float(segment_segment(pts[i], pts[j], pts[k], pts[l], a, b));
      if (dist < best) { best = dist; pick[0] = i; pick[1] = j; pick[2] = k; pick[3] = l; inv1 = float(a); inv2 = float(b); }
    }
  if (pick[0] < 0 || pick[1] < 0 || pick[2] < 0 || pick[3] < 0) return false;
  const int old[4] = {ids[0], ids[1], ids[2], ids[3]};
  for (int t = 0; t < 4; ++t) ids[t] = old[pick[t]];
  return true;
}

// Device selection: the draws of one attempt are made here, the searches run in k_select_*.
// Returns the status of s4p_select_base_points, or -1 on a device error (m->err set).
constexpr size_t kDeviceSelectMin = size_t(1) << 20;
int device_attempt(s4p_matcher* m, int ids[4], V3 pts[4]) {
  const uint32_t n = uint32_t(m->Ps.size());
  if (n == 0) return 1;
  const uint64_t magic = ~0ull / n + 1ull;
  uint32_t idx[2001];
  for (int t = 0; t < 2001; ++t) {
    const uint32_t a = uint32_t(m->rng());
    idx[t] = uint32_t((static_cast<unsigned __int128>(magic * a) * n) >> 64);
  }
  const float kBaseTooSmall = 0.2f;
  const float limit = m->max_base_diameter * m->max_base_diameter;
  const float too_small = float(std::pow(double(m->max_base_diameter * kBaseTooSmall), 2));
  int32_t got[4], status = -1; float xyz[12];
  if (s4p_select_base_points(m->ctx, idx, limit, too_small, got, xyz, &status) != S4P_OK) { m->select_err = s4p_last_error(m->ctx); return -1; }
  for (int t = 0; t < 4; ++t) { ids[t] = got[t]; pts[t] = V3{xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]}; }
  return status;
}

constexpr int kAttemptFound = 0, kAttemptNoTriangle = 1, kAttemptRetry = 2;
// One attempt of SelectQuadrilateral's loop on the host structures, given its draws (match4pcsBase.cc:283-349)
int eval_attempt_host(const s4p_matcher* m, const uint32_t idx[2001], int ids[4], float& inv1, float& inv2) {
  const float kBaseTooSmall = 0.2f;
  int b1, b2, b3;
  if (!eval_triangle(m, idx, b1, b2, b3)) return kAttemptNoTriangle;
  const V3 A = m->P(b1), B = m->P(b2), C = m->P(b3);
  const double x1 = A.x, y1 = A.y, z1 = A.z, x2 = B.x, y2 = B.y, z2 = B.z, x3 = C.x, y3 = C.y, z3 = C.z;
  const float denom = float(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
  if (!(denom != 0)) return kAttemptRetry;
  const float pa = float((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
  const float pb = float((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
  const float pc = float((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
  const float too_small = float(std::pow(double(m->max_base_diameter * kBaseTooSmall), 2));
  const float pA[3] = {A.x, A.y, A.z}, pB[3] = {B.x, B.y, B.z}, pC[3] = {C.x, C.y, C.z};
  static thread_local std::vector<float> scratch;           // (two evaluator threads query the same index concurrently)
  const int b4 = m->fourth.query(pa, pb, pc, pA, pB, pC, too_small, scratch);
  if (b4 == -1) return kAttemptRetry;
  ids[0] = b1; ids[1] = b2; ids[2] = b3; ids[3] = b4;
  const V3 pts[4] = {A, B, C, m->P(b4)};
  return order_quadrilateral(pts, ids, inv1, inv2) ? kAttemptFound : kAttemptRetry;
}

// match4pcsBase.cc:279-351
bool select_quadrilateral(s4p_matcher* m, float& inv1, float& inv2, int ids[4]) {
  const float kBaseTooSmall = 0.2f;
  for (int attempt = 0; attempt < 1000; ++attempt) {
    if (m->device_select) {
      V3 pts[4];
      const int st = device_attempt(m, ids, pts);
      if (st < 0) { m->select_failed.store(true, std::memory_order_release); return false; }
      if (st == 1) return false;                    // SelectRandomTriangle failed: the reference gives up (:285-287)
      if (st == 0 && order_quadrilateral(pts, ids, inv1, inv2)) return true;
      continue;
    }
    int b1, b2, b3;
    if (!pick_triangle(m, b1, b2, b3)) return false;
    const V3 A = m->P(b1), B = m->P(b2), C = m->P(b3);
    const double x1 = A.x, y1 = A.y, z1 = A.z, x2 = B.x, y2 = B.y, z2 = B.z, x3 = C.x, y3 = C.y, z3 = C.z;
    const float denom = float(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
    if (denom != 0) {
      const float pa = float((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
      const float pb = float((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
      const float pc = float((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
      int b4 = -1;
      const float too_small = float(std::pow(double(m->max_base_diameter * kBaseTooSmall), 2));
      // Reference loop (match4pcsBase.cc:324-338): among points not closer than too_small (squared) to the three
      // base points, the first one with the strictly smallest plane distance |A x + B y + C z - 1|.
      // The reference evaluates the distance in double, float(|double(v) - 1.0|); for float v that equals the
      // correctly rounded float |v - 1.0f| (the double difference is exact), so the search runs in float, and through
      // FourthPointIndex, which skips the blocks of P whose bounding box lies further from the plane than the best
      // point found so far (s4p_host_structs.hpp).
      const float pA[3] = {A.x, A.y, A.z}, pB[3] = {B.x, B.y, B.z}, pC[3] = {C.x, C.y, C.z};
      static thread_local std::vector<float> scratch;
      b4 = m->fourth.query(pa, pb, pc, pA, pB, pC, too_small, scratch);
      if (b4 != -1) {
        ids[0] = b1; ids[1] = b2; ids[2] = b3; ids[3] = b4;
        const V3 pts[4] = {A, B, C, m->P(b4)};
        if (order_quadrilateral(pts, ids, inv1, inv2)) return true;
      }
    }
  }
  return false;
}

void fill_base_arrays(const s4p_matcher* m, const int ids[4], float* xyz, float* nrm, float* rgb) {
  for (int t = 0; t < 4; ++t) {
    const int i = ids[t];
    xyz[3 * t] = m->Ps.x[i]; xyz[3 * t + 1] = m->Ps.y[i]; xyz[3 * t + 2] = m->Ps.z[i];
    if (m->Ps.has_n) { nrm[3 * t] = m->Ps.nx[i]; nrm[3 * t + 1] = m->Ps.ny[i]; nrm[3 * t + 2] = m->Ps.nz[i]; }
    else { nrm[3 * t] = nrm[3 * t + 1] = nrm[3 * t + 2] = 0.f; }
    if (m->Ps.has_c) { rgb[3 * t] = m->Ps.r[i]; rgb[3 * t + 1] = m->Ps.g[i]; rgb[3 * t + 2] = m->Ps.b[i]; }
    else { rgb[3 * t] = rgb[3 * t + 1] = rgb[3 * t + 2] = -1.f; }
  }
}

// match4pcsBase.hpp:224-229.  computeRotationScaling(rot, scale) of a rigid [R|t] returns
// rot*scale == R up to SVD round-off; the linear part is used directly (DESIGN.md deviation D4).
void global_transform(const s4p_matcher* m, float* M) {
  std::memcpy(M, m->transform, sizeof(float) * 16);
  const float a[3] = {m->qc2[0] + m->centroid_q[0], m->qc2[1] + m->centroid_q[1], m->qc2[2] + m->centroid_q[2]};
  for (int r = 0; r < 3; ++r) {
    const float ra = m->transform[4 * r] * a[0] + (m->transform[4 * r + 1] * a[1] + m->transform[4 * r + 2] * a[2]);
    M[4 * r + 3] = (m->qc1[r] + m->centroid_p[r]) - ra;
  }
  M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
}

// ---------------------------------------------------------------------------------------------------------
// producer threads
inline void spin_until_ready(const std::atomic<size_t>& ready, const std::atomic<bool>& stop, int micros) {
  if (ready.load(std::memory_order_acquire) != 0) return;
  const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(micros);
  while (ready.load(std::memory_order_acquire) == 0 && !stop.load(std::memory_order_relaxed) && std::chrono::steady_clock::now() < until)
    for (int k = 0; k < 16; ++k) __builtin_ia32_pause();
}

// waits until `st` holds `want`; false if the producer is stopping.  Spins first (a streaming stage never sleeps), then
// naps: when the GPU is the bottleneck the queues are full and the helpers idle cheaply.
inline bool wait_state(const std::atomic<uint64_t>& st, uint64_t want, const std::atomic<bool>& stop) {
  for (int spin = 0;; ++spin) {
    if (st.load(std::memory_order_acquire) == want) return true;
    if (stop.load(std::memory_order_relaxed)) return false;
    if (spin < 4000) __builtin_ia32_pause();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

void drawer_main(s4p_matcher* m) {
  auto& P = m->prod;
  const uint32_t n = uint32_t(m->Ps.size());
  for (uint64_t seq = P.drawn.load();; ++seq) {
    const uint32_t slot = uint32_t(seq % s4p_matcher::Producer::kRing);
    if (!wait_state(P.ring_state[slot], s4p_matcher::Producer::tag(seq, 0u), P.stop_flag)) return;
    s4p_matcher::Producer::Attempt& a = P.ring[slot];
    const auto t0 = std::chrono::steady_clock::now();
    a.rng_before = m->rng;
    a.status = -1;
    if (n) draw_attempt(m->rng, n, a.idx);
    P.draw_ns.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()), std::memory_order_relaxed);
    P.ring_state[slot].store(s4p_matcher::Producer::tag(seq, 1u), std::memory_order_release);
    P.drawn.store(seq + 1, std::memory_order_release);
  }
}

void evaluator_main(s4p_matcher* m) {
  auto& P = m->prod;
  using clk = std::chrono::steady_clock;
  const float kBaseTooSmall = 0.2f;
  if (!m->device_select) {
    while (true) {
      const uint64_t seq = P.claimed.fetch_add(1);
      const uint32_t slot = uint32_t(seq % s4p_matcher::Producer::kRing);
      if (!wait_state(P.ring_state[slot], s4p_matcher::Producer::tag(seq, 1u), P.stop_flag)) return;
      s4p_matcher::Producer::Attempt& a = P.ring[slot];
      const auto t0 = clk::now();
      a.status = m->Ps.size() ? eval_attempt_host(m, a.idx, a.ids, a.inv1, a.inv2) : kAttemptNoTriangle;
      P.select_ns.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count()));
      P.batches.fetch_add(1, std::memory_order_relaxed); P.attempts_done.fetch_add(1, std::memory_order_relaxed);
      P.ring_state[slot].store(s4p_matcher::Producer::tag(seq, 2u), std::memory_order_release);
    }
  }
  // searches on the device: every attempt that has been drawn so far (up to the batch size) goes into ONE set of launches
  const int bmax = s4p_select_batch_max();                   // one scan of P serves the whole batch (k_select_fourth)
  std::vector<uint32_t> draws(size_t(bmax) * 2001);
  std::vector<int32_t> got(size_t(bmax) * 4), st(static_cast<size_t>(bmax));
  std::vector<float> xyz(size_t(bmax) * 12);
  const float limit = m->max_base_diameter * m->max_base_diameter;
  const float too_small = float(std::pow(double(m->max_base_diameter * kBaseTooSmall), 2));
  while (true) {
    const uint64_t seq0 = P.claimed.load();
    if (!wait_state(P.ring_state[seq0 % s4p_matcher::Producer::kRing], s4p_matcher::Producer::tag(seq0, 1u), P.stop_flag)) return;
    int nb = 1;
    while (nb < bmax && P.ring_state[(seq0 + uint64_t(nb)) % s4p_matcher::Producer::kRing].load(std::memory_order_acquire) == s4p_matcher::Producer::tag(seq0 + uint64_t(nb), 1u)) ++nb;
    for (int k = 0; k < nb; ++k) std::memcpy(draws.data() + size_t(k) * 2001, P.ring[(seq0 + uint64_t(k)) % s4p_matcher::Producer::kRing].idx, 2001 * sizeof(uint32_t));
    const auto t0 = clk::now();
    const int32_t rc = m->Ps.size() ? s4p_select_base_points_batch(m->ctx, draws.data(), nb, limit, too_small, got.data(), xyz.data(), st.data()) : S4P_OK;
    if (rc != S4P_OK) { m->select_err = s4p_last_error(m->ctx); m->select_failed.store(true, std::memory_order_release); }
    for (int k = 0; k < nb; ++k) {
      s4p_matcher::Producer::Attempt& a = P.ring[(seq0 + uint64_t(k)) % s4p_matcher::Producer::kRing];
      a.status = kAttemptNoTriangle;                         // (also the answer after a device error: the trial finds no base, the loop reports the error)
      if (rc == S4P_OK && m->Ps.size()) {
        if (st[size_t(k)] == 1) a.status = kAttemptNoTriangle;
        else if (st[size_t(k)] == 0) {
          V3 pts[4];
          for (int t = 0; t < 4; ++t) { a.ids[t] = got[size_t(4 * k + t)]; pts[t] = V3{xyz[size_t(12 * k + 3 * t)], xyz[size_t(12 * k + 3 * t + 1)], xyz[size_t(12 * k + 3 * t + 2)]}; }
          a.status = order_quadrilateral(pts, a.ids, a.inv1, a.inv2) ? kAttemptFound : kAttemptRetry;
        } else a.status = kAttemptRetry;
      }
    }
    P.select_ns.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count()));
    P.batches.fetch_add(1, std::memory_order_relaxed); P.attempts_done.fetch_add(uint64_t(nb), std::memory_order_relaxed);
    for (int k = 0; k < nb; ++k) P.ring_state[(seq0 + uint64_t(k)) % s4p_matcher::Producer::kRing].store(s4p_matcher::Producer::tag(seq0 + uint64_t(k), 2u), std::memory_order_release);
    P.claimed.store(seq0 + uint64_t(nb));
  }
}

// attempts -> trials, strictly in order: SelectQuadrilateral's loop (match4pcsBase.cc:283-349) over evaluated attempts
void assembler_main(s4p_matcher* m) {
  auto& P = m->prod;
  while (true) {
    { std::unique_lock<std::mutex> lk(P.mu);
      P.cv_a_space.wait(lk, [&] { return P.stop || P.qa.size() < P.cap_a; });
      if (P.stop) return; }
    s4p_matcher::Trial t;
    t.found = false;
    bool first = true;
    for (int attempt = 0; attempt < 1000; ++attempt) {
      const uint64_t seq = P.taken.load();
      const uint32_t slot = uint32_t(seq % s4p_matcher::Producer::kRing);
      if (!wait_state(P.ring_state[slot], s4p_matcher::Producer::tag(seq, 2u), P.stop_flag)) return;      // (P.partial tells producer_stop where this trial's draws began)
      s4p_matcher::Producer::Attempt& a = P.ring[slot];
      if (first) { t.rng_before = a.rng_before; P.partial_rng = a.rng_before; P.partial = true; first = false; }
      const int status = a.status;
      if (status == kAttemptFound) { t.found = true; t.inv1 = a.inv1; t.inv2 = a.inv2; for (int k = 0; k < 4; ++k) t.ids[k] = a.ids[k]; }
      P.ring_state[slot].store(s4p_matcher::Producer::tag(seq + s4p_matcher::Producer::kRing, 0u), std::memory_order_release);      // free for the next lap
      P.taken.store(seq + 1, std::memory_order_release);
      if (status != kAttemptRetry) break;                    // found, or SelectRandomTriangle failed: the reference gives up (:285-287)
    }
    if (t.found) fill_base_arrays(m, t.ids, t.bx, t.bn, t.bc);
    { std::lock_guard<std::mutex> lk(P.mu);
      P.partial = false;
      t.index = P.next_index++;
      t.owned = (t.index % P.world) == P.rank;
      P.qa.push_back(std::move(t));       // pushed even when stopping: nothing that advanced the RNG is ever dropped
      P.qa_ready.store(P.qa.size(), std::memory_order_release); }
    P.cv_a_item.notify_one();
  }
}

void tree_main(s4p_matcher* m) {
  auto& P = m->prod;
  while (true) {
    s4p_matcher::Trial t;
    bool wake_selector = false;
    spin_until_ready(P.qa_ready, P.stop_flag, s4p_matcher::Producer::kSpinMicros);
    { std::unique_lock<std::mutex> lk(P.mu);
      P.cv_a_item.wait(lk, [&] { return P.stop || !P.qa.empty(); });
      if (P.stop) return;
      t = std::move(P.qa.front()); P.qa.pop_front();
      P.qa_ready.store(P.qa.size(), std::memory_order_release);
      wake_selector = P.qa.size() == P.cap_a / 2; }
    if (wake_selector) P.cv_a_space.notify_one();
    const auto tree_t0 = std::chrono::steady_clock::now();
    if (t.found && t.owned) {
      std::unique_lock<std::mutex> lk(P.mu);
      P.cv_slot.wait(lk, [&] { return P.stop || !P.free_slots.empty(); });
      if (P.stop) { P.qa.push_front(std::move(t)); return; }     // untouched: goes back to the head of qa
      t.slot = P.free_slots.back(); P.free_slots.pop_back();
    }
    t.pair_before.resize(size_t(s4p_pair_state_words(m->ctx)));
    s4p_pair_state_save(m->ctx, t.pair_before.data());
    if (t.found) (void)s4p_stage_base(m->ctx, t.bx, t.bn, t.owned ? 1 : 0, t.owned ? t.slot : 0);
    t.staged = true;
    P.tree_ns.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tree_t0).count()), std::memory_order_relaxed);
    P.trials_done.fetch_add(1, std::memory_order_relaxed);
    { std::unique_lock<std::mutex> lk(P.mu);
      P.cv_b_space.wait(lk, [&] { return P.stop || P.qb.size() < P.cap_b; });
      P.qb.push_back(std::move(t));       // also when stopping: its octree effect is already applied
      P.qb_ready.store(P.qb.size(), std::memory_order_release); }
    P.cv_b_item.notify_one();
  }
}

void producer_stop(s4p_matcher* m);

// Where the helper threads pay.  The host chain of a trial is ~20 us with the host search structures (n_P < 2^20): run inline
// on the launch thread it costs a 1-GPU job nothing (the GPU step is ~115 us) and a hand-off through four queues costs more
// than the chain -- measured on the bench workload, one GPU playing rank 0 of a world of 1 / 2 / 3 / 4 / 6 / 8
// (tools/sim_world.py --threads-ab): 0.135 / 0.137 / 0.133 / 0.144 / 0.157 / 0.197 ms per window inline against
// 0.139 / 0.138 / 0.132 / 0.141 / 0.130 / 0.120 with the threads; a short call also pays ~1 ms for starting them.  With the
// searches on the device (n_P >= 2^20) an attempt is a synchronous round trip unless the evaluator thread batches them:
// 0.139 -> 0.051 ms per trial at n_P = 4.2 M.
void update_helper_threads(s4p_matcher* m) {
  auto& P = m->prod;
  const bool want = P.mode == 1 || (P.mode == 2 && (P.world >= 4 || m->device_select));
  if (want != P.enabled) { producer_stop(m); P.enabled = want; }
}

void producer_start(s4p_matcher* m) {
  auto& P = m->prod;
  if (P.running) return;
  P.stop = false; P.stop_flag.store(false);
  P.free_slots.clear();
  const int nslots = s4p_stage_slots(m->ctx);
  for (int sl = nslots / 2; sl < nslots; ++sl) P.free_slots.push_back(sl);      // the lower half belongs to s4p_try_base_async
  P.next_index = P.consumed;
  if (!P.ring) P.ring.reset(new s4p_matcher::Producer::Attempt[s4p_matcher::Producer::kRing]);
  for (auto& st : P.ring_state) st.store(0ull);
  P.drawn.store(0); P.claimed.store(0); P.taken.store(0); P.partial = false;
  P.draw_ns.store(0); P.tree_ns.store(0); P.batches.store(0); P.attempts_done.store(0); P.trials_done.store(0);
  P.started = std::chrono::steady_clock::now();
  P.n_eval = m->device_select ? 1 : 2;
  P.drawer = std::thread(drawer_main, m);
  for (int k = 0; k < P.n_eval; ++k) P.evals[k] = std::thread(evaluator_main, m);
  P.assembler = std::thread(assembler_main, m);
  P.tree = std::thread(tree_main, m);
  P.running = true;
}

// Stops the helpers and rewinds RNG + octree permutation to just before the first trial the main thread has not
// consumed, i.e. to where a sequential run would be.
void producer_stop(s4p_matcher* m) {
  auto& P = m->prod;
  if (!P.running) return;
  { std::lock_guard<std::mutex> lk(P.mu); P.stop = true; P.stop_flag.store(true); }
  P.wake_all();
  P.drawer.join();
  for (int k = 0; k < P.n_eval; ++k) P.evals[k].join();
  P.assembler.join(); P.tree.join();
  P.running = false;
  if (std::getenv("S4P_TRACE_CHAIN")) {
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - P.started).count();
    std::fprintf(stderr, "{\"s4p_trace\": \"host chain\", \"world\": %d, \"device_selection\": %d, \"wall_ms\": %.3f, \"attempts_drawn\": %llu, "
                 "\"attempts_evaluated\": %llu, \"evaluator_calls\": %llu, \"trials_staged\": %llu, \"trials_consumed\": %ld, \"drawer_busy_ms\": %.3f, "
                 "\"evaluators_busy_ms\": %.3f, \"tree_busy_ms\": %.3f}\n", P.world, m->device_select ? 1 : 0, wall_ms,
                 (unsigned long long)P.drawn.load(), (unsigned long long)P.attempts_done.load(), (unsigned long long)P.batches.load(),
                 (unsigned long long)P.trials_done.load(), P.consumed, double(P.draw_ns.load()) * 1e-6, double(P.select_ns.load()) * 1e-6,
                 double(P.tree_ns.load()) * 1e-6);
  }
  // the random stream goes back to where the first trial that was not handed to the main thread began
  if (!P.qb.empty()) {
    m->rng = P.qb.front().rng_before;
    s4p_pair_state_restore(m->ctx, P.qb.front().pair_before.data());
  } else if (!P.qa.empty()) {
    m->rng = P.qa.front().rng_before;
  } else if (P.partial) {
    m->rng = P.partial_rng;                                  // the assembler was inside a trial
  } else if (P.taken.load() < P.drawn.load()) {
    m->rng = P.ring[P.taken.load() % s4p_matcher::Producer::kRing].rng_before;      // attempts drawn ahead, none of them consumed
  }
  P.partial = false;
  P.qa.clear(); P.qb.clear(); P.qa_ready.store(0); P.qb_ready.store(0);
  m->seconds_select += double(P.select_ns.exchange(0)) * 1e-9 + P.select_s; P.select_s = 0;
}

bool producer_pop(s4p_matcher* m, s4p_matcher::Trial& t) {
  auto& P = m->prod;
  producer_start(m);
  spin_until_ready(P.qb_ready, P.stop_flag, s4p_matcher::Producer::kSpinMicros);
  bool wake_tree = false;
  { std::unique_lock<std::mutex> lk(P.mu);
    P.cv_b_item.wait(lk, [&] { return !P.qb.empty(); });
    t = std::move(P.qb.front()); P.qb.pop_front();
    P.qb_ready.store(P.qb.size(), std::memory_order_release);
    wake_tree = P.qb.size() == P.cap_b / 2;
    P.consumed = t.index + 1; }
  if (wake_tree) P.cv_b_space.notify_one();
  return true;
}

void producer_release_slot(s4p_matcher* m, int slot) {
  if (slot < 0) return;
  { std::lock_guard<std::mutex> lk(m->prod.mu); m->prod.free_slots.push_back(slot); }
  m->prod.cv_slot.notify_one();
}

// first half of TryOneBase (match4pcsBase.hpp:281-351): base selection + device pass (or state advance only)
int32_t next_base_async(s4p_matcher* m, bool run_device, bool snapshot, s4p_matcher::Prepared& pr);
int32_t wait_base(s4p_matcher* m, const s4p_matcher::Prepared& pr, s4p_base_result& r);

// The helper threads decide who owns a trial from (rank, world).  A caller that skips trials by its own rule (run_device = 0 on a
// matcher that was never declared part of a sharded job) cannot be served by threads the engine turned on by itself: they
// are stopped -- which rewinds RNG and octree permutation to this trial -- and stay off for this matcher.
void caller_owns_trials(s4p_matcher* m, bool run_device) {
  auto& P = m->prod;
  if (P.enabled && P.mode == 2 && P.world == 1 && !run_device) { producer_stop(m); P.mode = 0; P.enabled = false; }
}

int32_t next_base(s4p_matcher* m, bool run_device, bool& found, int ids[4], s4p_base_result& r) {
  using clk = std::chrono::steady_clock;
  float inv1 = 0, inv2 = 0;
  std::memset(&r, 0, sizeof(r));
  caller_owns_trials(m, run_device);
  if (m->prod.enabled) {                 // same thing through the producer queues
    s4p_matcher::Prepared pr;
    if (int32_t rc = next_base_async(m, run_device, false, pr)) return rc;
    found = pr.found;
    for (int t = 0; t < 4; ++t) ids[t] = pr.ids[t];
    return wait_base(m, pr, r);
  }
  auto t0 = clk::now();
  found = select_quadrilateral(m, inv1, inv2, ids);
  m->seconds_select += std::chrono::duration<double>(clk::now() - t0).count();
  if (m->select_failed.load(std::memory_order_acquire)) return m->fail(S4P_ERR_HIP, "device base selection failed: " + m->select_err);
  if (!found) return S4P_OK;                                     // :313-316
  float bx[12], bn[12], bc[12];
  fill_base_arrays(m, ids, bx, bn, bc);
  if (int32_t rc = s4p_set_base(m->ctx, bx, bn, bc)) return m->ctx_fail(rc);
  t0 = clk::now();
  if (run_device) {
    if (int32_t rc = s4p_try_base(m->ctx, ids, inv1, inv2, &r)) return m->ctx_fail(rc);
    m->bases_tried++;
    m->pairs_total += r.n_pairs1 + r.n_pairs2; m->quads_total += r.n_quads; m->candidates_verified += r.n_verified;
  } else {
    if (int32_t rc = s4p_skip_base(m->ctx)) return m->ctx_fail(rc);
  }
  m->seconds_device += std::chrono::duration<double>(clk::now() - t0).count();
  return S4P_OK;
}

// pipelined first half: enqueue the device pass and return; results come back through wait_base()
int32_t next_base_async(s4p_matcher* m, bool run_device, bool snapshot, s4p_matcher::Prepared& pr) {
  using clk = std::chrono::steady_clock;
  float inv1 = 0, inv2 = 0;
  caller_owns_trials(m, run_device);
  if (m->prod.enabled) {
    s4p_matcher::Trial t;
    producer_pop(m, t);
    if (m->select_failed.load(std::memory_order_acquire)) { if (t.slot >= 0) producer_release_slot(m, t.slot); return m->fail(S4P_ERR_HIP, "device base selection failed: " + m->select_err); }
    if (t.owned != run_device) return m->fail(S4P_ERR_STATE, "sharding mismatch: the producer and the caller disagree on who owns this trial");
    pr.found = t.found; pr.device = false; pr.slot = -1;
    for (int k = 0; k < 4; ++k) pr.ids[k] = t.ids[k];
    pr.rng_before = t.rng_before; pr.pair_state_before = std::move(t.pair_before);
    if (t.found && t.owned) {
      const auto t0 = clk::now();
      if (int32_t rc = s4p_set_base(m->ctx, t.bx, t.bn, t.bc)) return m->ctx_fail(rc);
      if (int32_t rc = s4p_try_base_staged_async(m->ctx, t.slot, t.ids, t.inv1, t.inv2)) { producer_release_slot(m, t.slot); return m->ctx_fail(rc); }
      m->seconds_device += std::chrono::duration<double>(clk::now() - t0).count();
      pr.device = true; pr.slot = t.slot;
    }
    return S4P_OK;
  }
  if (snapshot) {
    pr.rng_before = m->rng;
    pr.pair_state_before.resize(size_t(s4p_pair_state_words(m->ctx)));
    s4p_pair_state_save(m->ctx, pr.pair_state_before.data());
  }
  auto t0 = clk::now();
  pr.found = select_quadrilateral(m, inv1, inv2, pr.ids);
  m->seconds_select += std::chrono::duration<double>(clk::now() - t0).count();
  if (m->select_failed.load(std::memory_order_acquire)) return m->fail(S4P_ERR_HIP, "device base selection failed: " + m->select_err);
  pr.device = false;
  if (!pr.found) return S4P_OK;
  float bx[12], bn[12], bc[12];
  fill_base_arrays(m, pr.ids, bx, bn, bc);
  if (int32_t rc = s4p_set_base(m->ctx, bx, bn, bc)) return m->ctx_fail(rc);
  t0 = clk::now();
  if (run_device) {
    if (int32_t rc = s4p_try_base_async(m->ctx, pr.ids, inv1, inv2)) return m->ctx_fail(rc);
    pr.device = true;
  } else {
    if (int32_t rc = s4p_skip_base(m->ctx)) return m->ctx_fail(rc);
  }
  m->seconds_device += std::chrono::duration<double>(clk::now() - t0).count();
  return S4P_OK;
}

int32_t wait_base(s4p_matcher* m, const s4p_matcher::Prepared& pr, s4p_base_result& r) {
  using clk = std::chrono::steady_clock;
  std::memset(&r, 0, sizeof(r));
  if (!pr.device) return S4P_OK;
  const auto t0 = clk::now();
  const int32_t wrc = s4p_try_base_wait(m->ctx, &r);
  producer_release_slot(m, pr.slot);
  if (wrc) return m->ctx_fail(wrc);
  m->seconds_device += std::chrono::duration<double>(clk::now() - t0).count();
  m->bases_tried++;
  m->pairs_total += r.n_pairs1 + r.n_pairs2; m->quads_total += r.n_quads; m->candidates_verified += r.n_verified;
  return S4P_OK;
}

// second half: best update + TryOneBase's return value
bool commit_base(s4p_matcher* m, bool found, const int ids[4], const s4p_base_result& r) {
  if (!found) return false;
  if (r.n_pairs1 == 0 || r.n_pairs2 == 0) return false;         // :335-337
  if (r.n_quads == 0) return false;                              // :340-347
  if (r.has_best) {
    const float lcp = float(r.best_count) / float(m->Qs.size()); // Verify's return value, .cc:566
    if (lcp > m->best_lcp) {                                     // :467-484 (first strictly greater wins)
      for (int t = 0; t < 4; ++t) { m->base[t] = ids[t]; m->congruent[t] = r.best_quad[t]; }
      m->best_lcp = lcp; m->best_count = r.best_count;
      std::memcpy(m->transform, r.best_transform, sizeof(float) * 16);
      for (int k = 0; k < 3; ++k) { m->qc1[k] = r.centroid1[k]; m->qc2[k] = r.best_centroid2[k]; }
      if (m->in_loop) (void)s4p_set_best_hint(m->ctx, m->best_count);          // bases launched from now on may abandon what cannot beat this
    }
  }
  return m->best_lcp > m->opt.terminate_threshold;               // :496
}

int32_t try_one_base(s4p_matcher* m, bool& ok, s4p_base_result* last) {
  ok = false;
  bool found = false; int ids[4] = {0, 0, 0, 0}; s4p_base_result r;
  if (int32_t rc = next_base(m, true, found, ids, r)) return rc;
  if (last) *last = r;
  ok = commit_base(m, found, ids, r);
  return S4P_OK;
}

bool view_ok(const s4p_cloud_view* v) { return v && v->x && v->y && v->z && v->n >= 0; }

// S4P_TRACE_INIT=1 (lab aid, like S4P_TRACE_CALL): where init spends its time, one line on stderr per call
struct InitTrace {
  using clk = std::chrono::steady_clock;
  const bool on = std::getenv("S4P_TRACE_INIT") != nullptr;
  clk::time_point t = clk::now();
  std::string line;
  void lap(const char* what) {
    if (!on) return;
    const auto now = clk::now();
    char buf[96];
    std::snprintf(buf, sizeof buf, "%s\"%s_ms\": %.3f", line.empty() ? "" : ", ", what, std::chrono::duration<double, std::milli>(now - t).count());
    line += buf; t = now;
  }
  void print(const char* tag) const { if (on) std::fprintf(stderr, "{\"s4p_trace\": \"%s\", %s}\n", tag, line.c_str()); }
};

// Bases started with s4p_matcher_next_base_async and never waited for: their kernels still read the lane buffers, and
// the context's FIFO must stay in step with m->inflight.  Called before anything that re-initialises or restarts.
void drain_inflight(s4p_matcher* m) {
  for (auto& pr : m->inflight) {
    if (!pr.device) continue;
    s4p_base_result dummy;
    (void)s4p_try_base_wait(m->ctx, &dummy);
    producer_release_slot(m, pr.slot);
  }
  m->inflight.clear();
}

}  // namespace

extern "C" {

int32_t s4p_matcher_create(const s4p_options* opt, const s4p_limits* lim, int32_t device, s4p_matcher** out) {
  if (!opt || !out) return S4P_ERR_BAD_ARG;
  *out = nullptr;
  s4p_ctx* ctx = nullptr;
  if (int32_t rc = s4p_create(opt, lim, device, &ctx)) return rc;
  s4p_matcher* m = new s4p_matcher();
  m->opt = *opt; m->ctx = ctx; m->device = device; m->rng.seed(opt->random_seed);
  m->set_identity();
  *out = m;
  return S4P_OK;
}

void s4p_matcher_destroy(s4p_matcher* m) {
  if (!m) return;
  producer_stop(m);
  s4p_destroy(m->ctx);
  delete m;
}

const char* s4p_matcher_last_error(const s4p_matcher* m) { return m ? m->err.c_str() : s4p_last_error(nullptr); }
s4p_ctx* s4p_matcher_ctx(s4p_matcher* m) { return m ? m->ctx : nullptr; }
float s4p_matcher_terminate_threshold(const s4p_matcher* m) { return m ? m->opt.terminate_threshold : 0.f; }
int32_t s4p_matcher_max_time_seconds(const s4p_matcher* m) { return m ? m->opt.max_time_seconds : 0; }
int64_t s4p_matcher_init_generation(const s4p_matcher* m) { return m ? m->init_generation : -1; }

int64_t s4p_uniform_dist_sample(const float* x, const float* y, const float* z, int64_t n, float delta, int64_t* out) {
  if (!x || !y || !z || !out || n <= 0 || !(delta > 0.f)) return 0;
  return voxel_first_hits(x, y, z, n, delta, out);
}

int32_t s4p_matcher_init(s4p_matcher* m, const s4p_cloud_view* p, const s4p_cloud_view* q, int32_t q_needs_shuffle) {
  if (!m) return S4P_ERR_BAD_ARG;
  if (!view_ok(p) || !view_ok(q) || p->n == 0 || q->n == 0) return m->fail(S4P_ERR_BAD_ARG, "s4p_matcher_init: empty or null cloud");
  producer_stop(m);
  drain_inflight(m);
  m->prod.consumed = 0; m->prod.next_index = 0;
  m->ready = false; m->init_generation++;
  InitTrace trace;
  Cloud& Ps = m->Ps; Cloud& Qs = m->Qs;
  Ps = Cloud(); Qs = Cloud();
  Ps.init_flags(*p); Qs.init_flags(*q);
  Ps.reserve(size_t(p->n));
  for (int64_t i = 0; i < p->n; ++i) Ps.push_from(*p, i);
  if (q_needs_shuffle) {                                           // match4pcsBase.hpp:129-133
    std::vector<uint32_t> perm(size_t(q->n));
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = uint32_t(i);
    std::shuffle(perm.begin(), perm.end(), m->rng);                // identical swap sequence to shuffling the points
    const size_t keep = std::min<size_t>(perm.size(), size_t(m->opt.sample_size));
    Qs.reserve(keep);
    for (size_t i = 0; i < keep; ++i) Qs.push_from(*q, int64_t(perm[i]));
  } else {
    Qs.reserve(size_t(q->n));
    for (int64_t i = 0; i < q->n; ++i) Qs.push_from(*q, i);
  }
  trace.lap("copy_in_and_shuffle_q");
  centre_cloud(Ps, m->centroid_p);
  centre_cloud(Qs, m->centroid_q);
  trace.lap("centre");
  // P_diameter_: 1000 random pair distances in the sampled Q (quirk), match4pcsBase.hpp:155-164
  m->p_diameter = 0.f;
  const unsigned long nq = (unsigned long)Qs.size();
  for (int t = 0; t < 1000; ++t) {
    const int at = int(m->rng() % nq);
    const int bt = int(m->rng() % nq);
    const float l = len(sub(V3{Qs.x[bt], Qs.y[bt], Qs.z[bt]}, V3{Qs.x[at], Qs.y[at], Qs.z[at]}));
    if (l > m->p_diameter) m->p_diameter = l;
  }
  // MeanDistance() (match4pcsBase.cc:158-182) only fills P_mean_distance_, which nothing reads; it draws
  // no random numbers, so it is skipped here.
  m->max_base_diameter = m->p_diameter;                            // :172
  const float kSmallError = 0.00001f;                               // :175-185
  const float first_estimation = float(std::log(kSmallError) /
      std::log(1.0 - std::pow(double(m->opt.overlap_estimation), double(4.0f))));
  m->number_of_trials = int(first_estimation * (m->p_diameter / 0.3f) / m->max_base_diameter);
  if (m->number_of_trials < 4) m->number_of_trials = 4;
  m->current_trial = 0;
  m->best_lcp = 0.f; m->best_count = 0;
  for (int t = 0; t < 4; ++t) { m->base[t] = 0; m->congruent[t] = 0; }
  m->set_identity();
  // SelectQuadrilateral's searches: device reductions for large sampled P (or when asked for), else the host-side
  // search structures, built while the device builds its own
  if (const char* e = std::getenv("S4P_DEVICE_SELECT")) { if (m->select_mode < 0) m->select_mode = std::atoi(e) != 0 ? 1 : 0; }
  m->device_select = m->select_mode >= 0 ? m->select_mode != 0 : Ps.size() >= kDeviceSelectMin;
  update_helper_threads(m);
  m->select_failed.store(false);
  std::thread host_index([m, &Ps] {
    if (m->device_select) { m->P4.clear(); m->P4.shrink_to_fit(); m->fourth = s4p::FourthPointIndex(); return; }
    const size_t np = Ps.size();
    m->P4.resize(4 * np);
    for (size_t i = 0; i < np; ++i) { m->P4[4 * i] = Ps.x[i]; m->P4[4 * i + 1] = Ps.y[i]; m->P4[4 * i + 2] = Ps.z[i]; m->P4[4 * i + 3] = 0.f; }
    m->fourth.build(Ps.x.data(), Ps.y.data(), Ps.z.data(), np);
  });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } join_host_index{host_index};
  // Initialize() -> device structures; then best_LCP_ = Verify(identity)   (:199-201)
  if (int32_t rc = s4p_set_clouds(m->ctx, Ps.x.data(), Ps.y.data(), Ps.z.data(), int64_t(Ps.size()),
                                  Qs.x.data(), Qs.y.data(), Qs.z.data(),
                                  Qs.has_n ? Qs.nx.data() : nullptr, Qs.has_n ? Qs.ny.data() : nullptr, Qs.has_n ? Qs.nz.data() : nullptr,
                                  Qs.has_c ? Qs.r.data() : nullptr, Qs.has_c ? Qs.g.data() : nullptr, Qs.has_c ? Qs.b.data() : nullptr,
                                  int64_t(Qs.size())))
    return m->ctx_fail(rc);
  trace.lap("set_clouds");
  uint32_t c0 = 0;
  if (int32_t rc = s4p_verify_transforms(m->ctx, m->transform, 1, &c0)) return m->ctx_fail(rc);
  trace.lap("initial_lcp");
  trace.print("s4p_matcher_init");
  m->best_count = c0;
  m->best_lcp = float(c0) / float(Qs.size());
  m->ready = true;
  return S4P_OK;
}

int32_t s4p_matcher_init_full(s4p_matcher* m, const s4p_cloud_view* P, const s4p_cloud_view* Q) {
  if (!m) return S4P_ERR_BAD_ARG;
  if (!view_ok(P) || !view_ok(Q) || P->n == 0 || Q->n == 0) return m->fail(S4P_ERR_BAD_ARG, "empty or null cloud");
  bool sampler_failed = false;
  auto subset = [&](const s4p_cloud_view& v, bool sample, std::vector<std::vector<float>>& store) -> s4p_cloud_view {
    if (!sample) return v;                                          // "use whole cloud", match4pcsBase.hpp:115-119
    const std::unique_ptr<int64_t[]> idx(new int64_t[size_t(v.n)]);  // only the first k entries are written and read: no zero-fill pass
    const int64_t k = voxel_first_hits(v.x, v.y, v.z, v.n, m->opt.delta, idx.get(), m->device);
    if (k < 0) { sampler_failed = true; return v; }
    const float* src[9] = {v.x, v.y, v.z, v.nx, v.ny, v.nz, v.r, v.g, v.b};
    store.assign(9, {});
    const float* dst[9];
    for (int a = 0; a < 9; ++a) {
      if (!src[a]) { dst[a] = nullptr; continue; }
      store[a].resize(size_t(k));
      for (int64_t i = 0; i < k; ++i) store[a][size_t(i)] = src[a][idx[size_t(i)]];
      dst[a] = store[a].data();
    }
    return s4p_cloud_view{dst[0], dst[1], dst[2], dst[3], dst[4], dst[5], dst[6], dst[7], dst[8], k};
  };
  std::vector<std::vector<float>> sp, sq;
  const bool sample_p = uint64_t(P->n) > m->opt.sample_size;
  const bool sample_q = uint64_t(Q->n) > m->opt.sample_size;
  InitTrace trace;
  const s4p_cloud_view pv = subset(*P, sample_p, sp);
  trace.lap("sample_and_gather_p");
  const s4p_cloud_view qv = subset(*Q, sample_q, sq);
  trace.lap("sample_and_gather_q");
  trace.print("s4p_matcher_init_full");
  if (sampler_failed) return m->fail(S4P_ERR_HIP, "device UniformDistSampler failed (HIP error, see stderr); no silent host fallback");
  return s4p_matcher_init(m, &pv, &qv, sample_q ? 1 : 0);
}

int32_t s4p_matcher_get_info(s4p_matcher* m, s4p_matcher_info* o) {
  if (!m || !o) return S4P_ERR_BAD_ARG;
  std::memset(o, 0, sizeof(*o));
  o->number_of_trials = m->number_of_trials; o->current_trial = m->current_trial;
  o->n_sampled_p = int32_t(m->Ps.size()); o->n_sampled_q = int32_t(m->Qs.size());
  o->best_lcp = m->best_lcp; o->best_count = m->best_count; o->p_diameter = m->p_diameter;
  for (int k = 0; k < 3; ++k) { o->centroid_p[k] = m->centroid_p[k]; o->centroid_q[k] = m->centroid_q[k]; o->qcentroid1[k] = m->qc1[k]; o->qcentroid2[k] = m->qc2[k]; }
  std::memcpy(o->transform, m->transform, sizeof(float) * 16);
  for (int t = 0; t < 4; ++t) { o->base[t] = m->base[t]; o->congruent[t] = m->congruent[t]; }
  o->candidates_verified = m->candidates_verified; o->quads_total = m->quads_total; o->pairs_total = m->pairs_total;
  o->bases_tried = m->bases_tried; o->seconds_select = m->seconds_select; o->seconds_device = m->seconds_device;
  return S4P_OK;
}

int32_t s4p_matcher_get_sampled(s4p_matcher* m, int32_t which, float* x, float* y, float* z) {
  if (!m || !x || !y || !z) return S4P_ERR_BAD_ARG;
  const Cloud& c = which == 0 ? m->Ps : m->Qs;
  std::memcpy(x, c.x.data(), c.size() * 4); std::memcpy(y, c.y.data(), c.size() * 4); std::memcpy(z, c.z.data(), c.size() * 4);
  return S4P_OK;
}

int32_t s4p_matcher_get_sampled_attrs(s4p_matcher* m, int32_t which, float* nx, float* ny, float* nz, float* r, float* g, float* b,
                                      int32_t* has_normals, int32_t* has_rgb) {
  if (!m) return S4P_ERR_BAD_ARG;
  const Cloud& c = which == 0 ? m->Ps : m->Qs;
  const size_t n = c.size();
  auto put = [n](float* dst, const std::vector<float>& src, bool have, float dflt) {
    if (!dst) return;
    if (have) std::memcpy(dst, src.data(), n * 4); else for (size_t i = 0; i < n; ++i) dst[i] = dflt;
  };
  put(nx, c.nx, c.has_n, 0.f); put(ny, c.ny, c.has_n, 0.f); put(nz, c.nz, c.has_n, 0.f);
  put(r, c.r, c.has_c, -1.f); put(g, c.g, c.has_c, -1.f); put(b, c.b, c.has_c, -1.f);
  if (has_normals) *has_normals = c.has_n ? 1 : 0;
  if (has_rgb) *has_rgb = c.has_c ? 1 : 0;
  return S4P_OK;
}

int32_t s4p_matcher_select_quadrilateral(s4p_matcher* m, int32_t* found, float* inv1, float* inv2, int32_t* base_ids, float* base_xyz) {
  if (!m || !found || !inv1 || !inv2 || !base_ids) return S4P_ERR_BAD_ARG;
  if (!m->ready) return m->fail(S4P_ERR_STATE, "matcher not initialised");
  producer_stop(m);                       // (helper threads that ran ahead: back to the first trial nobody consumed)
  int ids[4] = {0, 0, 0, 0};
  *found = select_quadrilateral(m, *inv1, *inv2, ids) ? 1 : 0;
  if (m->select_failed.load(std::memory_order_acquire)) return m->fail(S4P_ERR_HIP, "device base selection failed: " + m->select_err);
  for (int t = 0; t < 4; ++t) base_ids[t] = ids[t];
  if (base_xyz && *found) { float bn[12], bc[12]; fill_base_arrays(m, ids, base_xyz, bn, bc); }
  return S4P_OK;
}

int32_t s4p_matcher_try_one_base(s4p_matcher* m, int32_t* ok, s4p_base_result* last) {
  if (!m || !ok) return S4P_ERR_BAD_ARG;
  if (!m->ready) return m->fail(S4P_ERR_STATE, "matcher not initialised");
  bool b = false;
  const int32_t rc = try_one_base(m, b, last);
  *ok = b ? 1 : 0;
  return rc;
}

int32_t s4p_matcher_next_base(s4p_matcher* m, int32_t run_device, int32_t* found, int32_t* base_ids, s4p_base_result* result) {
  if (!m || !found || !base_ids || !result) return S4P_ERR_BAD_ARG;
  if (!m->ready) return m->fail(S4P_ERR_STATE, "matcher not initialised");
  bool f = false; int ids[4] = {0, 0, 0, 0};
  const int32_t rc = next_base(m, run_device != 0, f, ids, *result);
  *found = f ? 1 : 0;
  for (int t = 0; t < 4; ++t) base_ids[t] = ids[t];
  return rc;
}

int32_t s4p_matcher_set_device_selection(s4p_matcher* m, int32_t mode) {
  if (!m || mode < -1 || mode > 1) return S4P_ERR_BAD_ARG;
  m->select_mode = mode;                  // takes effect at the next init
  return S4P_OK;
}

int32_t s4p_matcher_device_selection(const s4p_matcher* m) { return m && m->device_select ? 1 : 0; }

int32_t s4p_matcher_grow_on_overflow(s4p_matcher* m, int32_t enable) {
  if (!m) return S4P_ERR_BAD_ARG;
  m->grow_on_overflow = enable != 0;
  return s4p_set_auto_grow(m->ctx, enable);
}

int32_t s4p_matcher_capacity_growths(const s4p_matcher* m) { return m ? int32_t(s4p_lane_growths(m->ctx)) : 0; }

// Trial loops call this around themselves: inside, every commit refreshes the device's best-count hint.
static void loop_hint(s4p_matcher* m, bool enter) {
  static const bool env_off = std::getenv("S4P_EARLY_EXIT") && std::atoi(std::getenv("S4P_EARLY_EXIT")) == 0;
  m->in_loop = enter && m->early_exit && !env_off && !m->visit_candidates;
  (void)s4p_set_best_hint(m->ctx, m->in_loop ? m->best_count : 0u);
}

int32_t s4p_matcher_set_early_exit(s4p_matcher* m, int32_t enable) {
  if (!m) return S4P_ERR_BAD_ARG;
  m->early_exit = enable != 0;
  return S4P_OK;
}
// (the sharded loop of s4p_shard.cpp brackets its windows with these)
int32_t s4p_matcher_loop_begin(s4p_matcher* m) { if (!m) return S4P_ERR_BAD_ARG; loop_hint(m, true); return S4P_OK; }
int32_t s4p_matcher_loop_end(s4p_matcher* m) { if (!m) return S4P_ERR_BAD_ARG; loop_hint(m, false); return S4P_OK; }

int32_t s4p_matcher_visit_candidates(s4p_matcher* m, int32_t enable) {
  if (!m) return S4P_ERR_BAD_ARG;
  m->visit_candidates = enable != 0;
  return S4P_OK;
}

int32_t s4p_matcher_set_sharding(s4p_matcher* m, int32_t rank, int32_t world, int32_t producer_threads) {
  if (!m || world < 1 || rank < 0 || rank >= world) return S4P_ERR_BAD_ARG;
  producer_stop(m);
  m->prod.rank = rank; m->prod.world = world; m->prod.mode = producer_threads < 0 || producer_threads > 2 ? 2 : producer_threads;
  m->prod.enabled = false; update_helper_threads(m);
  m->prod.consumed = 0; m->prod.next_index = 0;
  m->prod.cap_b = std::max<size_t>(8, 3 * size_t(world));      // the launch thread consumes a window (world trials) at a time
  m->prod.cap_a = std::max<size_t>(24, 3 * size_t(world));
  return S4P_OK;
}

int32_t s4p_matcher_next_base_async(s4p_matcher* m, int32_t run_device, int32_t* found, int32_t* base_ids) {
  if (!m || !found || !base_ids) return S4P_ERR_BAD_ARG;
  if (!m->ready) return m->fail(S4P_ERR_STATE, "matcher not initialised");
  if (int(m->inflight.size()) >= s4p_pipeline_depth(m->ctx)) return m->fail(S4P_ERR_STATE, "all lanes busy: wait_base first");
  s4p_matcher::Prepared pr;
  const int32_t rc = next_base_async(m, run_device != 0, false, pr);
  if (rc != S4P_OK) return rc;
  *found = pr.found ? 1 : 0;
  for (int t = 0; t < 4; ++t) base_ids[t] = pr.ids[t];
  if (pr.device) m->inflight.push_back(std::move(pr));
  return S4P_OK;
}

int32_t s4p_matcher_wait_base(s4p_matcher* m, s4p_base_result* result) {
  if (!m || !result) return S4P_ERR_BAD_ARG;
  if (m->inflight.empty()) return m->fail(S4P_ERR_STATE, "no base in flight");
  s4p_matcher::Prepared pr = std::move(m->inflight.front());
  m->inflight.erase(m->inflight.begin());
  return wait_base(m, pr, *result);
}

int32_t s4p_matcher_commit(s4p_matcher* m, int32_t found, const int32_t* base_ids, const s4p_base_result* result, int32_t* ok) {
  if (!m || !base_ids || !result || !ok) return S4P_ERR_BAD_ARG;
  if (!m->ready) return m->fail(S4P_ERR_STATE, "matcher not initialised");
  const int ids[4] = {base_ids[0], base_ids[1], base_ids[2], base_ids[3]};
  *ok = commit_base(m, found != 0, ids, *result) ? 1 : 0;
  return S4P_OK;
}

// current_trial_ += n (match4pcsBase.hpp:258) for a driver that runs the trial loop itself through the stage-level calls
// (the facade does when a subclass overrides the virtual hooks).
int32_t s4p_matcher_advance_trials(s4p_matcher* m, int32_t n) {
  if (!m || n < 0) return S4P_ERR_BAD_ARG;
  m->current_trial += n;
  return S4P_OK;
}

int32_t s4p_matcher_global_transform(s4p_matcher* m, float* M) {
  if (!m || !M) return S4P_ERR_BAD_ARG;
  global_transform(m, M);
  return S4P_OK;
}

// Undoes everything that ran ahead of the committed trials: the producer threads are stopped (which rewinds to the first
// trial not handed to this thread), the bases handed over but not committed are waited for and dropped, and the RNG
// stream and the octree permutation go back to what they were before the first of them.
static void rewind_speculation(s4p_matcher* m) {
  std::vector<s4p_matcher::Prepared>& fifo = m->inflight;
  if (m->prod.enabled) producer_stop(m);
  if (fifo.empty()) return;
  m->rng = fifo.front().rng_before;
  std::vector<uint32_t> st = fifo.front().pair_state_before;
  for (auto& pr : fifo) {
    s4p_base_result dummy;
    if (pr.device) { (void)s4p_try_base_wait(m->ctx, &dummy); producer_release_slot(m, pr.slot); }
  }
  if (!st.empty()) s4p_pair_state_restore(m->ctx, st.data());
  if (m->prod.enabled) m->prod.consumed -= long(fifo.size());
  fifo.clear();
}

// The reference's per-candidate visitor call v(-1, lcp, T) (match4pcsBase.hpp:458-465) as a candidate sink of the context: the
// verified candidates of a base arrive in reference order, pass by pass for a base that takes several (s4p_capi.h).
namespace {
struct CandidateVisit { s4p_matcher* m; s4p_visitor_fn visitor; void* user; int32_t needs_global; };
void candidate_sink(void* u, const uint32_t* counts, const float* transforms16, int64_t n) {
  const CandidateVisit& cv = *static_cast<const CandidateVisit*>(u);
  const s4p_matcher* m = cv.m;
  for (int64_t k = 0; k < n; ++k) {
    float T[16];
    std::memcpy(T, transforms16 + 16 * size_t(k), sizeof T);
    if (cv.needs_global) {      // getGlobalTransform of :446-456: t_global = t + centroid_P - R * centroid_Q
      for (int a = 0; a < 3; ++a) {
        const float rq = T[4 * a] * m->centroid_q[0] + (T[4 * a + 1] * m->centroid_q[1] + T[4 * a + 2] * m->centroid_q[2]);
        T[4 * a + 3] = (T[4 * a + 3] + m->centroid_p[a]) - rq;
      }
    }
    cv.visitor(cv.user, -1.f, float(counts[size_t(k)]) / float(m->Qs.size()), T);
  }
}
}  // namespace

int32_t s4p_matcher_perform_n_steps(s4p_matcher* m, int32_t n, s4p_visitor_fn visitor, void* user, int32_t needs_global,
                                    float* transformation, int32_t* improved, int32_t* done) {
  if (!m || !transformation || !improved || !done) return S4P_ERR_BAD_ARG;
  if (!m->ready) return m->fail(S4P_ERR_STATE, "matcher not initialised");
  using sclock = std::chrono::system_clock;
  const float last_best = m->best_lcp;
  if (visitor) visitor(user, 0.f, m->best_lcp, transformation);          // match4pcsBase.hpp:232
  bool ok = false;
  const auto t0 = sclock::now();
  // The reference loop "for i: TryOneBase; ...; if (stop) break" with the device pass of trial i+1 enqueued
  // before the result of trial i is waited for.  Base selection never reads results, so the speculation only
  // has to be rolled back (RNG + pair-octree permutation) when the loop stops.
  const int end = m->current_trial + n;
  int next_prep = m->current_trial;
  std::vector<s4p_matcher::Prepared>& fifo = m->inflight;
  drain_inflight(m);                       // leftovers of next_base_async calls the caller never waited for
  loop_hint(m, true);
  int32_t rc = S4P_OK;
  static const bool trace_call = std::getenv("S4P_TRACE_CALL") != nullptr;      // lab aid: where a short call spends its fixed cost
  using hclock = std::chrono::steady_clock;
  const auto tc0 = hclock::now();
  hclock::time_point tc_first_enq = tc0, tc_first_res = tc0, tc_loop_end = tc0;
  bool seen_enq = false, seen_res = false;
  for (int i = m->current_trial; i < end && rc == S4P_OK; ++i) {
    while (int(fifo.size()) < s4p_pipeline_depth(m->ctx) && next_prep < end) {
      fifo.emplace_back();
      if ((rc = next_base_async(m, true, true, fifo.back())) != S4P_OK) break;
      if (trace_call && !seen_enq) { tc_first_enq = hclock::now(); seen_enq = true; }
      ++next_prep;
    }
    if (rc != S4P_OK) break;
    s4p_matcher::Prepared pr = std::move(fifo.front());
    fifo.erase(fifo.begin());
    s4p_base_result r;
    // match4pcsBase.hpp:458-465: the per-candidate calls of this base are made from inside the wait (the sink is set for this
    // wait only: bases that are drained when the loop stops must not produce calls the sequential loop never made)
    CandidateVisit cv{m, visitor, user, needs_global};
    const bool listen = visitor && m->visit_candidates && pr.device;
    if (listen) (void)s4p_set_candidate_sink(m->ctx, candidate_sink, &cv);
    rc = wait_base(m, pr, r);
    if (listen) (void)s4p_set_candidate_sink(m->ctx, nullptr, nullptr);
    if (rc != S4P_OK) break;
    if (trace_call && !seen_res) { tc_first_res = hclock::now(); seen_res = true; }
    ok = commit_base(m, pr.found, pr.ids, r);
    const float fraction_try = float(i) / float(m->number_of_trials);
    // integer seconds / integer max_time_seconds: reference quirk, :240-243
    const float fraction_time = float(std::chrono::duration_cast<std::chrono::seconds>(sclock::now() - t0).count() /
                                      (long)m->opt.max_time_seconds);
    const float fraction = std::max(fraction_time, fraction_try);
    if (needs_global) global_transform(m, transformation);
    else std::memcpy(transformation, m->transform, sizeof(float) * 16);
    if (visitor) visitor(user, fraction, m->best_lcp, transformation);
    if (ok || i > m->number_of_trials || fraction >= 0.99 || m->best_lcp == 1.0) break;
  }
  // drain speculative work and put the host state back where the sequential loop stopped
  if (trace_call) tc_loop_end = hclock::now();
  rewind_speculation(m);
  loop_hint(m, false);
  if (trace_call) {
    auto us = [](hclock::time_point a, hclock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "[s4p] perform_n_steps(%d): first base enqueued after %.0f us, first result after %.0f us, loop %.0f us, rewind %.0f us\n",
                 n, us(tc0, tc_first_enq), us(tc0, tc_first_res), us(tc0, tc_loop_end), us(tc_loop_end, hclock::now()));
  }
  if (rc != S4P_OK) return rc;
  m->current_trial += n;
  *improved = m->best_lcp > last_best ? 1 : 0;
  if (*improved) global_transform(m, transformation);
  *done = (ok || m->current_trial >= m->number_of_trials) ? 1 : 0;
  return S4P_OK;
}

int32_t s4p_matcher_compute_transformation(s4p_matcher* m, const s4p_cloud_view* P, const s4p_cloud_view* Q,
                                           float* qx, float* qy, float* qz, float* M, float* lcp) {
  if (!m || !M || !lcp) return S4P_ERR_BAD_ARG;
  *lcp = 1e9f;                                                     // kLargeNumber, match4pcsBase.hpp:69-70
  if (!Q || !P) return S4P_OK;
  if (!view_ok(P) || !view_ok(Q)) return m->fail(S4P_ERR_BAD_ARG, "null coordinate arrays");
  if (P->n == 0 || Q->n == 0) return S4P_OK;
  if (int32_t rc = s4p_matcher_init_full(m, P, Q)) return rc;
  int32_t improved = 0, done = 0;
  if (m->best_lcp != 1.f)
    if (int32_t rc = s4p_matcher_perform_n_steps(m, m->number_of_trials, nullptr, nullptr, 0, M, &improved, &done)) return rc;
  *lcp = m->best_lcp;
  if (improved && qx && qy && qz) {                                // match4pcsBase.hpp:259-268
    if (qx != Q->x) std::memcpy(qx, Q->x, size_t(Q->n) * 4);
    if (qy != Q->y) std::memcpy(qy, Q->y, size_t(Q->n) * 4);
    if (qz != Q->z) std::memcpy(qz, Q->z, size_t(Q->n) * 4);
    if (int32_t rc = s4p_transform_points(m->ctx, M, qx, qy, qz, Q->n)) return m->ctx_fail(rc);
  }
  return S4P_OK;
}

}  // extern "C"
