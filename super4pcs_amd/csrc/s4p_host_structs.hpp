// s4p_host_structs.hpp -- host-side (C++) structures that feed the gfx950 kernels.
//
//  * PairOctree : loop 1 of IntersectionFunctor::process
//      (src/super4pcs/accelerators/pairExtraction/intersectionFunctor.h:139-191,
//       intersectionNode.h:156-245).  It owns the persistent `ids` permutation that the
//      reference keeps in PairCreationFunctor (pairCreationFunctor.h:36,120) and that is
//      partitioned in place by every split, never reset between ExtractPairs calls --
//      so the emission order of call k depends on calls 1..k-1.
//      Output: the flat "sequence" (leaf-major, position-minor) the k_pairs kernel walks.
//  * UnitFrame  : PairCreationFunctor::synch3DContent (pairCreationFunctor.h:90-122).
//  * LcpGridHost: uniform grid over the sampled P cloud replacing the kd-tree of
//      Match4PCSBase::initKdTree (match4pcsBase.cc:353-363); same inlier predicate.
//
// All float arithmetic here must stay un-fused (-ffp-contract=off) because the unit
// coordinates and cell indices are inputs of bit-exact device predicates.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace s4p {

struct Leaf { float cx, cy, cz, h; };   // node centre + half-edge argument handed to the box test

// ---------------------------------------------------------------------------
struct UnitFrame {
  float gcenter[3] = {0, 0, 0};
  float ratio = 1.f;
  // qx,qy,qz: sampled, centred Q.  Writes unit-cube coordinates.
  void build(const std::vector<float>& qx, const std::vector<float>& qy, const std::vector<float>& qz,
             std::vector<float>& ux, std::vector<float>& uy, std::vector<float>& uz) {
    const size_t n = qx.size();
    const float big = std::numeric_limits<float>::max() / 2;   // accelerators/bbox.h:71-73
    float lo[3] = {big, big, big}, hi[3] = {-big, -big, -big};
    for (size_t i = 0; i < n; ++i) {
      const float v[3] = {qx[i], qy[i], qz[i]};
      for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); }
    }
    float ext = -std::numeric_limits<float>::infinity();
    for (int k = 0; k < 3; ++k) {
      gcenter[k] = (lo[k] + hi[k]) / 2.f;                       // AlignedBox::center()
      const float d = hi[k] - lo[k];                             // AlignedBox::diagonal()
      if (k == 0 || d > ext) ext = d;
    }
    ratio = float(double(ext) + 0.001);                          // pairCreationFunctor.h:111
    ux.resize(n); uy.resize(n); uz.resize(n);
    for (size_t i = 0; i < n; ++i) {                             // worldToUnit, :65-69
      ux[i] = (qx[i] - gcenter[0]) / ratio + 0.5f;
      uy[i] = (qy[i] - gcenter[1]) / ratio + 0.5f;
      uz[i] = (qz[i] - gcenter[2]) / ratio + 0.5f;
    }
  }
};

// ---------------------------------------------------------------------------
class PairOctree {
 public:
  std::vector<uint32_t> ids;                 // persistent permutation
  // outputs of build()
  std::vector<uint32_t> seq_id, seq_leaf;
  std::vector<Leaf> leaves;
  float eps_unit = 0.f;                      // power-of-two rounded epsilon
  float n_radius = 0.f;

  void reset(uint32_t n) { ids.resize(n); for (uint32_t i = 0; i < n; ++i) ids[i] = i; }

  // GetRoundedEpsilonValue, intersectionFunctor.h:59-67
  static float rounded_epsilon(float eps, int* lvl) {
    const int lvlMax = int(-std::log2(eps));
    *lvl = lvlMax;
    return float(1.f / std::pow(2, lvlMax));
  }

  void build(const float* ux, const float* uy, const float* uz, uint32_t n, float radius_unit, float eps_in,
             uint32_t min_node_size = 50) {
    u_[0] = ux; u_[1] = uy; u_[2] = uz;
    n_radius = radius_unit;
    int lvlMax = 0;
    eps_unit = rounded_epsilon(eps_in, &lvlMax);
    if (ids.size() != n) reset(n);                                // intersectionFunctor.h:139-144
    cur_.clear(); nxt_.clear(); early_.clear();
    nxt_.push_back(Node{{0.5f, 0.5f, 0.5f}, 0u, n});              // buildUnitRootNode
    int lvl = 0;
    while (lvl != lvlMax - 1) {                                   // first loop, :154-191
      if (nxt_.empty()) break;
      const float edge = float(1.f / std::pow(2, lvl));
      const float half = edge / 2.f;
      cur_.swap(nxt_);
      nxt_.clear();
      const float reach = half + eps_unit;
      for (const Node& nd : cur_) {
        bool hit = false;
        for (uint32_t p = 0; p < n && !hit; ++p) hit = sphere_touches_box(ux[p], uy[p], uz[p], radius_unit, nd.c, reach);
        if (!hit) continue;
        if (int(nd.end) - int(nd.begin) > int(min_node_size)) split8(nd, half);
        else early_.push_back(EarlyNode{nd, reach});
      }
      ++lvl;
    }
    // flatten: final-level children first, then the parked early leaves (second loop order, :201-232)
    leaves.clear(); seq_id.clear(); seq_leaf.clear();
    auto emit = [&](const Node& nd, float h) {
      const uint32_t li = uint32_t(leaves.size());
      leaves.push_back(Leaf{nd.c[0], nd.c[1], nd.c[2], h});
      for (uint32_t k = nd.begin; k < nd.end; ++k) { seq_id.push_back(ids[k]); seq_leaf.push_back(li); }
    };
    for (const Node& nd : nxt_) emit(nd, eps_unit * 2.f);
    for (const EarlyNode& en : early_) emit(en.node, en.reach);
  }

 private:
  struct Node { float c[3]; uint32_t begin, end; };
  struct EarlyNode { Node node; float reach; };
  const float* u_[3] = {nullptr, nullptr, nullptr};
  std::vector<Node> cur_, nxt_;
  std::vector<EarlyNode> early_;

  // HyperSphere::intersect, intersectionPrimitive.h:117-142 (Arvo box/sphere-surface test)
  static bool sphere_touches_box(float cx, float cy, float cz, float r, const float* nc, float h) {
    const float c[3] = {cx, cy, cz};
    float lo_t[3], hi_t[3];
    for (int k = 0; k < 3; ++k) {
      const float mn = nc[k] - h, mx = nc[k] + h;
      const float a = (c[k] - mn) * (c[k] - mn);
      const float b = (c[k] - mx) * (c[k] - mx);
      lo_t[k] = (c[k] < mn) ? a : ((c[k] > mx) ? b : 0.f);
      hi_t[k] = (a < b) ? b : a;
    }
    const float r2 = r * r;
    return (lo_t[0] + (lo_t[1] + lo_t[2])) < r2 && r2 < (hi_t[0] + (hi_t[1] + hi_t[2]));
  }

  // NdNode::_split, intersectionNode.h:156-176: in-place partition of ids[start,end) around v on axis d
  uint32_t partition(int start, int end, unsigned d, float v) {
    const float* a = u_[d];
    int l = start, r = end - 1;
    for (; l < r; ++l, --r) {
      while (l < end && a[ids[l]] < v) l++;
      while (r >= start && a[ids[r]] >= v) r--;
      if (l > r) break;
      std::swap(ids[l], ids[r]);
    }
    if (l >= end) return uint32_t(end);
    return a[ids[l]] < v ? uint32_t(l + 1) : uint32_t(l);
  }

  // NdNode::split, intersectionNode.h:187-245: 8 children in x-fastest... order fixed by the
  // successive per-dimension splits; empty children are dropped, order otherwise preserved.
  void split8(const Node& parent, float parent_half) {
    Node ch[8];
    for (auto& c : ch) c = parent;
    const float q = parent_half / 2.f;
    for (unsigned d = 0; d < 3; ++d) {
      const unsigned n_split = 1u << d;         // nbInterval/2
      const unsigned span = 8u / n_split;       // intervalNode
      const unsigned mid = span / 2u;           // midNode
      for (unsigned s = 0; s < n_split; ++s) {
        const unsigned b = s * span, e = (s + 1) * span;
        const float centre = ch[b].c[d];
        const uint32_t cut = partition(int(ch[b].begin), int(ch[e - 1].end), d, centre);
        const float lo = centre - q, hi = centre + q;
        for (unsigned i = b; i < b + mid; ++i) { ch[i].c[d] = lo; ch[i].end = cut; }
        for (unsigned i = b + mid; i < e; ++i) { ch[i].c[d] = hi; ch[i].begin = cut; }
      }
    }
    for (const Node& c : ch) if (c.end != c.begin) nxt_.push_back(c);
  }
};

// ---------------------------------------------------------------------------
// Dimensioning of the LCP grid (the structure itself is built on the device, s4p_kernels.hip.hpp k_grid_*).
struct LcpGridHost {
  float ox = 0, oy = 0, oz = 0, h = 1, inv_h = 1;
  int nx = 1, ny = 1, nz = 1;
  int cshift = 0, cnx = 1, cny = 1, cnz = 1;
  uint32_t coarse_words = 0;
  double reach = 0;                         // 1.01 * delta
  uint64_t ncell() const { return uint64_t(nx) * uint64_t(ny) * uint64_t(nz); }

  // Cell edge h >= 1.002*delta.  A query q is mapped to cell floor((q-o)*inv_h) in float; a P point p with
  // |q-p| <= delta then satisfies dist(p, box(cell)) <= delta + rounding slack, so it is listed for that cell by the
  // 1.01*delta reach test (slack 1e-2*delta >> float rounding of the cell map).  One cell of padding on every side.
  bool plan(const std::vector<float>& px, const std::vector<float>& py, const std::vector<float>& pz,
            float delta, uint64_t max_cells, uint32_t max_coarse_words) {
    const size_t n = px.size();
    if (n == 0) return false;
    float lo[3] = {px[0], py[0], pz[0]}, hi[3] = {px[0], py[0], pz[0]};
    for (size_t i = 1; i < n; ++i) {
      lo[0] = std::min(lo[0], px[i]); hi[0] = std::max(hi[0], px[i]);
      lo[1] = std::min(lo[1], py[i]); hi[1] = std::max(hi[1], py[i]);
      lo[2] = std::min(lo[2], pz[i]); hi[2] = std::max(hi[2], pz[i]);
    }
    h = delta * 1.002f;
    if (!(h > 0.f)) return false;
    while (true) {
      const double ex = (double(hi[0]) - lo[0]) / h, ey = (double(hi[1]) - lo[1]) / h, ez = (double(hi[2]) - lo[2]) / h;
      if (ex < 2.0e9 && ey < 2.0e9 && ez < 2.0e9) {
        nx = int(ex) + 4; ny = int(ey) + 4; nz = int(ez) + 4;
        if (ncell() <= max_cells && ncell() < 0xFFFFFFF0ull) break;
      }
      h *= 1.25f;
    }
    inv_h = 1.0f / h;
    ox = lo[0] - 1.5f * h; oy = lo[1] - 1.5f * h; oz = lo[2] - 1.5f * h;
    reach = double(delta) * 1.01;
    cshift = 0;      // coarse level: smallest shift whose bitmap fits the LDS budget (padded to 16 B for the staging)
    while (true) {
      cnx = ((nx - 1) >> cshift) + 1; cny = ((ny - 1) >> cshift) + 1; cnz = ((nz - 1) >> cshift) + 1;
      const uint64_t cw = (uint64_t(cnx) * cny * cnz + 31) / 32;
      if (cw <= max_coarse_words) { coarse_words = uint32_t((cw + 3) & ~uint64_t(3)); break; }
      ++cshift;
    }
    return true;
  }
};

}  // namespace s4p
