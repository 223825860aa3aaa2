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
//  * FourthPointIndex: block-pruned form of the 4th-point scan of SelectQuadrilateral (match4pcsBase.cc:324-338).
//  * LcpGridHost: uniform grid over the sampled P cloud replacing the kd-tree of
//      Match4PCSBase::initKdTree (match4pcsBase.cc:353-363); same inlier predicate.
//
// All float arithmetic here must stay un-fused (-ffp-contract=off) because the unit
// coordinates and cell indices are inputs of bit-exact device predicates.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <limits>
#include <utility>
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
  float eps_unit = 0.f;                      // power-of-two rounded epsilon
  float n_radius = 0.f;

  void reset(uint32_t n) { ids.resize(n); for (uint32_t i = 0; i < n; ++i) ids[i] = i; forget_splits(); }

  // Must be called whenever `ids` is overwritten from outside (state restore): the remembered splits describe
  // the permutation they were made on.
  void forget_splits() { memo_.clear(); shells_.clear(); shell_words_ = 0; }

  // GetRoundedEpsilonValue, intersectionFunctor.h:59-67
  static float rounded_epsilon(float eps, int* lvl) {
    const int lvlMax = int(-std::log2(eps));
    *lvl = lvlMax;
    return float(1.f / std::pow(2, lvlMax));
  }

  // Loop 1: refines the tree and permutes `ids`.  The leaves stay in nxt_/early_ until flatten() is asked for them
  // (a rank that only advances the permutation for a base it does not own never flattens).
  void build(const float* ux, const float* uy, const float* uz, uint32_t n, float radius_unit, float eps_in,
             uint32_t min_node_size = 50) {
    u_[0] = ux; u_[1] = uy; u_[2] = uz;
    n_radius = radius_unit;
    int lvlMax = 0;
    eps_unit = rounded_epsilon(eps_in, &lvlMax);
    if (ids.size() != n) reset(n);                                // intersectionFunctor.h:139-144
    cur_.clear(); nxt_.clear(); early_.clear();
    if (memo_.empty()) memo_.push_back(Memo{});
    nxt_.push_back(Node{{0.5f, 0.5f, 0.5f}, 0u, n, 0});           // buildUnitRootNode
    int lvl = 0;
    while (lvl != lvlMax - 1) {                                   // first loop, :154-191
      if (nxt_.empty()) break;
      const float edge = float(1.f / std::pow(2, lvl));
      const float half = edge / 2.f;
      cur_.swap(nxt_);
      nxt_.clear();
      const float reach = half + eps_unit;
      for (const Node& nd : cur_) {
        if (!any_sphere_touches(nd, n, radius_unit, reach)) continue;
        if (int(nd.end) - int(nd.begin) > int(min_node_size)) split8(nd, half);
        else early_.push_back(EarlyNode{nd, reach});
      }
      ++lvl;
    }
    n_seq_ = 0;
    for (const Node& nd : nxt_) n_seq_ += nd.end - nd.begin;
    for (const EarlyNode& en : early_) n_seq_ += en.node.end - en.node.begin;
  }

  uint32_t n_seq() const { return n_seq_; }
  uint32_t n_leaf() const { return uint32_t(nxt_.size() + early_.size()); }

  // The flat "sequence" k_pairs walks: final-level children first, then the parked early leaves
  // (second loop order, intersectionFunctor.h:201-232).  seq_id holds n_seq() point ids; leaf l owns the slots
  // [leaf_off[l], leaf_off[l+1]) of it (n_leaf() + 1 offsets; leaves are never empty), leaves[l] is its box.
  void flatten(uint32_t* seq_id, uint32_t* leaf_off, Leaf* leaves) const {
    uint32_t li = 0, at = 0;
    auto emit = [&](const Node& nd, float h) {
      leaves[li] = Leaf{nd.c[0], nd.c[1], nd.c[2], h};
      leaf_off[li] = at;
      for (uint32_t k = nd.begin; k < nd.end; ++k, ++at) seq_id[at] = ids[k] | (li << 16);      // id | leaf << 16 (both < 2^16: n_Q <= 46 340)
      ++li;
    };
    for (const Node& nd : nxt_) emit(nd, eps_unit * 2.f);
    for (const EarlyNode& en : early_) emit(en.node, en.reach);
    leaf_off[li] = at;
  }

 private:
  struct Node { float c[3]; uint32_t begin, end; int32_t memo; };
  // Cell centres never change and `ids` persists, so the FIRST split of a cell fixes its eight child ranges for
  // good: children only permute inside their own ranges, and the reference's two-pointer partition of an already
  // partitioned range swaps nothing and returns the same cut (l stops at the cut, r just below it, `l > r` breaks,
  // intersectionNode.h:156-176).  Later splits of that cell therefore replay the remembered boundaries instead of
  // re-scanning the points: same `ids`, same children, a build costs the sphere/box tests only.
  struct Memo { bool split = false; int32_t shell = -1; uint32_t bound[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; int32_t child[8] = {-1, -1, -1, -1, -1, -1, -1, -1}; };
  std::vector<Memo> memo_;

  // "Does ANY primitive sphere touch this cell?" (the p-loop of intersectionFunctor.h:163-188 only needs existence).
  // The sphere centres are the cloud's own points and never move, and a cell's centre never moves either, so each
  // cell keeps the points binned by their distance to its centre (built the first time the cell is tested).  A sphere
  // of radius r can only touch a box of inflated half-edge h if its centre lies within sqrt(3) h of distance r from
  // the box centre, so only the bins of that shell are tested -- nearest-to-r first -- with the exact predicate;
  // everything outside the shell fails it by construction (the margin below is ~1000x the float rounding of either
  // side).  On the benchmark cloud this turns ~7200 sphere/box tests per build into a few hundred.
  static constexpr int kShellBins = 128;
  static constexpr float kShellWidth = 1.75f / kShellBins;      // unit cube: distances < sqrt(3)
  struct Shells { std::vector<uint32_t> order; uint32_t start[kShellBins + 1]; };
  std::deque<Shells> shells_;
  size_t shell_words_ = 0;
  static constexpr size_t kShellBudgetWords = size_t(16) << 20;  // 64 MB of bins at most; beyond that: plain loop

  static int shell_bin(float d) { const int b = int(d * (1.0f / kShellWidth)); return b < 0 ? 0 : (b >= kShellBins ? kShellBins - 1 : b); }

  void build_shells(Memo& mm, const float* c, uint32_t n) {
    if (shell_words_ + n > kShellBudgetWords) return;
    shells_.emplace_back();
    Shells& sh = shells_.back();
    std::vector<uint8_t> bin(n);
    for (int b = 0; b <= kShellBins; ++b) sh.start[b] = 0;
    for (uint32_t p = 0; p < n; ++p) {
      const float dx = u_[0][p] - c[0], dy = u_[1][p] - c[1], dz = u_[2][p] - c[2];
      const float d = std::sqrt(dx * dx + dy * dy + dz * dz);
      bin[p] = uint8_t(d == d ? shell_bin(d) : kShellBins - 1);
      ++sh.start[bin[p] + 1];
    }
    for (int b = 1; b <= kShellBins; ++b) sh.start[b] += sh.start[b - 1];
    sh.order.resize(n);
    uint32_t at[kShellBins];
    for (int b = 0; b < kShellBins; ++b) at[b] = sh.start[b];
    for (uint32_t p = 0; p < n; ++p) sh.order[at[bin[p]]++] = p;
    shell_words_ += n;
    mm.shell = int32_t(shells_.size() - 1);
  }

  bool any_sphere_touches(const Node& nd, uint32_t n, float r, float reach) {
    const float* ux = u_[0]; const float* uy = u_[1]; const float* uz = u_[2];
    Memo& mm = memo_[size_t(nd.memo)];
    if (mm.shell < 0) build_shells(mm, nd.c, n);
    const float D = 1.7320508f * reach + 1e-3f;
    if (mm.shell < 0 || !(r == r) || !(D == D) || shells_[size_t(mm.shell)].order.size() != n) {
      for (uint32_t p = 0; p < n; ++p) if (sphere_touches_box(ux[p], uy[p], uz[p], r, nd.c, reach)) return true;
      return false;
    }
    const Shells& sh = shells_[size_t(mm.shell)];
    // bins are taken one wider than the shell on either side; the last bin also holds everything beyond 1.75
    const int b_lo = r - D <= 0.f ? 0 : std::max(0, shell_bin(r - D) - 1);
    const int b_hi = std::min(kShellBins - 1, shell_bin(r + D) + 1);
    const int bc = std::min(b_hi, std::max(b_lo, shell_bin(r)));
    auto test_bin = [&](int b) {
      for (uint32_t k = sh.start[b]; k < sh.start[b + 1]; ++k) {
        const uint32_t p = sh.order[k];
        if (sphere_touches_box(ux[p], uy[p], uz[p], r, nd.c, reach)) return true;
      }
      return false;
    };
    const int span = std::max(bc - b_lo, b_hi - bc);
    for (int k = 0; k <= span; ++k) {
      if (bc + k <= b_hi && test_bin(bc + k)) return true;
      if (k > 0 && bc - k >= b_lo && test_bin(bc - k)) return true;
    }
    return false;
  }
  struct EarlyNode { Node node; float reach; };
  const float* u_[3] = {nullptr, nullptr, nullptr};
  std::vector<Node> cur_, nxt_;
  std::vector<EarlyNode> early_;
  uint32_t n_seq_ = 0;

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

  // NdNode::_split, intersectionNode.h:156-176: in-place partition of ids[start,end) around v on axis d.
  // (`ids` persists across builds and the cell centres never change, so after the first few bases every range
  // arrives almost partitioned and this two-pointer scan is branch-predictable; a branch-free count/collect/swap
  // formulation was measured 2x slower on such input.)
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
    const bool known = memo_[size_t(parent.memo)].split;
    for (unsigned d = 0; d < 3; ++d) {
      const unsigned n_split = 1u << d;         // nbInterval/2
      const unsigned span = 8u / n_split;       // intervalNode
      const unsigned mid = span / 2u;           // midNode
      for (unsigned s = 0; s < n_split; ++s) {
        const unsigned b = s * span, e = (s + 1) * span;
        const float centre = ch[b].c[d];
        const uint32_t cut = known ? memo_[size_t(parent.memo)].bound[b + mid]
                                   : partition(int(ch[b].begin), int(ch[e - 1].end), d, centre);
        const float lo = centre - q, hi = centre + q;
        for (unsigned i = b; i < b + mid; ++i) { ch[i].c[d] = lo; ch[i].end = cut; }
        for (unsigned i = b + mid; i < e; ++i) { ch[i].c[d] = hi; ch[i].begin = cut; }
      }
    }
    if (!known) {
      Memo& mm = memo_[size_t(parent.memo)];
      for (unsigned i = 0; i < 8; ++i) mm.bound[i] = ch[i].begin;
      mm.bound[8] = ch[7].end;
      mm.split = true;
    }
    for (unsigned i = 0; i < 8; ++i) {
      if (ch[i].end == ch[i].begin) continue;
      int32_t cm = memo_[size_t(parent.memo)].child[i];
      if (cm < 0) { cm = int32_t(memo_.size()); memo_.push_back(Memo{}); memo_[size_t(parent.memo)].child[i] = cm; }
      ch[i].memo = cm;
      nxt_.push_back(ch[i]);
    }
  }
};

// ---------------------------------------------------------------------------
// FourthPointIndex: the 4th-point search of SelectQuadrilateral (match4pcsBase.cc:324-338) without touching every
// sampled P point.  The reference walks all of P for the point with the smallest |a x + b y + c z - 1| that is not
// within `too_small` of the three base points; the first index wins ties.  Here P is kept a second time in Morton
// order, in blocks of 32 points with their bounding boxes: a block whose box cannot come closer to the plane than the
// best distance found so far is skipped after one interval evaluation, so a query reads the ~5 % of P that lies near
// the plane.  The per-point distance is the same float expression as the linear scan, and the winner is the
// lexicographic minimum of (distance, original index) -- exactly the first strictly smaller distance in index order.
// The box bound is relaxed by a margin far above the float rounding of either evaluation, so no block that could
// hold the winner (or a tie) is ever skipped.
class FourthPointIndex {
 public:
  static constexpr uint32_t kBlock = 32;
  bool empty() const { return n_ == 0; }

  void build(const float* x, const float* y, const float* z, size_t n) {
    n_ = n;
    nb_ = (n + kBlock - 1) / kBlock;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
      const float v[3] = {x[i], y[i], z[i]};
      for (int k = 0; k < 3; ++k) { if (i == 0 || v[k] < lo[k]) lo[k] = v[k]; if (i == 0 || v[k] > hi[k]) hi[k] = v[k]; }
    }
    for (int k = 0; k < 3; ++k) absmax_[k] = std::max(std::fabs(lo[k]), std::fabs(hi[k]));
    // spatial order: counting sort by the Morton code of a coarse grid cell (about four points per cell), input
    // order kept inside a cell -- O(n), a fraction of a millisecond where a comparison sort took several
    int bits = 3;
    while (bits < 7 && (size_t(1) << (3 * (bits + 1))) <= n / 4 + 1) ++bits;
    const float cells = float(1u << bits);
    auto spread = [](uint32_t v) { v &= 0x3FFu; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu;
                                   v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u; return v; };
    std::vector<uint32_t> code(n), start((size_t(1) << (3 * bits)) + 1, 0u), order(n);
    float scale[3];
    for (int k = 0; k < 3; ++k) scale[k] = hi[k] > lo[k] ? cells / (hi[k] - lo[k]) : 0.f;
    for (size_t i = 0; i < n; ++i) {
      uint32_t q[3];
      const float v[3] = {x[i], y[i], z[i]};
      for (int k = 0; k < 3; ++k) {
        const float t = (v[k] - lo[k]) * scale[k];
        q[k] = uint32_t(std::min(cells - 1.f, std::max(0.f, t)));       // NaN -> 0
      }
      code[i] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
      ++start[code[i] + 1];
    }
    for (size_t c = 1; c < start.size(); ++c) start[c] += start[c - 1];
    for (size_t i = 0; i < n; ++i) order[start[code[i]]++] = uint32_t(i);
    const size_t padded = nb_ * kBlock;
    bx_.assign(padded, 0.f); by_.assign(padded, 0.f); bz_.assign(padded, 0.f); bi_.assign(padded, 0xFFFFFFFFu);
    for (int k = 0; k < 3; ++k) { blo_[k].assign(nb_, 0.f); bhi_[k].assign(nb_, 0.f); }
    for (size_t b = 0; b < nb_; ++b) {
      const size_t s = b * kBlock, e = std::min(n, s + kBlock);
      for (size_t k = s; k < padded && k < s + kBlock; ++k) {
        const uint32_t src = order[std::min(k, e - 1)];      // pad the last block with copies of a real point
        bx_[k] = x[src]; by_[k] = y[src]; bz_[k] = z[src];
        bi_[k] = k < e ? src : 0xFFFFFFFFu;                          // ...that can never win (index = +inf)
      }
      for (size_t k = s; k < e; ++k) {
        const float v[3] = {bx_[k], by_[k], bz_[k]};
        for (int d = 0; d < 3; ++d) {
          if (k == s || v[d] < blo_[d][b]) blo_[d][b] = v[d];
          if (k == s || v[d] > bhi_[d][b]) bhi_[d][b] = v[d];
        }
      }
    }
  }

  // Returns the index (into the original arrays) or -1.  A, B, C: the base triangle; pa, pb, pc: the plane.
  // lb: the caller's scratch (one float per block; resized here) -- concurrent queries from several threads each bring their own
  int query(float pa, float pb, float pc, const float* A, const float* B, const float* C, float too_small, std::vector<float>& lb) const {
    if (lb.size() < nb_) lb.resize(nb_);
    const float margin = 4e-6f * (std::fabs(pa) * absmax_[0] + std::fabs(pb) * absmax_[1] + std::fabs(pc) * absmax_[2] + 1.0f);
    // interval of (a x + b y + c z - 1) over each block's box: the low end takes, per axis, the box face on the side
    // the coefficient's sign points away from
    const float* __restrict x0 = (pa >= 0.f ? blo_[0] : bhi_[0]).data(); const float* __restrict x1 = (pa >= 0.f ? bhi_[0] : blo_[0]).data();
    const float* __restrict y0 = (pb >= 0.f ? blo_[1] : bhi_[1]).data(); const float* __restrict y1 = (pb >= 0.f ? bhi_[1] : blo_[1]).data();
    const float* __restrict z0 = (pc >= 0.f ? blo_[2] : bhi_[2]).data(); const float* __restrict z1 = (pc >= 0.f ? bhi_[2] : blo_[2]).data();
    float* __restrict lbp = lb.data();
    for (size_t b = 0; b < nb_; ++b) {
      const float vmin = ((pa * x0[b] + pb * y0[b]) + pc * z0[b]) - 1.0f;
      const float vmax = ((pa * x1[b] + pb * y1[b]) + pc * z1[b]) - 1.0f;
      const float away = vmin > -vmax ? vmin : -vmax;
      lbp[b] = (away > 0.0f ? away : 0.0f) - margin;
    }
    float best = std::numeric_limits<float>::max();
    uint32_t best_i = 0xFFFFFFFFu;
    auto visit = [&](size_t b) {
      const size_t s = b * kBlock;
      float dist[kBlock];
      float blockmin = std::numeric_limits<float>::max();
      for (uint32_t k = 0; k < kBlock; ++k) {
        const float v = (pa * bx_[s + k] + pb * by_[s + k]) + pc * bz_[s + k];
        const float d = std::fabs(v - 1.0f);
        dist[k] = d;
        blockmin = d < blockmin ? d : blockmin;
      }
      if (!(blockmin <= best)) return;
      for (uint32_t k = 0; k < kBlock; ++k) {
        const float d = dist[k];
        const uint32_t i = bi_[s + k];
        if (!(d < best || (d == best && i < best_i && best_i != 0xFFFFFFFFu))) continue;
        const float p[3] = {bx_[s + k], by_[s + k], bz_[s + k]};
        if (i != 0xFFFFFFFFu && far_enough(p, A, too_small) && far_enough(p, B, too_small) && far_enough(p, C, too_small)) { best = d; best_i = i; }
      }
    };
    for (size_t b = 0; b < nb_; ++b) if (lbp[b] <= 0.0f) visit(b);          // boxes the plane passes through
    for (size_t b = 0; b < nb_; ++b) if (lbp[b] > 0.0f && lbp[b] <= best) visit(b);
    return best_i == 0xFFFFFFFFu ? -1 : int(best_i);
  }

 private:
  static bool far_enough(const float* p, const float* q, float too_small) {
    const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    return dx * dx + (dy * dy + dz * dz) >= too_small;                  // (p - q).squaredNorm(), Eigen order
  }
  size_t n_ = 0, nb_ = 0;
  float absmax_[3] = {0, 0, 0};
  std::vector<float> bx_, by_, bz_, blo_[3], bhi_[3];
  std::vector<uint32_t> bi_;
};

// ---------------------------------------------------------------------------
// Dimensioning of the LCP grid (the structure itself is built on the device, s4p_kernels.hip.hpp k_grid_*).
struct LcpGridHost {
  static constexpr float kMinCellFactor = 1.02f;
  float ox = 0, oy = 0, oz = 0, h = 1, inv_h = 1;
  int nx = 1, ny = 1, nz = 1;
  int cshift = 0, cnx = 1, cny = 1, cnz = 1;
  uint32_t coarse_words = 0;
  double reach = 0;                         // delta + 0.01 * h
  uint64_t ncell() const { return uint64_t(nx) * uint64_t(ny) * uint64_t(nz); }

  // Cell edge h >= 1.02 * delta.  The sweep LOCATES a query approximately: cell = floor of a fused-multiply-add transform
  // of the query quantised to 16 bit (at most 0.004 cell per axis, s4p_set_clouds), i.e. up to e = 0.007 cell away from
  // the cell of the exactly transformed query.  A P point p with |q - p| <= delta must still be in the list of the LOCATED
  // cell c.  (i) Lists (and sub-cell masks) hold the points within reach = delta + 0.01 h of the cell's box, and
  // dist(p, box(c)) <= delta + e h: the slack is a fraction of the CELL, like the locate error, so it also covers the
  // cells enlarged by the loop below (x 1.25 per step when the grid would exceed max_cells or the 2^24 index limits) or by
  // S4P_CELL_FACTOR -- with the former fixed 1.01 delta the condition e h <= 0.01 delta failed from h ~ 1.4 delta on.
  // (ii) The lists are built from each point's 27-cell neighbourhood only, so p's own cell must be a neighbour of c:
  // p is at most delta + e h from box(c), which is less than one cell iff h (1 - e) >= delta, i.e. h >= 1.007 delta.
  // With the former 1.002 an inlier at distance ~delta straight across a cell face could land two cells from the located
  // cell and be missed (8 of 127 902 in tools/probe_locate_slack.py; tests/test_gpu_kernels.py holds that case now).
  bool plan(const std::vector<float>& px, const std::vector<float>& py, const std::vector<float>& pz,
            float delta, uint64_t max_cells, uint32_t max_coarse_words, float cell_factor = kMinCellFactor) {
    const size_t n = px.size();
    if (n == 0) return false;
    float lo[3] = {px[0], py[0], pz[0]}, hi[3] = {px[0], py[0], pz[0]};
    for (size_t i = 1; i < n; ++i) {
      lo[0] = std::min(lo[0], px[i]); hi[0] = std::max(hi[0], px[i]);
      lo[1] = std::min(lo[1], py[i]); hi[1] = std::max(hi[1], py[i]);
      lo[2] = std::min(lo[2], pz[i]); hi[2] = std::max(hi[2], pz[i]);
    }
    h = delta * (cell_factor >= kMinCellFactor ? cell_factor : kMinCellFactor);
    if (!(h > 0.f)) return false;
    while (true) {
      const double ex = (double(hi[0]) - lo[0]) / h, ey = (double(hi[1]) - lo[1]) / h, ez = (double(hi[2]) - lo[2]) / h;
      if (ex < 2.0e9 && ey < 2.0e9 && ez < 2.0e9) {
        nx = int(ex) + 4; ny = int(ey) + 4; nz = int(ez) + 4;
        // nx and ny*nz below 2^24: the kernels form the linear cell index with two 24-bit multiply-adds
        if (ncell() <= max_cells && ncell() < 0xFFFFFFF0ull && nx < (1 << 24) && uint64_t(ny) * uint64_t(nz) < (1ull << 24)) break;
      }
      h *= 1.25f;
    }
    inv_h = 1.0f / h;
    ox = lo[0] - 1.5f * h; oy = lo[1] - 1.5f * h; oz = lo[2] - 1.5f * h;
    reach = double(delta) + 0.01 * double(h);
    cshift = 0;      // coarse level: smallest shift whose bitmap fits the LDS budget (padded to 16 B for the staging)
    while (true) {
      // cnx, cny, cnz: PITCHES of the coarse bitmap = cubes per axis + one cube that no cell maps to (always 0): the lean sweep
      // of k_verify clamps a coordinate outside the grid onto it instead of testing bounds (s4p_kernels.hip.hpp)
      cnx = ((nx - 1) >> cshift) + 2; cny = ((ny - 1) >> cshift) + 2; cnz = ((nz - 1) >> cshift) + 2;
      const uint64_t cw = (uint64_t(cnx) * cny * cnz + 31) / 32;
      if (cw <= max_coarse_words) { coarse_words = uint32_t((cw + 3) & ~uint64_t(3)); break; }
      ++cshift;
    }
    return true;
  }
};

}  // namespace s4p
