// s4p_capi.hip -- context + C ABI (include/s4p_capi.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see super4pcs_amd/build.py)
#include "s4p_capi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "s4p_host_structs.hpp"
#include "s4p_kernels.hip.hpp"

using namespace s4p;

namespace {
std::string g_create_error;

template <class T>
struct DevBuf {
  T* p = nullptr; size_t n = 0;
  hipError_t alloc(size_t count) { free(); n = count; return count ? hipMalloc((void**)&p, count * sizeof(T)) : hipSuccess; }
  void free() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
};
template <class T>
struct PinBuf {
  T* p = nullptr; size_t n = 0;
  hipError_t alloc(size_t count) { free(); n = count; return count ? hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault) : hipSuccess; }
  void free() { if (p) { (void)hipHostFree(p); p = nullptr; } n = 0; }
};
uint32_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return uint32_t(p); }
uint32_t float_key(float f) { uint32_t b; std::memcpy(&b, &f, 4); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
float key_float(uint32_t k) { const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; std::memcpy(&f, &b, 4); return f; }
float angle_threshold(double theta, bool* monotone) {
  auto pass = [&](float x) { return double(std::acos(x)) <= theta; };
  *monotone = true;
  if (!pass(1.0f)) return 2.0f;                            // nothing passes (theta < 0)
  if (pass(-1.0f)) return -1.0f;                           // everything in [-1, 1] passes (theta >= pi)
  uint32_t lo = float_key(-1.0f), hi = float_key(1.0f);    // lo fails, hi passes
  while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (pass(key_float(mid))) hi = mid; else lo = mid; }
  for (uint32_t d = 0; d < 512u; ++d) {
    if (hi + d <= float_key(1.0f) && !pass(key_float(hi + d))) *monotone = false;
    if (lo - d >= float_key(-1.0f) && lo >= d && pass(key_float(lo - d))) *monotone = false;
  }
  return key_float(hi);
}
}  // namespace

// The lanes are HIP streams; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of
// streams that share a queue serialise (measured on the bench workload, round 4: 2 queues 120.7, 4 queues 176.6, 8 queues
// 180.3 M candidates/s).  The variable is read when the HIP runtime initialises, so it is the APPLICATION's to set: the
// library does not touch the process environment (a load-time setenv was a process-wide side effect on every other HIP user
// and is not thread-safe against a concurrent getenv -- ADVICE r04).  The Python entry points (super4pcs_amd/capi.py, bench.py)
// set it before HIP initialises unless S4P_KEEP_HW_QUEUES=1; INTEGRATION.md tells a C++ application to export it.

struct s4p_ctx {
  int device = 0;
  std::string err;
  s4p_options opt{};
  uint64_t max_pairs = 0, max_quads = 0, max_grid_cells = 0;
  uint64_t need_pairs = 0, need_quads = 0;      // counts of the base that overflowed last (s4p_grow_limits)
  char devname[256] = {0};

  // host mirrors
  std::vector<float> hpx, hpy, hpz;          // sampled P, original order (for base lookups)
  std::vector<float> hqx, hqy, hqz, hux, huy, huz;
  bool has_normals = false, has_rgb = false;
  uint32_t n_p = 0, n_q = 0;
  UnitFrame frame;
  PairOctree tree;
  LcpGridHost hgrid;
  float base_xyz[12] = {0}, base_nrm[12] = {0}, base_rgb[12];
  bool clouds_set = false;

  // device state
  DevBuf<uint2> greach; DevBuf<uint4> glist_hdr; DevBuf<uint32_t> gcoarse; DevBuf<float4> gnbr; DevBuf<float4> q4, q4v;
  DevBuf<uint2> qquant; QuantQ qq{}; bool qlds = false;      // 16-bit copy of the Morton-ordered queries for the LDS-resident sweep
  uint32_t verify_blocks_surv = 0;   // workgroups of k_verify when it scores the sweep's survivors (S4P_VERIFY_BLOCKS_SURV; 0: as k_sweep)
  DevBuf<float> qtiles; uint32_t tile_q = 0, n_tiles = 0; int sweep_pass_env = -1;      // k_sweep's view of the sampled Q: tiles of tile_q points (x | y | z), any sample size.  The first pass runs for samples that do not fit LDS (n_tiles > 1); S4P_SWEEP_PASS=0 / 1 forces it off / on (A/B aid)
  DevBuf<float> qsoa; bool lean = false, lean_lds = false;   // lean_lds: the float copy fits LDS (else the lean sweep reads q4v from global memory)
                      // float copy x | y | z of the same, padded (the lean sweep of k_verify: early-exit mode)
  DevBuf<float> qx, qy, qz, ux, uy, uz, qnx, qny, qnz, qcr, qcg, qcb;
  // Lanes = HIP streams with private per-base device buffers.  Consecutive bases rotate over the lanes, so the
  // small kernels of base t+1 (pairs, hash build, quad enumeration) run concurrently with the
  // LCP scoring of base t instead of leaving most of the 256 CUs idle between them.
  // per-base buffers sized by the limits (pairs: 11 arrays, quads: 5, the cell hash): allocated as a set, so that a
  // growth can build the new set first and swap it in only when every allocation of every lane has succeeded
  struct LaneBufs {
    DevBuf<int2> ab1, ab2; DevBuf<uint32_t> okey1, okey2, cell1, bucket1, next1; DevBuf<float4> ew1;
    DevBuf<int4> quads; DevBuf<unsigned long long> tags; DevBuf<uint32_t> counts, cand_idx; DevBuf<float4> cand_T, surv_T;      // surv_T: the candidates k_sweep lets through (the list k_verify scores when a bound is in force)
    DevBuf<unsigned long long> ht_keys, ht_heads; uint32_t ht_mask = 0, epoch = 0;
    uint64_t cap_pairs = 0, cap_quads = 0;    // entries these buffers hold (a lane whose base needed more has grown on its own)
    void free_all() {
      cap_pairs = cap_quads = 0;
      ab1.free(); ab2.free(); okey1.free(); okey2.free(); cell1.free(); bucket1.free(); next1.free();
      ew1.free(); quads.free(); tags.free(); counts.free(); cand_idx.free(); cand_T.free(); surv_T.free(); ht_keys.free(); ht_heads.free();
    }
  };
  // What a group launch needs of one base: the parameter records of its four kernels and the upload of its staged sequences,
  // built when the base is submitted (s4p_try_base_staged_async), consumed when its group is launched (flush_group).
  struct LaunchRec {
    PairParams2 pp; PrepParams p1; QuadParams q; BaseFrame bf; bool fused_prep = false;
    const uint32_t* up_src = nullptr; uint32_t* up_dst = nullptr; size_t up_bytes = 0;
  };
  struct Lane : LaneBufs {
    hipStream_t stream = nullptr;     // the lane's own stream: group launches run on the stream of the group's first lane, solo passes (stage-level calls, chunk passes, a base redone after a growth) on the lane's own
    DevBuf<DevCounters> ctr;          // [0] live counters of the base in flight (its result record is pinned host memory: hctr)
    LaunchRec rec;
    bool pending = false;             // submitted, its group not launched yet
    int32_t failed_rc = 0; std::string failed_msg;      // the launch of this lane's base could not be enqueued (flush_lanes): what its wait returns
    uint32_t seq = 0;                 // number of the launch whose result record the host waits for (DevCounters::seq)
    DevBuf<uint4> slots;              // k_verify: per-workgroup best, reduced by its last workgroup
    DevBuf<uint32_t> border;          // k_verify: candidates whose Euler-angle gate the host settles (max_angle >= 0), kBorderCap entries
    bool dirty = false;               // a stage-level call left the live counters non-zero: clear before a fused pass
    DevBuf<uint32_t> seqbuf;          // device copy of a staged sequence blob, both pair sets (layout: StageSlot)
    // launch record of the base in flight: what a relaunch after a buffer growth needs (finish_result)
    int sv_slot = -1; int32_t sv_ids[4] = {0, 0, 0, 0}; float sv_inv1 = 0.f, sv_inv2 = 0.f; float sv_bx[12] = {0}, sv_brgb[12] = {0};
    uint64_t sv_gen = 0;              // generation of the staging slot when the base was launched (a replay needs the same content)
    uint32_t sv_nseq1 = 0;            // sequence length of the base's first pair set: its order keys are below 2 * n_q * sv_nseq1
  };
  static constexpr int kMaxLanes = 24;
  Lane lane[kMaxLanes];
  // Bases in flight = lanes (S4P_LANES, 1..24); consecutive lanes form GROUPS of `group` bases (S4P_GROUP, 1..kGroupMax) that
  // go through every kernel in ONE launch (s4p_kernels.hip.hpp "BASE GROUPS").  A group is launched when its last base has been
  // submitted -- or earlier, with the bases it has, when somebody waits for one of them -- so any call pattern (one base at a
  // time, the engine's pipelined loop, the sharded loops) gets the same results; only the packing differs.
  // Measured on the bench workload (round 5, one box per row, M candidates/s): 6 lanes x 1: 185 (the round-4 shape); 9 x 3: 200;
  // 12 x 3: 206; 12 x 2: 225; 14 x 2: 228-236; 16 x 2: 212-229; 12 x 1 (twelve streams on 8 hardware queues): 111 -- the number of
  // STREAMS in use should not exceed GPU_MAX_HW_QUEUES (8); k_verify of a group on a lower-priority stream of its own: 20 (!).
  int n_lanes = 14, group = 2;
  // Grids of k_prep / k_quads: their trip counts (pairs of a base) live in device memory, so the grids are sized by what the
  // registration's bases have needed so far (a decaying maximum with head-room) instead of the worst case -- a grid of 1024 /
  // 2048 workgroups per base of which a hundred find work is mostly dispatch cost.  A base that needs more takes a second
  // grid-stride pass: slower, same result.  0 = no estimate yet (first bases, stage-level calls): the full grids.
  uint32_t est_m1 = 0, est_m2 = 0;
  bool two_prio = false; int prio_hi = 0, prio_mid = 0;      // group streams on two priority levels (s4p_create: fewer hardware queues than streams)
  bool fuse_prep = true;             // S4P_FUSE_PREP=0: always the k_prep launch (A/B aid)
  uint32_t pair_split = 2;           // waves that share one (tile, chunk) item of k_pairs2 (S4P_PAIR_SPLIT: 1, 2, 4)
  uint64_t prep_redos = 0;           // bases redone because the estimate-sized cell hash was too small
  uint32_t launch_seq = 0;           // group launches so far (written into the result records: DevCounters::seq)
  DevBuf<uint32_t> group_done;       // one k_verify ticket counter per lane (a launch uses the one of its first lane)
  // Result records: pinned host memory the last workgroup of k_verify writes directly (no read-back copy in the stream)
  struct HostRec { DevCounters* p = nullptr; DevCounters* dev = nullptr; size_t n = 0;
    hipError_t alloc(size_t) { free(); hipError_t e = hipHostMalloc((void**)&p, sizeof(DevCounters), hipHostMallocMapped | hipHostMallocCoherent); if (e != hipSuccess) { p = nullptr; return e; }
      n = 1; std::memset(p, 0, sizeof(DevCounters)); return hipHostGetDevicePointer((void**)&dev, p, 0); }
    void free() { if (p) { (void)hipHostFree(p); p = nullptr; dev = nullptr; } n = 0; } };
  HostRec hctr[kMaxLanes];           // [pipeline slot == lane]
  // Staging ring: host-built octree sequences of the two pair sets of a base, in pinned memory.  A slot is
  // written by whoever stages the base (the caller thread, or the engine's octree thread through s4p_stage_base)
  // and read by the H2D copies of s4p_try_base_staged_async; the engine recycles a slot after that base's wait.
  struct StageSlot {
    // one blob per pair set: seq_id[n_seq] | leaf_off[n_leaf + 1] | pad to 16 B | leaves[n_leaf]; set 1 right behind set 0
    // (off[1], a multiple of 4 words) so that ONE copy uploads both
    PinBuf<uint32_t> blob;
    uint32_t off[2] = {0, 0};
    uint32_t n_seq[2] = {0, 0}, n_leaf[2] = {0, 0};
    static uint32_t set_words(uint32_t n_seq, uint32_t n_leaf) { return (leaf_word(n_seq, n_leaf) + 4u * n_leaf + 3u) & ~3u; }
    uint64_t gen = 0;                 // bumped whenever the slot is (re)written
    static uint32_t leaf_word(uint32_t n_seq, uint32_t n_leaf) { return (n_seq + n_leaf + 1u + 3u) & ~3u; }
    static size_t blob_words(size_t n_q) { return 2 * n_q + 8 + 4 * n_q; }      // n_leaf <= n_seq <= n_q
    float eps_unit[2] = {0, 0}, n_radius[2] = {0, 0}, distance[2] = {0, 0}, normal_angle[2] = {0, 0};
  };
  static constexpr int kStageSlots = 56;   // 0..27: self-staging of s4p_try_base_async; 28..55: a threaded driver
  StageSlot stage[kStageSlots];
  uint32_t stage_rr = 0;             // round-robin slot for the self-staging (single-thread) paths
  int cur = 0;                       // slot used by the call in progress
  uint32_t q_head = 0, q_tail = 0;   // async FIFO of s4p_try_base_async (depth 2)
  BaseFrame slot_bf[kMaxLanes];
  QuadParams slot_q[kMaxLanes];      // enumeration record of the base in flight on each lane (the chunk loop relaunches it per range)
  PinBuf<uint32_t> hmm[kMaxLanes];   // pinned {m1, m2} of a chunked base: restored before every chunk pass
  hipEvent_t done[kMaxLanes] = {};
  // Bases whose congruent quads do not fit max_quads are processed in CHUNKS (run_chunked): quads of a range of set-2
  // entries -> gate -> score -> fold, range after range, instead of failing with S4P_ERR_CAPACITY.  The reference's
  // std::vector<Quadrilateral> simply grows (super4pcs.cc:166-174, match4pcsBase.hpp:340-351); at the 20 000-point sample
  // of SURVEY 8d a base has ~10^9 quads.  quad_grow_cap: the quad capacity never grows beyond this many entries.
  bool chunking = true; uint64_t quad_grow_cap = 32ull << 20;
  // A lane whose base overflowed a PAIR buffer (or, without chunking, the quad buffers) grows its own buffers to what the
  // base's counters ask for and runs the base again, inside the wait (finish_result): what the reference's std::vectors do.
  // No other lane, no host state (RNG stream, octree permutation) is involved, so single-GPU and sharded loops alike go on
  // as if the buffers had been large enough.  Off: S4P_ERR_CAPACITY, the stage-level contract.
  bool auto_grow = true; uint64_t lane_growths = 0;
  // s4p_set_best_hint: candidates that cannot EXCEED this inlier count may be abandoned by k_verify (0 = count all in full)
  uint32_t best_hint = 0;
  // s4p_set_quad_slice: this context enumerates only its share of every base's second pair set (SURVEY 8e level 2)
  uint32_t slice_num = 0, slice_den = 0;
  // max_angle (shared4pcs.h:160): > 0 -> the segment-angle pair filter through an exact cosine threshold; >= 0 -> the
  // Euler-angle bound of ComputeRigidTransformation, decided on the device up to a margin and settled on the host
  float cos_min = -1.f; bool angle_pairs = false; float angle_tol = 1e-6f;   // S4P_ANGLE_TOL (read at creation) widens the device margin: a test aid
  uint64_t border_settled = 0, border_rejected = 0;
  std::vector<uint32_t> border_failed;   // quads of the last pass whose undecided gate the host rejected (per-candidate outputs say -1 for them)
  bool last_chunked = false;         // the last base took several device passes (chunked fused base, sliced s4p_try_congruent_set)
  // Per-candidate records of such a base.  With keep_records (s4p_keep_candidate_records) or a sink (s4p_set_candidate_sink)
  // the passes run in REFERENCE ORDER -- chunks are ranges of the set-1 order key, the primary key of the std::set order
  // (super4pcs.cc:127,166) -- and every pass's records are read back, sorted by tag and appended to `kept` / handed to the
  // sink, so the reference's per-candidate visitor calls (match4pcsBase.hpp:458-465) and the list-returning debug calls work
  // at any size.  Without either, s4p_last_candidates / s4p_last_verified replay the base once in that mode.
  bool keep_records = false;
  s4p_candidate_sink sink = nullptr; void* sink_user = nullptr;
  struct Kept {
    bool valid = false;
    std::vector<int32_t> quads, qcounts;       // every quad of the base in reference order, -1 = gate failed
    std::vector<uint32_t> counts; std::vector<float> T16;     // the verified candidates in reference order
    void clear() { valid = false; quads.clear(); qcounts.clear(); counts.clear(); T16.clear(); }
  } kept;
  bool capturing() const { return keep_records || sink != nullptr; }
  bool broken = false;               // a growth failed half-way: the lane buffers are inconsistent, every pass is refused
  uint64_t chunk_bases = 0, chunk_passes = 0, chunk_splits = 0, chunk_quads = 0;
  DevBuf<float> tbuf; size_t tbuf_cap = 0;      // s4p_transform_points: two device + two pinned staging chunks
  PinBuf<float> tpin; hipEvent_t tev[2] = {nullptr, nullptr};
  // base selection on the device (s4p_select_base_points): the sampled P as float4 records in sampling order, one
  // attempt's 2001 draws and its result record; a stream of its own, so a selector thread never queues behind a lane
  DevBuf<float4> p4o; DevBuf<uint32_t> sel_draws; DevBuf<SelectRecord> sel_rec;
  PinBuf<uint32_t> sel_hdraws; PinBuf<SelectRecord> sel_hrec; hipStream_t sel_stream = nullptr;

  // profiling
  bool prof_events = false, prof_stages = false, prof_points = false;      // events around k_verify / around the other stages too / instrumented kernel
  hipEvent_t ev[kMaxLanes][6] = {};
  s4p_profile prof{};
  uint64_t last_K = 0;
  uint32_t verify_blocks = 256; bool verify_blocks_fixed = false, verify_blocks_env = false;   // per set_clouds (see there); S4P_VERIFY_BLOCKS fixes it
  int verify_threads = kVerifyThreadsCached;      // per set_clouds; S4P_VERIFY_THREADS overrides
  int ablate = 0;                    // S4P_ABLATE (profiling aid, read once at creation)
  double host_octree_s = 0, host_wait_s = 0;
  // S4P_TRACE_LAUNCH=1 (lab aid): where the launch thread's time goes inside flush_lanes, printed by s4p_destroy
  double wait_timeout_s = 600.0;      // S4P_WAIT_TIMEOUT_S: a device pass that has not finished by then is reported as an error (wait_lane)
  bool debug = false;                // S4P_DEBUG=1 (lab aid): launches and waits on stderr
  bool trace_launch = false; double lt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t lt_n = 0, lt_groups = 0;
  int lane_group_n[kMaxLanes] = {0};  // bases of the launch a lane was the FIRST lane of (its profiling events), else 0
  bool ev_pending[kMaxLanes] = {false}, ev_fused[kMaxLanes] = {false};      // events of that launch not read yet (harvest_events)
  double set_clouds_s[4] = {0, 0, 0, 0};      // last s4p_set_clouds: host copies + unit frame + grid plan | device build of the LCP structure | Q-side uploads | total

  size_t verify_lds_bytes() const {
    return gcoarse.n * 4 + (qlds ? size_t((n_q + kSweepStep - 1u) & ~(kSweepStep - 1u)) * 8 : 0) + size_t(verify_threads / 64) * kQueueWordsPerWave * 4 + sizeof(VerifyShared);
  }
  size_t lean_lds_bytes() const {
    return gcoarse.n * 4 + (lean_lds ? size_t((n_q + kSweepStep - 1u) & ~(kSweepStep - 1u)) * 12 : 0) + size_t(verify_threads / 64) * kLeanQueue * 2 + sizeof(VerifyShared);
  }
  bool use_lean() const { return lean && best_hint != 0u; }
  size_t sweep_lds_bytes() const { return gcoarse.n * 4 + size_t(tile_q) * 12 + sizeof(SweepShared); }
  // (a chunk pass scores ~10^7 candidates with the chip to itself: two workgroups per CU, as for the HBM-bound structure;
  // measured at the 20 000-point sample: 0.50 s per pass with 512 workgroups, 0.66 s with 256)
  bool chunk_pass = false;
  uint32_t verify_grid() const { return (chunk_pass && !verify_blocks_env) ? std::max(verify_blocks, 512u) : verify_blocks; }
  LcpGrid dev_grid() const {
    LcpGrid g;
    g.reach = greach.p; g.list_hdr = glist_hdr.p; g.nbr = gnbr.p;
    g.coarse = gcoarse.p; g.coarse_words = uint32_t(gcoarse.n);
    g.cshift = hgrid.cshift; g.cnx = hgrid.cnx; g.cny = hgrid.cny;
    g.ox = hgrid.ox; g.oy = hgrid.oy; g.oz = hgrid.oz; g.inv_h = hgrid.inv_h;
    g.nx = hgrid.nx; g.ny = hgrid.ny; g.nz = hgrid.nz;
    g.sq_eps = opt.delta * opt.delta;     // match4pcsBase.cc:517,522
    return g;
  }
};

#define S4P_FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); \
    return (e_ == hipErrorOutOfMemory) ? S4P_ERR_OOM : S4P_ERR_HIP; } } while (0)

namespace {

int32_t check_overflow(s4p_ctx* c, const DevCounters& d) {
  const uint32_t ov = d.overflow;
  if (!ov) return S4P_OK;
  // the device counters keep counting past the capacity, so they say what this base needs (the quads only once the
  // pairs fit: with truncated pair lists K is a lower bound)
  c->need_pairs = std::max<uint64_t>(d.m1, d.m2); c->need_quads = d.K;
  if (!(ov & 3u)) c->need_pairs = 0;                      // the pair lists fitted: only the quads ask for more
  char b[160];
  snprintf(b, sizeof b, "device buffer overflow (bits=%u: 1=pairs1 2=pairs2 4=quads); raise s4p_limits (max_pairs=%llu max_quads=%llu)",
           ov, (unsigned long long)c->lane[c->cur].cap_pairs, (unsigned long long)c->lane[c->cur].cap_quads);
  c->err = b;
  return S4P_ERR_CAPACITY;
}

// Host side of ExtractPairs (super4pcs.cc:193-217), part 1: functor radius/epsilon, octree loop 1, flat sequence
// into a pinned staging slot.  Host-only (touches c->tree and the slot): may run on the engine's octree thread.
void stage_pairs(s4p_ctx* c, int slot, int set, float pair_distance, float pair_normals_angle, float pair_distance_epsilon, bool copy) {
  s4p_ctx::StageSlot& st = c->stage[slot];
  const float nRadius = pair_distance / c->frame.ratio;                         // setRadius, pairCreationFunctor.h:124-129
  const float eps_n = pair_distance_epsilon / c->frame.ratio;                  // getNormalizedEpsilon, :131-133
  { auto t0 = std::chrono::steady_clock::now();
    c->tree.build(c->hux.data(), c->huy.data(), c->huz.data(), c->n_q, nRadius, eps_n, 50);
    c->host_octree_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  if (!copy) return;
  st.n_seq[set] = c->tree.n_seq(); st.n_leaf[set] = c->tree.n_leaf(); st.gen++;
  st.eps_unit[set] = c->tree.eps_unit; st.n_radius[set] = nRadius; st.distance[set] = pair_distance; st.normal_angle[set] = pair_normals_angle;
  static_assert(sizeof(Leaf) == sizeof(float4), "leaf records are uploaded as float4");
  // set 0 at the start of the slot's blob, set 1 right behind it (set 0 of a base is always staged first)
  st.off[set] = set == 0 ? 0u : s4p_ctx::StageSlot::set_words(st.n_seq[0], st.n_leaf[0]);
  uint32_t* blob = st.blob.p + st.off[set];
  c->tree.flatten(blob, blob + st.n_seq[set], reinterpret_cast<Leaf*>(blob + s4p_ctx::StageSlot::leaf_word(st.n_seq[set], st.n_leaf[set])));
}

// part 2: the kernel parameters of one staged set on lane c->cur (the upload of the blob is the caller's: one copy per base)
void fill_pair_params(s4p_ctx* c, int slot, int set, float pair_distance_epsilon, int bp1, int bp2, PairParams& P) {
  const s4p_ctx::StageSlot& st = c->stage[slot];
  s4p_ctx::Lane& L = c->lane[c->cur];
  const uint32_t n_seq = st.n_seq[set], n_leaf = st.n_leaf[set];
  const uint32_t leaf_word = s4p_ctx::StageSlot::leaf_word(n_seq, n_leaf);
  uint32_t* dseq = L.seqbuf.p + st.off[set];
  P = PairParams{};
  P.ux = c->ux.p; P.uy = c->uy.p; P.uz = c->uz.p; P.qx = c->qx.p; P.qy = c->qy.p; P.qz = c->qz.p;
  P.nx = c->has_normals ? c->qnx.p : nullptr; P.ny = c->qny.p; P.nz = c->qnz.p;
  P.cr = c->has_rgb ? c->qcr.p : nullptr; P.cg = c->qcg.p; P.cb = c->qcb.p;
  P.seq_id = dseq; P.n_seq = n_seq; P.leaf_off = dseq + n_seq; P.leaves = reinterpret_cast<const float4*>(dseq + leaf_word); P.n_leaf = n_leaf;
  P.n_q = c->n_q; P.nRadius = st.n_radius[set]; P.eps_unit = st.eps_unit[set];
  P.pair_distance = st.distance[set]; P.pair_distance_eps = pair_distance_epsilon; P.pair_normals_angle = st.normal_angle[set];
  P.max_normal_difference = c->opt.max_normal_difference; P.max_color_distance = c->opt.max_color_distance;
  P.max_translation_distance = c->opt.max_translation_distance;
  P.norm_threshold = float(0.5 * double(c->opt.max_normal_difference) * M_PI / 180.0);   // pairCreationFunctor.h:169-170
  for (int k = 0; k < 3; ++k) {
    P.b1pos[k] = c->base_xyz[3 * bp1 + k]; P.b2pos[k] = c->base_xyz[3 * bp2 + k];
    P.b1rgb[k] = c->base_rgb[3 * bp1 + k]; P.b2rgb[k] = c->base_rgb[3 * bp2 + k];
  }
  P.ab = set == 0 ? L.ab1.p : L.ab2.p; P.okey = set == 0 ? L.okey1.p : L.okey2.p;
  P.counter = set == 0 ? &L.ctr.p->m1 : &L.ctr.p->m2;
  P.cap = uint32_t(L.cap_pairs); P.overflow = &L.ctr.p->overflow; P.overflow_bit = set == 0 ? 1u : 2u;
  P.split = c->pair_split;
  { float sx = c->base_xyz[3 * bp2] - c->base_xyz[3 * bp1], sy = c->base_xyz[3 * bp2 + 1] - c->base_xyz[3 * bp1 + 1],
          sz = c->base_xyz[3 * bp2 + 2] - c->base_xyz[3 * bp1 + 2];                 // setBase, pairCreationFunctor.h:135-143
    normalize3(sx, sy, sz);
    P.seg1[0] = sx; P.seg1[1] = sy; P.seg1[2] = sz; P.cos_min = c->cos_min; }
}

// loop 2 + the pair filters (k_pairs2) of the first n_bases records of PG, n_sets sets each, in one launch: one wave per
// (tile of 64 primitives, chunk of 64 sequence slots); persistent 512-thread workgroups, at most two waves per SIMD over the
// two sets of a base
int32_t launch_pairs_kernel(s4p_ctx* c, const PairGroup& PG, int n_bases, int n_sets, hipStream_t st) {
  uint32_t n_seq_max = 0;
  for (int b = 0; b < n_bases; ++b) for (int k = 0; k < n_sets; ++k) n_seq_max = std::max(n_seq_max, PG.base[b].set[k].pair.n_seq);
  if (n_seq_max == 0) return S4P_OK;
  const uint64_t items = uint64_t((c->n_q + 63u) / 64u) * uint64_t((n_seq_max + 63u) / 64u) * c->pair_split;
  const uint32_t wgs = uint32_t(std::min<uint64_t>(std::max<uint64_t>((items + kPair2Waves - 1u) / kPair2Waves, 1u), (n_sets == 2 ? 128u : 256u) * c->pair_split));
  const dim3 grid(wgs, uint32_t(n_sets == 2 ? 2 * n_bases : 1));
  if (c->angle_pairs) hipLaunchKernelGGL(k_pairs2<true>, grid, dim3(64 * kPair2Waves), 0, st, PG);
  else hipLaunchKernelGGL(k_pairs2<false>, grid, dim3(64 * kPair2Waves), 0, st, PG);
  HIPCHK(c, hipGetLastError());
  return S4P_OK;
}

// one set through the stage-level entry point (s4p_extract_pairs): staged, uploaded and extracted on lane c->cur's own stream
int32_t launch_pairs(s4p_ctx* c, int set, float pair_distance, float pair_normals_angle, float pair_distance_epsilon,
                     int bp1, int bp2) {
  const int slot = int(c->stage_rr % uint32_t(s4p_ctx::kStageSlots));
  stage_pairs(c, slot, set, pair_distance, pair_normals_angle, pair_distance_epsilon, true);
  PairGroup PG{};
  fill_pair_params(c, slot, set, pair_distance_epsilon, bp1, bp2, PG.base[0].set[0].pair);
  const s4p_ctx::StageSlot& st = c->stage[slot];
  s4p_ctx::Lane& L = c->lane[c->cur];
  if (st.n_seq[set] == 0) return S4P_OK;
  HIPCHK(c, hipMemcpyAsync(L.seqbuf.p + st.off[set], st.blob.p + st.off[set], size_t(s4p_ctx::StageSlot::set_words(st.n_seq[set], st.n_leaf[set])) * 4, hipMemcpyHostToDevice, L.stream));
  return launch_pairs_kernel(c, PG, 1, 1, L.stream);
}

// segment lengths / normal "angles" of an ordered base (match4pcsBase.hpp:318-326)
inline float seg_len(const float* v, int a, int b) {
  const float d0 = v[3 * a] - v[3 * b], d1 = v[3 * a + 1] - v[3 * b + 1], d2 = v[3 * a + 2] - v[3 * b + 2];
  return std::sqrt(d0 * d0 + (d1 * d1 + d2 * d2));
}

// IndexedNormalSet ctor (normalset.h:114-124) + getNeighbors constants (normalset.hpp:174-191)
void quad_setup(const s4p_ctx* c, float distance_threshold2, QuadGrid& qg, ConeTable& cone) {
  const float eps = distance_threshold2 / c->frame.ratio;                       // super4pcs.cc:114
  const int gridDepth = int(-std::log2(eps));
  qg.egSize = int(std::pow(2, gridDepth));
  qg.gepsilon = 1.f / float(qg.egSize);
  qg.nepsilon = float(double(1.f / 7.f) + 0.00001);
  // cos(alpha) of the two base segments, super4pcs.cc:109-111
  float a[3], b[3];
  for (int k = 0; k < 3; ++k) { a[k] = c->base_xyz[3 + k] - c->base_xyz[k]; b[k] = c->base_xyz[9 + k] - c->base_xyz[6 + k]; }
  auto nrm = [](float* v) { float s2 = v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); if (s2 > 0.f) { float s = std::sqrt(s2); v[0] /= s; v[1] /= s; v[2] /= s; } };
  nrm(a); nrm(b);
  const float cosAlpha = a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
  const float alpha = std::acos(cosAlpha);
  const float perimeter = float(double(2.f) * M_PI * double(std::atan(alpha)));
  const float fnb = 2.f * std::ceil(perimeter * 7.f / 2.f);
  unsigned nb = (fnb == fnb && fnb > 0.f) ? unsigned(fnb) : 0u;                 // NaN (|cos|>1) -> no samples
  if (nb > unsigned(kMaxConeSamples)) nb = unsigned(kMaxConeSamples);           // cannot exceed 56 for alpha in [0,pi]
  const float angleStep = float(double(2.f) * M_PI / double(float(nb)));
  const float sinAlpha = std::sin(alpha);
  cone.nb = int(nb);
  for (unsigned s = 0; s < nb; ++s) {
    const float theta = float(s) * angleStep;
    cone.v[s][0] = sinAlpha * std::cos(theta);
    cone.v[s][1] = sinAlpha * std::sin(theta);
    cone.v[s][2] = cosAlpha;
  }
}

BaseFrame make_base_frame(const s4p_ctx* c, const int32_t* base_ids) {
  BaseFrame b{};
  for (int i = 0; i < 3; ++i) { b.p[i][0] = c->hpx[base_ids[i]]; b.p[i][1] = c->hpy[base_ids[i]]; b.p[i][2] = c->hpz[base_ids[i]]; }
  for (int k = 0; k < 3; ++k) b.c1[k] = ((b.p[0][k] + b.p[1][k]) + b.p[2][k]) / 3.f;   // match4pcsBase.hpp:385
  b.gate = 2.0f * c->opt.delta;                                                        // distance_factor * delta
  b.max_angle_rad = float(double(c->opt.max_angle) * std::acos(-1.0) / 180.0);         // match4pcsBase.hpp:392,426
  b.angle_gate = c->opt.max_angle >= 0.f ? 1 : 0;                                      // match4pcsBase.cc:457
  b.angle_tol = c->angle_tol;
  return b;
}

// Parameters of FindCongruentQuadrilaterals for the lane's current base: a fresh hash epoch, the per-set preparation
// records (consumed by k_pairs on the fused path, by k_prep otherwise) and the enumeration record.
int32_t quad_params(s4p_ctx* c, float inv1, float inv2, float thr2, PrepParams& P1, QuadParams& Q) {
  s4p_ctx::Lane& L = c->lane[c->cur];
  QuadGrid qg; ConeTable cone;
  quad_setup(c, thr2, qg, cone);
  if (qg.egSize > 1024) S4P_FAIL(c, S4P_ERR_UNSUPPORTED, "FindCongruentQuadrilaterals grid finer than 1024^3 cells (delta/extent too small)");
  L.epoch++;
  if (L.epoch == 0xFFFFFFFFu) {   // wrap, once every 4e9 bases of a lane: clear the table with nothing in flight anywhere (the base is launched on its group's stream, not necessarily the lane's: ADVICE r05)
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemset(L.ht_keys.p, 0, L.ht_keys.n * 8));
    HIPCHK(c, hipMemset(L.ht_heads.p, 0, L.ht_heads.n * 8));
    L.epoch = 1;
  }
  HashTable ht{L.ht_keys.p, L.ht_heads.p, L.ht_mask, L.epoch, &L.ctr.p->m1, uint32_t(L.cap_pairs), 0u};
  P1 = PrepParams{};
  P1.ux = c->ux.p; P1.uy = c->uy.p; P1.uz = c->uz.p; P1.qx = c->qx.p; P1.qy = c->qy.p; P1.qz = c->qz.p;
  P1.ab = L.ab1.p; P1.m_dev = &L.ctr.p->m1; P1.cap = uint32_t(L.cap_pairs); P1.invariant = inv1; P1.qg = qg;
  P1.cell = L.cell1.p; P1.bucket = L.bucket1.p; P1.ew = L.ew1.p; P1.next = L.next1.p; P1.ht = ht;
  Q = QuadParams{};
  Q.ab1 = L.ab1.p; Q.okey1 = L.okey1.p; Q.bucket1 = L.bucket1.p; Q.ew1 = L.ew1.p; Q.next1 = L.next1.p;
  Q.ab2 = L.ab2.p; Q.okey2 = L.okey2.p;
  Q.ux = c->ux.p; Q.uy = c->uy.p; Q.uz = c->uz.p; Q.qx = c->qx.p; Q.qy = c->qy.p; Q.qz = c->qz.p;
  Q.invariant2 = inv2; Q.qg = qg; Q.cone = cone;
  Q.m2_dev = &L.ctr.p->m2; Q.cap2 = uint32_t(L.cap_pairs); Q.ht = ht; Q.thr = thr2;
  Q.quads = L.quads.p; Q.tags = L.tags.p; Q.K_dev = &L.ctr.p->K; Q.K_cap = uint32_t(L.cap_quads); Q.overflow = &L.ctr.p->overflow;
  Q.r0 = 0u; Q.r1 = 0xFFFFFFFFu; Q.qsum_dev = &L.ctr.p->quad_sum; Q.csum_dev = &L.ctr.p->cand_sum;
  Q.slice_num = 0u; Q.slice_den = 0u;                       // (a share of the set, s4p_set_quad_slice, applies to the fused passes only: prepare_base)
  Q.k1_lo = 0u; Q.k1_hi = 0xFFFFFFFFu; Q.k1_all = 1;
  Q.do_gate = 0;
  return S4P_OK;
}

GateParams gate_params(s4p_ctx* c, const BaseFrame& bf) {
  s4p_ctx::Lane& L = c->lane[c->cur];
  GateParams G{};
  G.q4 = c->q4.p; G.base = bf; G.counts = L.counts.p; G.cand_idx = L.cand_idx.p; G.cand_T = L.cand_T.p; G.C_dev = &L.ctr.p->C;
  return G;
}

void harvest_events(s4p_ctx* c, int li);

// Group launches: the first n records of a group, on stream st.
uint32_t est_grid(uint32_t est, uint32_t full, uint32_t threads = 256u) {      // workgroups for an estimated entry count (+50 %), within [64, full]
  if (est == 0u) return full;
  const uint64_t want = (uint64_t(est) * 3u / 2u + threads - 1u) / threads;
  return uint32_t(std::min<uint64_t>(full, std::max<uint64_t>(64u, want)));
}
void launch_prep_group(const PrepGroup& G, int n, hipStream_t st, uint32_t est = 0) {
  hipLaunchKernelGGL(k_prep, dim3(est_grid(est, 1024u), uint32_t(n)), dim3(256), 0, st, G);      // set 1: hash build (set 2 is prepared inside k_quads)
}
void launch_quads_group(s4p_ctx* c, const QuadGroup& G, int n, hipStream_t st, uint32_t est = 0) {
  // one set-2 entry per thread in ONE pass for up to 512 k entries (a second grid-stride pass doubles the chain of
  // dependent gathers of the workgroups that get one); idle workgroups leave after reading the count
  uint64_t span = 1;
  for (int b = 0; b < n; ++b) span = std::max<uint64_t>(span, uint64_t(G.base[b].r1) - uint64_t(G.base[b].r0));    // (the whole set: 2^32 - 1)
  const uint32_t full = 2048u * 256u / uint32_t(kQuadThreads);
  const uint32_t blocks = std::min(est_grid(est, full, uint32_t(kQuadThreads)), uint32_t(std::min<uint64_t>(full, std::max<uint64_t>(1u, (span + kQuadThreads - 1u) / kQuadThreads))));
  if (c->opt.max_angle >= 0.f) hipLaunchKernelGGL(k_quads<true>, dim3(blocks, uint32_t(n)), dim3(kQuadThreads), 0, st, G);
  else hipLaunchKernelGGL(k_quads<false>, dim3(blocks, uint32_t(n)), dim3(kQuadThreads), 0, st, G);
}
// single-base forms on lane c->cur's own stream (stage-level calls, chunk passes)
void launch_prep_kernel(s4p_ctx* c, const PrepParams& P1) {
  PrepGroup G{}; G.base[0] = P1;
  launch_prep_group(G, 1, c->lane[c->cur].stream);
}
void launch_quads_kernel(s4p_ctx* c, const QuadParams& Q) {
  QuadGroup G{}; G.base[0] = Q;
  launch_quads_group(c, G, 1, c->lane[c->cur].stream);
}
void launch_gate_kernel(s4p_ctx* c, const GateParams& G) {
  s4p_ctx::Lane& L = c->lane[c->cur];
  GateKernelParams K{G, L.quads.p, L.tags.p, &L.ctr.p->K, uint32_t(L.cap_quads)};   // (K: 64-bit counter)
  if (c->opt.max_angle >= 0.f) hipLaunchKernelGGL(k_gate<true>, dim3(1024), dim3(256), 0, L.stream, K);
  else hipLaunchKernelGGL(k_gate<false>, dim3(1024), dim3(256), 0, L.stream, K);
}

// Verify of every gated candidate of the bases on lanes[0..n) + winner selection + result records (k_verify), on stream vs,
// bracketed by the profiling events of the first lane.  Every lane's result record gets the number of this launch.
int32_t launch_verify_group(s4p_ctx* c, const int* lanes, int n, hipStream_t vs) {
  VerifyParams V{};
  V.grid = c->dev_grid(); V.q4 = c->q4.p; V.q4v = c->q4v.p; V.qq = c->qq; V.qsoa = c->qsoa.p; V.n_q = c->n_q;
  V.n_bases = uint32_t(n);
  const uint32_t seq = ++c->launch_seq ? c->launch_seq : ++c->launch_seq;      // (never 0: a fresh record reads 0)
  for (int b = 0; b < n; ++b) {
    s4p_ctx::Lane& L = c->lane[lanes[b]];
    VerifyBase& B = V.b[b];
    B.base = c->slot_bf[lanes[b]];
    B.quads = L.quads.p; B.tags = L.tags.p; B.counts = L.counts.p; B.cand_idx = L.cand_idx.p; B.cand_T = L.cand_T.p; B.surv_T = L.surv_T.p;
    B.ctr = L.ctr.p; B.res = c->hctr[lanes[b]].dev; B.slots = L.slots.p; B.border = L.border.p;
    L.seq = seq;
  }
  V.group_done = c->group_done.p + lanes[0];
  V.seq = seq;
  V.count_tests = c->prof_points ? 1 : 0;
  V.prune = c->best_hint;
  V.ablate = c->ablate;
  harvest_events(c, lanes[0]);
  if (c->prof_events) HIPCHK(c, hipEventRecord(c->ev[lanes[0]][0], vs));
  const bool lean = c->use_lean();                           // a bound is in force: the lean sweep (s4p_kernels.hip.hpp)
  const size_t lds = lean ? c->lean_lds_bytes() : c->verify_lds_bytes();
  const dim3 grid(c->verify_grid()), block(c->verify_threads);
  if (lean && c->qtiles.p && (c->sweep_pass_env >= 0 ? c->sweep_pass_env != 0 : c->n_tiles > 1u)) {      // first pass: the coarse count of every candidate, survivors -> surv_T (k_sweep)
    SweepParams W{};
    W.grid = V.grid; W.qtiles = c->qtiles.p; W.n_q = c->n_q; W.tile_q = c->tile_q; W.n_tiles = c->n_tiles; W.n_bases = uint32_t(n); W.prune = c->best_hint;
    for (int b = 0; b < n; ++b) { s4p_ctx::Lane& L = c->lane[lanes[b]]; W.b[b] = SweepBase{L.cand_T.p, L.surv_T.p, L.ctr.p, L.counts.p}; }
    hipLaunchKernelGGL(k_sweep, grid, block, c->sweep_lds_bytes(), vs, W);
    V.use_surv = 1u;
  }
  const dim3 vgrid((V.use_surv && c->verify_blocks_surv && !c->chunk_pass) ? c->verify_blocks_surv : grid.x);
  if (lean && c->lean_lds) { if (c->prof_points) hipLaunchKernelGGL((k_verify<true, true, true>), vgrid, block, lds, vs, V); else hipLaunchKernelGGL((k_verify<false, true, true>), vgrid, block, lds, vs, V); }
  else if (lean) { if (c->prof_points) hipLaunchKernelGGL((k_verify<true, false, true>), vgrid, block, lds, vs, V); else hipLaunchKernelGGL((k_verify<false, false, true>), vgrid, block, lds, vs, V); }
  else if (c->prof_points) { if (c->qlds) hipLaunchKernelGGL((k_verify<true, true, false>), grid, block, lds, vs, V); else hipLaunchKernelGGL((k_verify<true, false, false>), grid, block, lds, vs, V); }
  else { if (c->qlds) hipLaunchKernelGGL((k_verify<false, true, false>), grid, block, lds, vs, V); else hipLaunchKernelGGL((k_verify<false, false, false>), grid, block, lds, vs, V); }
  if (c->prof_events) HIPCHK(c, hipEventRecord(c->ev[lanes[0]][1], vs));
  HIPCHK(c, hipGetLastError());
  if (c->debug) fprintf(stderr, "[s4p] k_verify launched: seq %u, %d base(s), first lane %d, lean %d, lds %zu, grid %u x %d, prune %u\n", seq, n, lanes[0], int(lean), lds, grid.x, c->verify_threads, V.prune);
  return S4P_OK;
}
// the base on lane c->cur alone, on the lane's own stream (chunk passes, s4p_try_congruent_set)
int32_t launch_verify(s4p_ctx* c, const BaseFrame& bf) {
  c->slot_bf[c->cur] = bf;
  const int lane = c->cur;
  c->lane_group_n[lane] = 1;
  return launch_verify_group(c, &lane, 1, c->lane[c->cur].stream);
}

// mark the completion of the pass of lane c->cur on its own stream (the result record itself is written by k_verify)
int32_t enqueue_result(s4p_ctx* c, const BaseFrame& bf) {
  c->slot_bf[c->cur] = bf;
  HIPCHK(c, hipEventRecord(c->done[c->cur], c->lane[c->cur].stream));
  return S4P_OK;
}

// Wait for the result record of lane li: the host polls the launch number k_verify writes last into the pinned record (a
// few microseconds sooner than the stream's event, and no read-back copy in the stream), with the event as the fallback and
// as the carrier of asynchronous errors.
int32_t wait_lane(s4p_ctx* c, int li) {
  if (c->lane[li].failed_rc != S4P_OK) {                     // its launch was never enqueued in full: nothing to wait for, nothing to trust
    const int32_t rc = c->lane[li].failed_rc;
    c->err = c->lane[li].failed_msg; c->lane[li].failed_rc = S4P_OK;
    return rc;
  }
  const volatile uint32_t* seq = &c->hctr[li].p->seq;
  const uint32_t want = c->lane[li].seq;
  const auto t0 = std::chrono::steady_clock::now();
  auto waited = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (c->debug) fprintf(stderr, "[s4p] wait lane %d for seq %u (record has %u)\n", li, want, unsigned(*seq));
  for (uint32_t spin = 0;; ++spin) {
    if (*seq == want) { std::atomic_thread_fence(std::memory_order_acquire); return S4P_OK; }
    if ((spin & 1023u) == 1023u) {
      const hipError_t q = hipEventQuery(c->done[li]);
      if (q == hipSuccess) break;                            // the launch is over: the record is complete whatever the poll saw
      if (q != hipErrorNotReady) { c->err = std::string("hipEventQuery: ") + hipGetErrorString(q); return S4P_ERR_HIP; }
      const double w = waited();
      if (w > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(w > 0.1 ? 500 : 20));      // a long pass (chunks: seconds): stop burning the core
      if (w > c->wait_timeout_s) {                           // a device pass that never finishes must not hang the caller for ever
        char b[320];
        snprintf(b, sizeof b, "the device pass on lane %d did not finish within %.0f s (record seq %u, expected %u): S4P_WAIT_TIMEOUT_S", li, c->wait_timeout_s, unsigned(*seq), unsigned(want));
        c->err = b;
        return S4P_ERR_HIP;
      }
    }
  }
  if (*seq != want) {
    HIPCHK(c, hipEventSynchronize(c->done[li]));
    if (*seq != want) S4P_FAIL(c, S4P_ERR_STATE, "k_verify finished without writing its result record");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return S4P_OK;
}
size_t lane_bytes(uint64_t mp, uint64_t mq);

hipError_t alloc_lane_buffers(uint64_t mp_, uint64_t mq_, s4p_ctx::LaneBufs& L, const char** what);

// Replaces the per-base buffers of ONE idle lane by a set for mp pairs / mq quads.  The lanes together may take 60 % of
// the device memory.  The new set is allocated before the old one is released whenever the device has room for both;
// otherwise the old set goes first, is re-created if the new one cannot be had, and only if that fails too is the context
// left refusing further passes (`broken`).
int32_t grow_lane(s4p_ctx* c, int li, uint64_t mp, uint64_t mq) {
  s4p_ctx::Lane& L = c->lane[li];
  auto refuse = [&]() {
    char b[220];
    snprintf(b, sizeof b, "a base needs device buffers of max_pairs=%llu max_quads=%llu (%.1f GB for this lane, %d lanes): refused",
             (unsigned long long)mp, (unsigned long long)mq, double(lane_bytes(mp, mq)) / 1e9, c->n_lanes);
    c->err = b;
    return S4P_ERR_CAPACITY;
  };
  if (mp > 0x7FFFFFFFull || mq > 0x7FFFFFFFull) return refuse();
  size_t free_b = 0, total_b = 0;
  HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
  double all = double(lane_bytes(mp, mq));
  for (int j = 0; j < c->n_lanes; ++j) if (j != li) all += double(lane_bytes(c->lane[j].cap_pairs, c->lane[j].cap_quads));
  if (all > 0.6 * double(total_b)) return refuse();
  const uint64_t old_p = L.cap_pairs, old_q = L.cap_quads;
  const bool both_fit = double(lane_bytes(mp, mq)) < 0.9 * double(free_b);
  if (!both_fit && double(lane_bytes(mp, mq)) - double(lane_bytes(old_p, old_q)) > 0.9 * double(free_b)) return refuse();
  s4p_ctx::LaneBufs fresh;
  const char* what = nullptr;
  if (!both_fit) L.free_all();
  hipError_t e = alloc_lane_buffers(mp, mq, fresh, &what);
  if (e != hipSuccess) {
    fresh.free_all();
    c->err = std::string(what ? what : "hipMalloc") + ": " + hipGetErrorString(e);
    if (!both_fit) {
      s4p_ctx::LaneBufs back;
      if (alloc_lane_buffers(old_p, old_q, back, &what) == hipSuccess) static_cast<s4p_ctx::LaneBufs&>(L) = back;
      else { back.free_all(); c->broken = true; c->err += " (context unusable)"; }
    }
    return e == hipErrorOutOfMemory ? S4P_ERR_OOM : S4P_ERR_HIP;
  }
  L.free_all();
  static_cast<s4p_ctx::LaneBufs&>(L) = fresh;              // (DevBuf holds plain pointers: the set moves as a whole)
  c->lane_growths++;
  return S4P_OK;
}

// HIP-event times of the last launch whose first lane was `li`, into the profile (once).
void harvest_events(s4p_ctx* c, int li) {
  if (!c->ev_pending[li]) return;
  c->ev_pending[li] = false;
  float ms = 0.f;
  hipError_t e = hipEventElapsedTime(&ms, c->ev[li][0], c->ev[li][1]);
  if (e == hipErrorNotReady) { (void)hipEventSynchronize(c->ev[li][1]); e = hipEventElapsedTime(&ms, c->ev[li][0], c->ev[li][1]); }
  if (e == hipSuccess) { c->prof.verify_launches++; c->prof.verify_ms_total += ms; }
  if (c->ev_fused[li]) {
    if (hipEventElapsedTime(&ms, c->ev[li][2], c->ev[li][3]) == hipSuccess) { c->prof.pairs_ms_total += ms; c->prof.pairs_launches += 2; }
    if (hipEventElapsedTime(&ms, c->ev[li][3], c->ev[li][4]) == hipSuccess) { c->prof.quads_ms_total += ms; c->prof.quads_launches += 1; }
  }
}

void account_profile(s4p_ctx* c, const DevCounters& d, bool fused) {
  if (c->prof_events) {                                    // per base
    c->prof.verify_candidates += d.C; c->prof.verify_quads += std::min<uint64_t>(d.K, c->lane[c->cur].cap_quads); c->prof.verify_queries += uint64_t(d.C) * c->n_q;
  }
  // (the events of the launch are read LATER -- harvest_events: when the lane is launched again, or by s4p_profile_get: the host
  // has seen the result record, the stream may not have reached the closing event yet, and waiting for it here would stall
  // the launch thread by a few microseconds per launch inside the timed region)
  if (c->prof_events && c->lane_group_n[c->cur] > 0) { c->ev_pending[c->cur] = true; c->ev_fused[c->cur] = fused && c->prof_stages; }
  c->prof.verify_pruned += d.pruned;
  if (c->prof_points) { c->prof.verify_point_tests += d.point_tests; c->prof.verify_l0_pass += d.l0_pass; c->prof.verify_l1_pass += d.l1_pass; c->prof.verify_l2_pass += d.l2_pass; }
}

// winner record of a device pass -> s4p_base_result (counts are filled in by the caller)
void fill_winner(const DevCounters& d, bool have, const BaseFrame& bf, s4p_base_result* r) {
  r->best_count = have ? d.best_count : 0u; r->has_best = have ? 1 : 0;
  r->best_rank = have ? d.best_tag : ~0ull;
  for (int i = 0; i < 4; ++i) r->best_quad[i] = have ? d.best_quad[i] : 0;
  for (int i = 0; i < 16; ++i) r->best_transform[i] = have ? d.best_T[i] : ((i % 5 == 0) ? 1.f : 0.f);
  for (int k = 0; k < 3; ++k) { r->best_centroid2[k] = have ? d.best_c2[k] : 0.f; r->centroid1[k] = bf.c1[k]; }
}

// Candidates of the pass that just finished on lane c->cur whose Euler-angle bound (match4pcsBase.cc:457-472) the device
// could not decide (euler_verdict == 2): they were scored but kept out of the selection.  Here the reference's own
// expression (libm, host) decides each one; those that pass are folded into the pass's record with the rule of the
// selection (greater count, then smaller tag), those that fail leave the candidate count and checksum and get the
// "gate failed" mark in the per-quad count array.  d: the pass's record, updated in place.
int32_t settle_borderline(s4p_ctx* c, DevCounters& d, const BaseFrame& bf) {
  c->border_failed.clear();
  if (!d.n_border) return S4P_OK;
  if (d.n_border > kBorderCap) S4P_FAIL(c, S4P_ERR_STATE, "more candidates with an undecided Euler-angle gate than the device hands over (kBorderCap)");
  s4p_ctx::Lane& L = c->lane[c->cur];
  HIPCHK(c, hipEventSynchronize(c->done[c->cur]));          // (the wait returned on the record's launch number: the launch itself must be over before its buffers are read back)
  std::vector<uint32_t> pos(d.n_border);
  HIPCHK(c, hipMemcpy(pos.data(), L.border.p, size_t(d.n_border) * 4, hipMemcpyDeviceToHost));
  bool have = d.has_best != 0u;                            // the device selected among the decided candidates it scored (with a bound in force: the sweep's survivors)
  for (const uint32_t i : pos) {
    uint32_t kraw = 0, count = 0; int4 qd; unsigned long long tag = 0;
    HIPCHK(c, hipMemcpy(&kraw, L.cand_idx.p + i, 4, hipMemcpyDeviceToHost));
    const uint32_t k = kraw & ~kBorderFlag;
    HIPCHK(c, hipMemcpy(&qd, L.quads.p + k, sizeof qd, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&tag, L.tags.p + k, 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&count, L.counts.p + k, 4, hipMemcpyDeviceToHost));
    const int ids[3] = {qd.x, qd.y, qd.z};
    float q[3][3], T[12], c2[3];
    for (int a = 0; a < 3; ++a) { q[a][0] = c->hqx[size_t(ids[a])]; q[a][1] = c->hqy[size_t(ids[a])]; q[a][2] = c->hqz[size_t(ids[a])]; }
    c->border_settled++;
    if (rigid_verdict<true>(bf, q, T, c2) != 0) {          // host: the exact expression
      if (!have || count > d.best_count || (count == d.best_count && tag < d.best_tag)) {
        have = true; d.best_count = count; d.best_tag = tag; d.has_best = 1u;
        d.best_quad[0] = qd.x; d.best_quad[1] = qd.y; d.best_quad[2] = qd.z; d.best_quad[3] = qd.w;
        for (int t = 0; t < 12; ++t) d.best_T[t] = T[t];
        d.best_T[12] = 0.f; d.best_T[13] = 0.f; d.best_T[14] = 0.f; d.best_T[15] = 1.f;
        for (int t = 0; t < 3; ++t) d.best_c2[t] = c2[t];
      }
    } else {
      c->border_rejected++;
      d.C -= 1u; d.cand_sum -= quad_mix(qd.x, qd.y, qd.z, qd.w);
      const uint32_t failed = kGateFailed;
      HIPCHK(c, hipMemcpy(L.counts.p + k, &failed, 4, hipMemcpyHostToDevice));
      c->border_failed.push_back(k);
    }
  }
  d.n_border = 0;
  return S4P_OK;
}

int32_t launch_verify(s4p_ctx* c, const BaseFrame& bf);
int32_t enqueue_result(s4p_ctx* c, const BaseFrame& bf);
int32_t wait_lane(s4p_ctx* c, int li);

// Records of the device pass that just finished on lane c->cur (d: its counters, after settle_borderline), in reference
// order: verified candidates (count + row-major 4x4 of the centred frame) -> sink and/or `kept`; with want_quads also every
// quad with its count (-1 = gate failed) -> `kept` (s4p_last_candidates).  tag_base: added to the pass's tags (slices of a
// caller's list number their quads from 0).
int32_t capture_pass(s4p_ctx* c, const DevCounters& d, bool want_quads, bool to_kept) {
  s4p_ctx::Lane& L = c->lane[c->cur];
  const uint64_t K = std::min<uint64_t>(d.K, L.cap_quads);
  const uint32_t Cdev = c->hctr[c->cur].p->C;                // as the device counted them (incl. candidates the host rejected afterwards)
  if (K == 0) return S4P_OK;
  HIPCHK(c, hipEventSynchronize(c->done[c->cur]));          // (as in settle_borderline)
  std::vector<unsigned long long> t(K); std::vector<uint32_t> cn(K);
  HIPCHK(c, hipMemcpy(t.data(), L.tags.p, K * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(cn.data(), L.counts.p, K * 4, hipMemcpyDeviceToHost));
  if (Cdev) {
    std::vector<uint32_t> idx(Cdev); std::vector<float4> T(size_t(Cdev) * kCandStride);
    HIPCHK(c, hipMemcpy(idx.data(), L.cand_idx.p, size_t(Cdev) * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(T.data(), L.cand_T.p, size_t(Cdev) * 16 * kCandStride, hipMemcpyDeviceToHost));
    std::vector<uint32_t> order; order.reserve(Cdev);
    for (uint32_t a = 0; a < Cdev; ++a) { idx[a] &= ~kBorderFlag; if (cn[idx[a]] != kGateFailed) order.push_back(a); }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t[idx[a]] < t[idx[b]]; });
    const size_t n = order.size();
    std::vector<uint32_t> oc(n); std::vector<float> oT(n * 16);
    for (size_t i = 0; i < n; ++i) {
      const uint32_t a = order[i];
      oc[i] = cn[idx[a]];
      float* o = oT.data() + 16 * i;
      std::memcpy(o, &T[size_t(a) * kCandStride], 48);
      o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
    }
    if (c->sink && n) c->sink(c->sink_user, oc.data(), oT.data(), int64_t(n));
    if (to_kept) { c->kept.counts.insert(c->kept.counts.end(), oc.begin(), oc.end()); c->kept.T16.insert(c->kept.T16.end(), oT.begin(), oT.end()); }
  }
  if (want_quads && to_kept) {
    std::vector<int4> q(K);
    HIPCHK(c, hipMemcpy(q.data(), L.quads.p, K * 16, hipMemcpyDeviceToHost));
    std::vector<uint32_t> order(K);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t[a] < t[b]; });
    for (uint64_t i = 0; i < K; ++i) {
      const int4 v = q[order[i]];
      c->kept.quads.push_back(v.x); c->kept.quads.push_back(v.y); c->kept.quads.push_back(v.z); c->kept.quads.push_back(v.w);
      c->kept.qcounts.push_back(cn[order[i]] == kGateFailed ? -1 : int32_t(cn[order[i]]));
    }
  }
  return S4P_OK;
}
void launch_quads_kernel(s4p_ctx* c, const QuadParams& Q);
void launch_gate_kernel(s4p_ctx* c, const GateParams& G);
GateParams gate_params(s4p_ctx* c, const BaseFrame& bf);

// A base whose congruent quads do not fit the quad buffers (first.overflow == 4, first.K = how many there are):
// FindCongruentQuadrilaterals + TryCongruentSet (super4pcs.cc:132-174, match4pcsBase.hpp:363-497) over RANGES of the
// set-2 pairs.  The pair sets, their preparation and the cell hash of the base are still on the lane; every range is
// enumerated into the (reused) quad buffers, gated and scored by the same kernels, and its best is folded with the rule
// of the single pass -- greatest count, then smallest tag = earliest in the reference's std::set order -- so the winner
// is the first maximum of the whole base (match4pcsBase.hpp:467-484) whatever the chunking.  A range that still
// overflows is halved.  The per-range quad counts and checksums must add up to those of the counting pass.
int32_t run_chunked(s4p_ctx* c, const DevCounters& first, s4p_base_result* r) {
  const int li = c->cur;
  s4p_ctx::Lane& L = c->lane[li];
  QuadParams Q = c->slot_q[li];
  const BaseFrame bf = c->slot_bf[li];
  const uint32_t m1 = first.m1, m2 = first.m2;            // exact: the pair lists fitted
  const uint64_t Ktot = first.K;
  const uint64_t target = std::max<uint64_t>(L.cap_quads * 6 / 10, 1);          // entries are in append order, i.e. shuffled: ranges are even
  const uint64_t nch = (Ktot + target - 1) / target;
  // (a share of the set, s4p_set_quad_slice, is a predicate on the pairs' order keys inside k_quads: the ranges cover the
  // whole list on every GPU, and Ktot already counts this GPU's share only)
  // Two ways to cut the base.  Default: ranges of the SECOND pair set's entries (even, cheap: a pass enumerates only its
  // range).  With a listener for per-candidate records: ranges of the FIRST pair set's order key -- the primary key of the
  // reference's candidate order, so that pass after pass the records come out in that order; every pass then walks the
  // whole second set and keeps the quads whose set-1 pair lies in the range (k_quads: k1_lo, k1_hi).
  const bool ordered = c->capturing();
  const uint64_t span = ordered ? 2ull * uint64_t(c->n_q) * uint64_t(std::max<uint32_t>(L.sv_nseq1, 1u)) : uint64_t(m2);
  const uint64_t step = std::max<uint64_t>(1, (span + nch - 1) / nch);
  std::vector<std::pair<uint64_t, uint64_t>> todo;
  for (uint64_t a = 0; a < span; a += step) todo.emplace_back(a, std::min<uint64_t>(a + step, span));
  std::reverse(todo.begin(), todo.end());
  c->kept.clear();                                         // (records of an earlier base must never answer for this one)
  uint64_t Ksum = 0, Csum = 0, qsum = 0, csum = 0;
  DevCounters best{}; bool have = false;
  c->chunk_bases++; c->chunk_quads += Ktot;
  if (!c->hmm[li].p) HIPCHK(c, c->hmm[li].alloc(2));
  while (!todo.empty()) {
    const std::pair<uint64_t, uint64_t> rg = todo.back(); todo.pop_back();
    c->hmm[li].p[0] = m1; c->hmm[li].p[1] = m2;            // k_verify cleared the live counters: the pair counts come back
    HIPCHK(c, hipMemcpyAsync(&L.ctr.p->m1, c->hmm[li].p, 8, hipMemcpyHostToDevice, L.stream));
    if (ordered) { Q.r0 = 0u; Q.r1 = 0xFFFFFFFFu; Q.k1_all = 0; Q.k1_lo = uint32_t(rg.first); Q.k1_hi = uint32_t(std::min<uint64_t>(rg.second, 0xFFFFFFFFull)); }
    else { Q.r0 = uint32_t(rg.first); Q.r1 = uint32_t(rg.second); }
    launch_quads_kernel(c, Q);
    if (!Q.do_gate) launch_gate_kernel(c, gate_params(c, bf));
    HIPCHK(c, hipGetLastError());
    c->chunk_pass = true;
    const int32_t vrc = launch_verify(c, bf);
    c->chunk_pass = false;
    if (vrc) return vrc;
    if (int32_t rc = enqueue_result(c, bf)) return rc;
    if (int32_t rc = wait_lane(c, li)) return rc;
    DevCounters d = *c->hctr[li].p;
    if (d.overflow & 4u) {
      if (rg.second - rg.first < 2u) S4P_FAIL(c, S4P_ERR_CAPACITY, "one pair has more congruent quads than max_quads: raise s4p_limits.max_quads");
      const uint64_t mid = rg.first + (rg.second - rg.first) / 2u;
      todo.emplace_back(mid, rg.second); todo.emplace_back(rg.first, mid);
      c->chunk_splits++;
      continue;
    }
    if (d.overflow) S4P_FAIL(c, S4P_ERR_STATE, "chunk pass: unexpected overflow bits");
    if (int32_t rc = settle_borderline(c, d, bf)) return rc;
    if (ordered) if (int32_t rc = capture_pass(c, d, c->keep_records, c->keep_records)) return rc;
    c->chunk_passes++;
    account_profile(c, d, false);
    Ksum += d.K; Csum += d.C; qsum += d.quad_sum; csum += d.cand_sum;
    if (d.has_best && (!have || d.best_count > best.best_count || (d.best_count == best.best_count && d.best_tag < best.best_tag))) { best = d; have = true; }
  }
  if (Ksum != Ktot || qsum != first.quad_sum) {
    char b[200];
    snprintf(b, sizeof b, "chunked enumeration disagrees with the counting pass (quads %llu vs %llu, checksum %016llx vs %016llx)",
             (unsigned long long)Ksum, (unsigned long long)Ktot, (unsigned long long)qsum, (unsigned long long)first.quad_sum);
    c->err = b;
    return S4P_ERR_STATE;
  }
  std::memset(r, 0, sizeof(*r));
  r->n_pairs1 = m1; r->n_pairs2 = m2; r->n_quads = Ksum; r->n_verified = Csum;
  r->quad_checksum = qsum; r->cand_checksum = csum;
  fill_winner(best, have, bf, r);
  if (ordered && c->keep_records) c->kept.valid = true;
  c->last_K = 0; c->last_chunked = true;
  c->need_quads = Ktot; c->need_pairs = 0;
  return S4P_OK;
}

int32_t prepare_base(s4p_ctx* c, int32_t slot, const int32_t* base_ids, float inv1, float inv2);
int32_t flush_lanes(s4p_ctx* c, const int* lanes, int n, hipStream_t st);
int32_t reset_counters(s4p_ctx* c);

// The base of lane c->cur once more, after the lane's buffers have grown: same staged sequences (the staging slot is
// recycled only after the wait has returned), same base points, same parameters -- alone, on the lane's own stream.
int32_t relaunch_base(s4p_ctx* c) {
  s4p_ctx::Lane& L = c->lane[c->cur];
  float kx[12], kc[12];
  std::memcpy(kx, c->base_xyz, sizeof kx); std::memcpy(kc, c->base_rgb, sizeof kc);
  std::memcpy(c->base_xyz, L.sv_bx, sizeof kx); std::memcpy(c->base_rgb, L.sv_brgb, sizeof kc);
  int32_t rc = prepare_base(c, L.sv_slot, L.sv_ids, L.sv_inv1, L.sv_inv2);
  std::memcpy(c->base_xyz, kx, sizeof kx); std::memcpy(c->base_rgb, kc, sizeof kc);
  if (rc) return rc;
  const int lane = c->cur;
  if ((rc = flush_lanes(c, &lane, 1, L.stream)) != S4P_OK) return rc;
  return wait_lane(c, lane);
}

// wait for slot c->cur and turn its counters into an s4p_base_result
int32_t finish_result(s4p_ctx* c, s4p_base_result* r, bool fused) {
  { auto t0 = std::chrono::steady_clock::now();
    const int32_t wrc = wait_lane(c, c->cur);
    c->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (wrc) return wrc; }
  const int li = c->cur;
  for (int attempt = 0;; ++attempt) {
    DevCounters d = *c->hctr[li].p;                        // (a copy: relaunches and chunk passes reuse the pinned record)
    const BaseFrame& bf = c->slot_bf[li];
    if (attempt == 0) account_profile(c, d, fused);
    if (fused) {                                           // what the next launches size their k_prep / k_quads grids by: rises at once, decays slowly
      c->est_m1 = std::max(d.m1, c->est_m1 - c->est_m1 / 16u);
      c->est_m2 = std::max(d.m2, c->est_m2 - c->est_m2 / 16u);
    }
    if (fused && (d.overflow & 8u) && !(d.overflow & 3u)) {
      // the cell hash of this base was sized from the registration's earlier bases (set-1 preparation inside k_pairs2) and this
      // base has more pairs than that: its pair count is exact (the counter kept counting), est_m1 holds it now -- once more
      if (attempt >= 4) S4P_FAIL(c, S4P_ERR_STATE, "set-1 cell hash: the size estimate keeps failing");
      c->prep_redos++;
      if (int32_t rc = relaunch_base(c)) return rc;
      continue;
    }
    if (!d.overflow) {
      if (int32_t rc = settle_borderline(c, d, bf)) return rc;
      std::memset(r, 0, sizeof(*r));
      r->n_pairs1 = d.m1; r->n_pairs2 = d.m2; r->n_quads = d.K; r->n_verified = d.C;
      r->quad_checksum = d.quad_sum; r->cand_checksum = d.cand_sum;
      fill_winner(d, d.has_best != 0u, bf, r);
      c->last_K = d.K; c->last_chunked = false;
      if (c->sink && fused) if (int32_t rc = capture_pass(c, d, false, false)) return rc;
      return S4P_OK;
    }
    s4p_ctx::Lane& L = c->lane[li];
    const bool pairs_over = (d.overflow & 3u) != 0u;
    if (fused && !pairs_over && c->chunking) {              // only the quads did not fit: chunk the base, then widen the lane
      if (int32_t rc = run_chunked(c, d, r)) return rc;
      if (c->auto_grow && L.cap_quads < c->quad_grow_cap) {
        const uint64_t want = std::min<uint64_t>(c->quad_grow_cap, std::max<uint64_t>(2 * L.cap_quads, d.K + d.K / 4));
        std::string keep = c->err;
        if (grow_lane(c, li, L.cap_pairs, want) != S4P_OK) c->err = keep;      // no room: later bases are chunked as well
      }
      return S4P_OK;
    }
    if (!fused || !c->auto_grow || attempt >= 4) return check_overflow(c, d);
    // a pair list (or, without chunking, the quad list) did not fit: the counters kept counting, so they say what the
    // base needs (the quads only once the pairs fit)
    uint64_t mp = L.cap_pairs, mq = L.cap_quads;
    const uint64_t need_p = std::max<uint64_t>(d.m1, d.m2);
    if (need_p > mp) mp = std::max<uint64_t>(2 * mp, need_p + need_p / 4);
    if (!pairs_over && d.K > mq) mq = std::max<uint64_t>(2 * mq, d.K + d.K / 4);
    if (int32_t rc = grow_lane(c, li, mp, mq)) return rc;
    if (int32_t rc = relaunch_base(c)) return rc;
  }
}

// First half of a base's device pass, on lane c->cur: the parameter records of its four kernels (and the upload of its staged
// sequences) into the lane's launch record.  Nothing is enqueued: the launch belongs to the base's GROUP (flush_lanes).
int32_t prepare_base(s4p_ctx* c, int32_t slot, const int32_t* base_ids, float inv1, float inv2) {
  s4p_ctx::Lane& L = c->lane[c->cur];
  s4p_ctx::LaunchRec& R = L.rec;
  const float eps = 2.0f * c->opt.delta;
  R.bf = make_base_frame(c, base_ids);
  if (int32_t rc = quad_params(c, inv1, inv2, eps, R.p1, R.q)) return rc;
  R.pp = PairParams2{};
  fill_pair_params(c, slot, 0, eps, 0, 1, R.pp.set[0].pair);
  fill_pair_params(c, slot, 1, eps, 2, 3, R.pp.set[1].pair);
  const s4p_ctx::StageSlot& st = c->stage[slot];
  R.up_src = st.blob.p; R.up_dst = L.seqbuf.p;
  R.up_bytes = (size_t(st.off[1]) + s4p_ctx::StageSlot::set_words(st.n_seq[1], st.n_leaf[1])) * sizeof(uint32_t);      // both sets, one copy
  // Set-1 preparation inside k_pairs2 when the registration's recent bases say how large the cell hash has to be (est_m1 with
  // 50 % head-room, 4 slots per pair, >= 64 Ki, a power of two within the allocation); otherwise -- first bases, or after a base
  // that needed more -- the k_prep launch sizes the table from the final count on the device.
  R.fused_prep = c->fuse_prep && c->est_m1 != 0u;
  if (R.fused_prep) {
    const uint64_t want = std::max<uint64_t>(65536u, 4ull * (uint64_t(c->est_m1) * 3u / 2u));      // (a floor of 64 Ki slots: a redo costs far more than a sparser table)
    const uint32_t fm = std::min<uint32_t>(next_pow2(want) - 1u, L.ht_mask);
    R.p1.ht.fixed_mask = fm; R.q.ht.fixed_mask = fm;
    PairParams& S0 = R.pp.set[0].pair;
    S0.prep_on = 1; S0.prep = R.p1; S0.prep_overflow = &L.ctr.p->overflow;
  }
  R.q.do_gate = 1; R.q.gate = gate_params(c, R.bf);
  R.q.slice_num = c->slice_num; R.q.slice_den = c->slice_den;
  c->slot_q[c->cur] = R.q;                                // (the chunk loop relaunches it range by range if the quads do not fit)
  c->slot_bf[c->cur] = R.bf;
  return S4P_OK;
}

// The device pass of the prepared bases on lanes[0..n) as ONE chain on stream st: per base one upload, then four launches
// that cover all of them (k_pairs2: both pair sets of every base; k_prep; k_quads: enumeration + rigid transform + rms gate;
// k_verify: LCP of every candidate + winners + result records, counters cleared for the lanes' next bases), then every
// lane's completion event.
int32_t flush_lanes_enqueue(s4p_ctx* c, const int* lanes, int n, hipStream_t st);
// ... and if any step of it fails, every lane of the launch remembers the failure: the wait for such a base returns it instead of
// polling a result record that still holds the lane's PREVIOUS launch number
int32_t flush_lanes(s4p_ctx* c, const int* lanes, int n, hipStream_t st) {
  const int32_t rc = flush_lanes_enqueue(c, lanes, n, st);
  if (rc != S4P_OK) for (int b = 0; b < n; ++b) { c->lane[lanes[b]].pending = false; c->lane[lanes[b]].failed_rc = rc; c->lane[lanes[b]].failed_msg = c->err; }
  return rc;
}
int32_t flush_lanes_enqueue(s4p_ctx* c, const int* lanes, int n, hipStream_t st) {
  using lclk = std::chrono::steady_clock;
  lclk::time_point tp[8];
  auto lap = [&](int k) { if (c->trace_launch) tp[k] = lclk::now(); };
  lap(0);
  PairGroup PG{}; PrepGroup G1{}; QuadGroup GQ{};
  for (int b = 0; b < n; ++b) {
    s4p_ctx::Lane& L = c->lane[lanes[b]];
    if (L.dirty) {                                         // a stage-level call left the live counters non-zero
      hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(1), 0, st, L.ctr.p);
      L.dirty = false;
    }
    PG.base[b] = L.rec.pp; G1.base[b] = L.rec.p1; GQ.base[b] = L.rec.q;
    L.pending = false;
  }
  harvest_events(c, lanes[0]);                              // (the previous launch's, before its events are recorded again)
  if (c->prof_stages) HIPCHK(c, hipEventRecord(c->ev[lanes[0]][2], st));
  lap(1);
  for (int b = 0; b < n; ++b) {
    const s4p_ctx::LaunchRec& R = c->lane[lanes[b]].rec;
    if (R.up_bytes) HIPCHK(c, hipMemcpyAsync(R.up_dst, R.up_src, R.up_bytes, hipMemcpyHostToDevice, st));
  }
  lap(2);
  if (int32_t rc = launch_pairs_kernel(c, PG, n, 2, st)) return rc;
  if (c->prof_stages) HIPCHK(c, hipEventRecord(c->ev[lanes[0]][3], st));
  lap(3);
  { // the bases whose set-1 preparation does not ride inside k_pairs2 (no size estimate yet) get the k_prep launch
    PrepGroup GP{}; int np = 0;
    for (int b = 0; b < n; ++b) if (!c->lane[lanes[b]].rec.fused_prep) GP.base[np++] = G1.base[b];
    if (np) launch_prep_group(GP, np, st, c->est_m1); }
  lap(4);
  launch_quads_group(c, GQ, n, st, c->est_m2);
  HIPCHK(c, hipGetLastError());
  if (c->prof_stages) HIPCHK(c, hipEventRecord(c->ev[lanes[0]][4], st));
  lap(5);
  if (int32_t rc = launch_verify_group(c, lanes, n, st)) return rc;
  lap(6);
  for (int b = 0; b < n; ++b) { HIPCHK(c, hipEventRecord(c->done[lanes[b]], st)); c->lane_group_n[lanes[b]] = (b == 0) ? n : 0; }
  lap(7);
  if (c->trace_launch) { for (int k = 0; k < 7; ++k) c->lt[k] += std::chrono::duration<double>(tp[k + 1] - tp[k]).count(); c->lt_n += uint64_t(n); c->lt_groups++; }
  return S4P_OK;
}

// The pending bases of group g (consecutive lanes, oldest first) as one launch on the stream of the group's first lane.
int32_t flush_group(s4p_ctx* c, int g) {
  int lanes[kGroupMax]; int n = 0;
  const int lo = g * c->group, hi = std::min(c->n_lanes, lo + c->group);
  for (int li = lo; li < hi; ++li) if (c->lane[li].pending && n < kGroupMax) lanes[n++] = li;
  if (n == 0) return S4P_OK;
  return flush_lanes(c, lanes, n, c->lane[lo].stream);
}

int32_t fetch_result(s4p_ctx* c, const BaseFrame& bf, s4p_base_result* r) {
  if (int32_t rc = enqueue_result(c, bf)) return rc;
  return finish_result(c, r, false);
}

#define S4P_NEED_IDLE(c) do { if ((c)->q_head != (c)->q_tail) S4P_FAIL(c, S4P_ERR_STATE, "asynchronous bases outstanding: call s4p_try_base_wait first"); (c)->cur = 0; } while (0)

int32_t reset_counters(s4p_ctx* c) {
  hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(1), 0, c->lane[c->cur].stream, c->lane[c->cur].ctr.p);
  HIPCHK(c, hipGetLastError());
  c->lane[c->cur].dirty = false;
  return S4P_OK;
}

// Per-lane buffers sized by the limits (pairs: 11 arrays, quads: 5, the cell hash): at creation and when the limits grow.
hipError_t alloc_lane_buffers(uint64_t mp_, uint64_t mq_, s4p_ctx::LaneBufs& L, const char** what) {
  const size_t mp = mp_, mq = mq_;
  const uint32_t hts = next_pow2(2 * mp);
  hipError_t e = hipSuccess;
#define A(buf, cnt) if ((e = (buf).alloc(cnt)) != hipSuccess) { *what = "hipMalloc " #buf; return e; }
  A(L.ab1, mp); A(L.ab2, mp); A(L.okey1, mp); A(L.okey2, mp); A(L.cell1, mp);
  A(L.bucket1, mp); A(L.next1, mp); A(L.ew1, mp);
  A(L.quads, mq); A(L.tags, mq); A(L.counts, mq); A(L.cand_idx, mq); A(L.cand_T, mq * kCandStride); A(L.surv_T, mq * kCandStride);
  A(L.ht_keys, hts); A(L.ht_heads, hts); L.ht_mask = hts - 1;
#undef A
  *what = "hipMemset";
  if ((e = hipMemset(L.ht_keys.p, 0, size_t(hts) * 8)) != hipSuccess) return e;
  if ((e = hipMemset(L.ht_heads.p, 0, size_t(hts) * 8)) != hipSuccess) return e;
  L.epoch = 0; L.cap_pairs = mp_; L.cap_quads = mq_;
  return hipSuccess;
}
size_t lane_bytes(uint64_t mp, uint64_t mq) {            // what alloc_lane_buffers takes per lane
  return size_t(mp) * (8 + 8 + 4 * 5 + 16) + size_t(mq) * (16 + 8 + 4 + 4 + 2 * 16 * kCandStride) + size_t(next_pow2(2 * mp)) * 16;
}

}  // namespace

// ============================================================================
extern "C" {

uint64_t s4p_quad_mix(int32_t a, int32_t b, int32_t c, int32_t d) { return quad_mix(a, b, c, d); }

const char* s4p_last_error(const s4p_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t s4p_create(const s4p_options* opt, const s4p_limits* lim, int32_t device, s4p_ctx** out) {
  if (!opt || !out) { g_create_error = "null argument"; return S4P_ERR_BAD_ARG; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device visible: the MI355X path has no CPU fallback";
    return S4P_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { g_create_error = "bad device index"; return S4P_ERR_BAD_ARG; }
  if (!(opt->delta > 0.f)) { g_create_error = "delta must be > 0"; return S4P_ERR_BAD_ARG; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return S4P_ERR_HIP; }
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return S4P_ERR_NO_DEVICE;
  }
  s4p_ctx* c = new s4p_ctx();
  c->device = device; c->opt = *opt;
  if (const char* ln = getenv("S4P_LANES")) { const int v = atoi(ln); if (v >= 1 && v <= s4p_ctx::kMaxLanes) c->n_lanes = v; }
  if (const char* ab = getenv("S4P_ABLATE")) {       // profiling aid: drops parts of k_verify, so counts are WRONG
    c->ablate = atoi(ab);
    if (c->ablate) fprintf(stderr, "super4pcs_amd: S4P_ABLATE=%d is set: k_verify skips work, every result of this context is invalid\n", c->ablate);
  }
  if (const char* vb = getenv("S4P_VERIFY_BLOCKS")) { const int v = atoi(vb); if (v >= 64 && v <= kVerifyMaxBlocks) { c->verify_blocks = uint32_t(v); c->verify_blocks_fixed = true; c->verify_blocks_env = true; } }   // tuning knob
  if (const char* fp = getenv("S4P_FUSE_PREP")) c->fuse_prep = atoi(fp) != 0;
  if (const char* sp = getenv("S4P_SWEEP_PASS")) c->sweep_pass_env = atoi(sp) != 0 ? 1 : 0;
  if (const char* sb = getenv("S4P_VERIFY_BLOCKS_SURV")) { const int v = atoi(sb); if (v >= 16 && v <= kVerifyMaxBlocks) c->verify_blocks_surv = uint32_t(v); }
  if (const char* ps = getenv("S4P_PAIR_SPLIT")) { const int v = atoi(ps); if (v == 1 || v == 2 || v == 4) c->pair_split = uint32_t(v); }
  if (const char* gr = getenv("S4P_GROUP")) { const int v = atoi(gr); if (v >= 1 && v <= kGroupMax) c->group = v; }
  c->trace_launch = getenv("S4P_TRACE_LAUNCH") != nullptr;
  c->debug = getenv("S4P_DEBUG") != nullptr;
  if (const char* wt = getenv("S4P_WAIT_TIMEOUT_S")) { const double v = atof(wt); if (v > 0.0) c->wait_timeout_s = v; }
  if (const char* at = getenv("S4P_ANGLE_TOL")) { const float v = float(atof(at)); if (v > 1e-6f) c->angle_tol = v; }
  if (const char* qc = getenv("S4P_QUAD_GROW_CAP")) { const long long v = atoll(qc); if (v > 0 && v <= 0x7FFFFFFFll) c->quad_grow_cap = uint64_t(v); }
  snprintf(c->devname, sizeof c->devname, "%s (%s)", prop.name, prop.gcnArchName);
  // defaults: 1 Mi pairs per set, 4 Mi quads per base -- 0.45 GB per lane, a context in ~20 ms (4 Mi / 16 Mi took 0.3-0.6 s to
  // allocate and clear); Perform_N_steps grows them when a base needs more (s4p_grow_limits)
  c->max_pairs = (lim && lim->max_pairs) ? lim->max_pairs : (1ull << 20);
  c->max_quads = (lim && lim->max_quads) ? lim->max_quads : (4ull << 20);
  c->max_grid_cells = (lim && lim->max_grid_cells) ? lim->max_grid_cells : (1ull << 27);
  if (c->max_pairs > 0x7FFFFFFFull || c->max_quads > 0x7FFFFFFFull) { g_create_error = "limits exceed 2^31 entries"; delete c; return S4P_ERR_BAD_ARG; }
  for (int k = 0; k < 12; ++k) c->base_rgb[k] = -1.f;
  if (opt->max_angle > 0.f) {
    bool monotone = false;
    c->cos_min = angle_threshold(double(opt->max_angle) * M_PI / 180.0, &monotone);     // pairCreationFunctor.h:205
    c->angle_pairs = true;
    if (!monotone) { g_create_error = "libm acosf is not monotone around the max_angle threshold: the segment-angle pair filter cannot be reproduced"; delete c; return S4P_ERR_UNSUPPORTED; }
  }
  auto fail = [&](hipError_t e, const char* what) { g_create_error = std::string(what) + ": " + hipGetErrorString(e); s4p_destroy(c); return e == hipErrorOutOfMemory ? S4P_ERR_OOM : S4P_ERR_HIP; };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
  { // hardware queues the runtime was given against the group streams this context will use (see the stream creation below)
    const char* hq = getenv("GPU_MAX_HW_QUEUES");
    const int queues = (hq && atoi(hq) > 0) ? atoi(hq) : 4, streams = (c->n_lanes + c->group - 1) / c->group;
    int plo = 0, phi = 0;
    c->two_prio = queues < streams && hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess && plo != phi;
    if (const char* sp = getenv("S4P_STREAM_PRIO")) c->two_prio = atoi(sp) != 0 && plo != phi;
    c->prio_hi = phi; c->prio_mid = (plo + phi) / 2;
    if (c->debug) fprintf(stderr, "[s4p] %d group stream(s) on %d hardware queue(s) per priority level: %s\n", streams, queues, c->two_prio ? "two priority levels" : "one priority level");
  }
#define A(buf, cnt) if ((e = (buf).alloc(cnt)) != hipSuccess) return fail(e, "hipMalloc " #buf)
  for (int li = 0; li < c->n_lanes; ++li) {
    s4p_ctx::Lane& L = c->lane[li];
    { // The groups' streams share the runtime's hardware queues (GPU_MAX_HW_QUEUES, default 4: read by the runtime when it
      // initialises -- the application's to set, never this library's), and streams that share a queue serialise: 7 group
      // streams on 4 queues run at 198 M candidates/s against 231 M on 8.  The runtime keeps a pool of queues PER PRIORITY
      // LEVEL, so when the environment leaves fewer queues than the context has group streams, the groups alternate between
      // the normal and the high level and find a queue each: 223 M with nothing exported (profiles/r06_lab/summary_prio.txt;
      // with 8 queues the one-level form stays: 231 vs 221).  S4P_STREAM_PRIO=0 / 1 forces one / two levels.
      e = c->two_prio ? hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, ((li / c->group) & 1) ? c->prio_hi : c->prio_mid)
                      : hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
      if (e != hipSuccess) return fail(e, "hipStreamCreate"); }
    A(L.ctr, 2); A(L.slots, kVerifyMaxBlocks); A(L.border, kBorderCap);
    if ((e = hipMemset(L.ctr.p, 0, 2 * sizeof(DevCounters))) != hipSuccess) return fail(e, "hipMemset");
    const char* what = nullptr;
    if ((e = alloc_lane_buffers(c->max_pairs, c->max_quads, L, &what)) != hipSuccess) return fail(e, what);
    // the first clear of the lane's counters (best_tag = ~0) right here: it also makes the runtime create the stream's hardware
    // queue NOW -- lazily that costs ~0.1 ms on the first launch of every stream, inside the first bases of a registration
    // (measured: a 20-base burst after 5 warm-up bases ran at 100 us per base with 14 lanes, 74 us once every lane had been used)
    hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(1), 0, L.stream, L.ctr.p);
    L.dirty = false;
  }
  for (int li = 0; li < c->n_lanes; ++li) if ((e = hipStreamSynchronize(c->lane[li].stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
  A(c->group_done, s4p_ctx::kMaxLanes);
  if ((e = hipMemset(c->group_done.p, 0, s4p_ctx::kMaxLanes * sizeof(uint32_t))) != hipSuccess) return fail(e, "hipMemset");
#undef A
  for (int sl = 0; sl < s4p_ctx::kMaxLanes; ++sl) {
    if ((e = c->hctr[sl].alloc(1)) != hipSuccess) return fail(e, "hipHostMalloc");
    if ((e = hipEventCreateWithFlags(&c->done[sl], hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
  }
  {  // allow the verify kernels their dynamic LDS (coarse bitmap + quantised queries + survivor queues)
    const int max_lds = kVerifyLdsOnePerCu + int(sizeof(VerifyShared));
    const void* fns[] = {(const void*)k_verify<false, false, false>, (const void*)k_verify<false, true, false>, (const void*)k_verify<true, false, false>, (const void*)k_verify<true, true, false>,
                         (const void*)k_verify<false, false, true>, (const void*)k_verify<true, false, true>, (const void*)k_verify<false, true, true>, (const void*)k_verify<true, true, true>,
                         (const void*)k_verify_T<false, false>, (const void*)k_verify_T<false, true>, (const void*)k_verify_T<true, false>, (const void*)k_verify_T<true, true>};
    for (const void* fn : fns)
      if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return fail(e, "hipFuncSetAttribute");
    if ((e = hipFuncSetAttribute((const void*)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess) return fail(e, "hipFuncSetAttribute");
  }
  for (auto& row : c->ev) for (auto& ev : row) if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "hipEventCreate");
  if ((e = hipStreamCreateWithFlags(&c->sel_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = c->sel_draws.alloc(size_t(kSelectDraws) * kSelectBatch)) != hipSuccess || (e = c->sel_rec.alloc(kSelectBatch)) != hipSuccess) return fail(e, "hipMalloc selection buffers");
  if ((e = c->sel_hdraws.alloc(size_t(kSelectDraws) * kSelectBatch)) != hipSuccess || (e = c->sel_hrec.alloc(kSelectBatch)) != hipSuccess) return fail(e, "hipHostMalloc selection buffers");
  *out = c;
  return S4P_OK;
}

// Raises the pair / quad capacities of EVERY lane to at least the given numbers (the counts of the base that overflowed
// last, if 0) with 25 % head-room, at least doubling what overflowed.  Nothing may be in flight.  Refused
// (S4P_ERR_CAPACITY, nothing changed) when the lanes would then take more than 60 % of the device memory.  The fused
// path does not need this call: a lane whose base overflows grows on its own (finish_result); it serves the stage-level
// entry points and callers that want to size the buffers up front.
int32_t s4p_grow_limits(s4p_ctx* c, uint64_t min_pairs, uint64_t min_quads) {
  if (!c) return S4P_ERR_BAD_ARG;
  if (c->q_head != c->q_tail) S4P_FAIL(c, S4P_ERR_STATE, "s4p_grow_limits: bases in flight");
  HIPCHK(c, hipSetDevice(c->device));
  if (!min_pairs) min_pairs = c->need_pairs;
  if (!min_quads) min_quads = c->need_quads;
  uint64_t mp = c->max_pairs, mq = c->max_quads;
  if (min_pairs > mp) mp = std::max<uint64_t>(2 * mp, min_pairs + min_pairs / 4);
  if (min_quads > mq) mq = std::max<uint64_t>(2 * mq, min_quads + min_quads / 4);
  size_t free_b = 0, total_b = 0;
  HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
  double all = 0;
  for (int li = 0; li < c->n_lanes; ++li) all += double(lane_bytes(std::max(mp, c->lane[li].cap_pairs), std::max(mq, c->lane[li].cap_quads)));
  if (mp > 0x7FFFFFFFull || mq > 0x7FFFFFFFull || all > 0.6 * double(total_b)) {
    char b[200];
    snprintf(b, sizeof b, "device buffers would have to grow to max_pairs=%llu max_quads=%llu (%.1f GB in %d lanes): refused",
             (unsigned long long)mp, (unsigned long long)mq, all / 1e9, c->n_lanes);
    c->err = b;
    return S4P_ERR_CAPACITY;
  }
  for (int li = 0; li < c->n_lanes; ++li) {
    s4p_ctx::Lane& L = c->lane[li];
    HIPCHK(c, hipStreamSynchronize(L.stream));
    if (L.cap_pairs >= mp && L.cap_quads >= mq) continue;
    if (int32_t rc = grow_lane(c, li, std::max(mp, L.cap_pairs), std::max(mq, L.cap_quads))) return rc;
  }
  c->max_pairs = mp; c->max_quads = mq;
  c->need_pairs = c->need_quads = 0;
  return S4P_OK;
}

// Chunked processing of bases whose congruent quads exceed the quad buffers (on by default).  grow_cap_quads = the
// largest quad capacity s4p_grow_limits may reach (0 keeps the current value); enable = 0 restores the loud
// S4P_ERR_CAPACITY of the stage-level contract for such bases.
int32_t s4p_set_quad_chunking(s4p_ctx* c, int32_t enable, uint64_t grow_cap_quads) {
  if (!c) return S4P_ERR_BAD_ARG;
  if (grow_cap_quads > 0x7FFFFFFFull) S4P_FAIL(c, S4P_ERR_BAD_ARG, "quad capacity beyond 2^31 entries");
  c->chunking = enable != 0;
  if (grow_cap_quads) c->quad_grow_cap = grow_cap_quads;
  return S4P_OK;
}
// out4 = {bases processed in chunks, chunk passes, range splits after an overflowing chunk, quads enumerated by chunked bases}
int32_t s4p_chunk_stats(const s4p_ctx* c, uint64_t* out4) {
  if (!c || !out4) return S4P_ERR_BAD_ARG;
  out4[0] = c->chunk_bases; out4[1] = c->chunk_passes; out4[2] = c->chunk_splits; out4[3] = c->chunk_quads;
  return S4P_OK;
}
// out2 = {candidates whose Euler-angle bound the host settled, how many of them it rejected} (max_angle >= 0)
int32_t s4p_border_stats(const s4p_ctx* c, uint64_t* out2) {
  if (!c || !out2) return S4P_ERR_BAD_ARG;
  out2[0] = c->border_settled; out2[1] = c->border_rejected;
  return S4P_OK;
}
// Candidates that cannot EXCEED `best_count` inliers may be abandoned by the fused pass from the next base on: they cannot
// become the registration's best (match4pcsBase.hpp:468), which is what the reference's Verify exits early for
// (match4pcsBase.cc:520,558-560).  Winner, best count above the hint, n_quads and n_verified are unaffected; the counts of
// abandoned candidates (s4p_last_candidates, s4p_last_verified) and a base's best_count AT OR BELOW the hint are lower
// bounds.  0 (the default) = every candidate is counted in full.
int32_t s4p_set_best_hint(s4p_ctx* c, uint32_t best_count) { if (!c) return S4P_ERR_BAD_ARG; c->best_hint = best_count; return S4P_OK; }
// SURVEY 8e level 2: this context takes its share of every base's second pair set (order key = part mod parts: the keys, unlike
// the list positions, are the same on every GPU), with its quads, candidates and their best; parts = 0 or 1 restores the whole
// set.  Pairs and the set-1 hash are still built in full.
int32_t s4p_set_quad_slice(s4p_ctx* c, uint32_t part, uint32_t parts) {
  if (!c || (parts && part >= parts)) return S4P_ERR_BAD_ARG;
  c->slice_num = parts > 1 ? part : 0u; c->slice_den = parts > 1 ? parts : 0u;
  return S4P_OK;
}
// Lanes growing their own buffers when a base overflows (on by default); s4p_lane_growths counts the regrowths.
int32_t s4p_set_auto_grow(s4p_ctx* c, int32_t enable) { if (!c) return S4P_ERR_BAD_ARG; c->auto_grow = enable != 0; return S4P_OK; }
int64_t s4p_lane_growths(const s4p_ctx* c) { return c ? int64_t(c->lane_growths) : 0; }

int32_t s4p_get_limits(const s4p_ctx* c, s4p_limits* out) {
  if (!c || !out) return S4P_ERR_BAD_ARG;
  out->max_pairs = c->max_pairs; out->max_quads = c->max_quads; out->max_grid_cells = c->max_grid_cells;
  for (int li = 0; li < c->n_lanes; ++li) {                // lanes grow on their own: report the largest buffers in force
    out->max_pairs = std::max<uint64_t>(out->max_pairs, c->lane[li].cap_pairs);
    out->max_quads = std::max<uint64_t>(out->max_quads, c->lane[li].cap_quads);
  }
  return S4P_OK;
}

void s4p_destroy(s4p_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto& L : c->lane) if (L.stream) (void)hipStreamSynchronize(L.stream);
  if (c->trace_launch && c->lt_n)
    fprintf(stderr, "{\"s4p_trace\": \"launch\", \"bases\": %llu, \"us_per_base\": {\"params\": %.2f, \"uploads\": %.2f, \"k_pairs\": %.2f, \"k_prep\": %.2f, \"k_quads\": %.2f, "
                    "\"k_verify\": %.2f, \"result\": %.2f, \"wait\": %.2f, \"octree\": %.2f}, \"group_launches\": %llu, \"prep_redos\": %llu}\n", (unsigned long long)c->lt_n, c->lt[0] / c->lt_n * 1e6, c->lt[1] / c->lt_n * 1e6,
            c->lt[2] / c->lt_n * 1e6, c->lt[3] / c->lt_n * 1e6, c->lt[4] / c->lt_n * 1e6, c->lt[5] / c->lt_n * 1e6, c->lt[6] / c->lt_n * 1e6,
            c->host_wait_s / c->lt_n * 1e6, c->host_octree_s / c->lt_n * 1e6, (unsigned long long)c->lt_groups, (unsigned long long)c->prep_redos);
  c->greach.free(); c->glist_hdr.free(); c->gnbr.free();
  c->gcoarse.free(); c->q4.free(); c->q4v.free(); c->qquant.free(); c->qsoa.free(); c->qtiles.free();
  c->qx.free(); c->qy.free(); c->qz.free(); c->ux.free(); c->uy.free(); c->uz.free();
  c->qnx.free(); c->qny.free(); c->qnz.free(); c->qcr.free(); c->qcg.free(); c->qcb.free();
  for (auto& L : c->lane) {
    L.free_all(); L.ctr.free(); L.slots.free(); L.border.free();
    L.seqbuf.free();
  }
  for (auto& h : c->hctr) h.free();
  for (auto& h : c->hmm) h.free();
  for (auto& st : c->stage) st.blob.free();
  c->group_done.free();
  c->tbuf.free(); c->tpin.free();
  if (c->sel_stream) { (void)hipStreamSynchronize(c->sel_stream); (void)hipStreamDestroy(c->sel_stream); }
  c->p4o.free(); c->sel_draws.free(); c->sel_rec.free(); c->sel_hdraws.free(); c->sel_hrec.free();
  for (auto& e : c->tev) if (e) (void)hipEventDestroy(e);
  for (auto& row : c->ev) for (auto& ev : row) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->done) if (ev) (void)hipEventDestroy(ev);
  for (auto& L : c->lane) if (L.stream) (void)hipStreamDestroy(L.stream);
  delete c;
}

int32_t s4p_device_name(const s4p_ctx* c, char* buf, int32_t buflen) {
  if (!c || !buf || buflen <= 0) return S4P_ERR_BAD_ARG;
  snprintf(buf, size_t(buflen), "%s", c->devname);
  return S4P_OK;
}

// Which k_verify instantiation the trial loops launch on the clouds that are set, and how (measurement provenance: the
// bench line records it next to the commit): e.g. "k_verify<false, true, true> lean sweep, queries in LDS; 256 x 768 threads, 77.1 KB LDS".
int32_t s4p_verify_kernel_info(const s4p_ctx* c, char* buf, int32_t buflen) {
  if (!c || !buf || buflen <= 0) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) { snprintf(buf, size_t(buflen), "no clouds set"); return S4P_OK; }
  const char* loop = c->lean ? (c->lean_lds ? "k_verify<false, true, true> lean sweep (coarse-only, 16-bit queue entries), float queries in LDS"
                                            : "k_verify<false, false, true> lean sweep (coarse-only, 16-bit queue entries), queries from global memory")
                             : (c->qlds ? "k_verify<false, true, false> fused/staged sweep, quantised queries in LDS" : "k_verify<false, false, false> fused/staged sweep, float queries from global memory");
  const char* full = c->qlds ? "k_verify<false, true, false>" : "k_verify<false, false, false>";
  snprintf(buf, size_t(buflen), "with an early-exit bound: %s; full counts: %s; %u x %d threads, %.1f KB LDS (lean) / %.1f KB (fused); %d lanes in groups of %d bases per launch",
           loop, full, c->verify_blocks, c->verify_threads,
           double(c->lean ? c->lean_lds_bytes() : 0) / 1024.0, double(c->verify_lds_bytes()) / 1024.0, c->n_lanes, c->group);
  return S4P_OK;
}

namespace {
// exclusive scan of v[0..n) in place on stream st, *total = the sum (k_scan_* in s4p_kernels.hip.hpp); tmp: >= n / kScanTile + 1 words
void launch_scan(uint32_t* v, uint32_t n, uint32_t* total, uint32_t* tmp, hipStream_t st) {
  if (n <= 4u * kScanTile) { hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, v, n, total); return; }
  const uint32_t tiles = (n + kScanTile - 1u) / kScanTile;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(1024), 0, st, v, n, tmp);
  hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, tmp, tiles, total);
  hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(1024), 0, st, v, n, tmp);
}
}  // namespace

int32_t s4p_set_clouds(s4p_ctx* c, const float* px, const float* py, const float* pz, int64_t n_p,
                       const float* qx, const float* qy, const float* qz,
                       const float* qnx, const float* qny, const float* qnz,
                       const float* qr, const float* qg, const float* qb, int64_t n_q) {
  if (!c) return S4P_ERR_BAD_ARG;
  if (!px || !py || !pz || !qx || !qy || !qz || n_p <= 0 || n_q <= 0) S4P_FAIL(c, S4P_ERR_BAD_ARG, "s4p_set_clouds: null or empty cloud");
  if (n_q > 46340) S4P_FAIL(c, S4P_ERR_UNSUPPORTED, "sampled Q larger than 46340 points: 32-bit pair order keys would overflow");
  if (n_p > 0x7FFFFFF0ll) S4P_FAIL(c, S4P_ERR_UNSUPPORTED, "sampled P too large");
  if (c->q_head != c->q_tail) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds: asynchronous bases outstanding (their kernels read the buffers this call replaces): call s4p_try_base_wait first");
  HIPCHK(c, hipSetDevice(c->device));
  const auto sc_t0 = std::chrono::steady_clock::now();
  auto sc_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
  c->clouds_set = false;
  c->n_p = uint32_t(n_p); c->n_q = uint32_t(n_q);
  c->hpx.assign(px, px + n_p); c->hpy.assign(py, py + n_p); c->hpz.assign(pz, pz + n_p);
  c->hqx.assign(qx, qx + n_q); c->hqy.assign(qy, qy + n_q); c->hqz.assign(qz, qz + n_q);
  c->frame.build(c->hqx, c->hqy, c->hqz, c->hux, c->huy, c->huz);
  c->tree.reset(c->n_q);
  float cell_factor = LcpGridHost::kMinCellFactor;
  if (const char* cf = getenv("S4P_CELL_FACTOR")) cell_factor = float(atof(cf));      // tuning knob: LCP cell edge / delta
  if (!c->hgrid.plan(c->hpx, c->hpy, c->hpz, c->opt.delta, c->max_grid_cells, kCoarseMaxWords, cell_factor)) S4P_FAIL(c, S4P_ERR_STATE, "LCP grid planning failed");
  c->set_clouds_s[0] = sc_since(sc_t0);
  const auto sc_t1 = std::chrono::steady_clock::now();
  {  // device build of the LCP structure (counting formulation, see k_grid_* in s4p_kernels.hip.hpp)
    hipStream_t st = c->lane[0].stream;
    const uint64_t nc = c->hgrid.ncell();
    const uint32_t nwords = uint32_t((nc + 31) / 32);
    DevBuf<float> dpx, dpy, dpz; DevBuf<uint32_t> cell_count, word_pop, hdr_count, cell_id, cursor, totals, scan_tmp;
    hipError_t e = hipSuccess;
    int32_t rc = S4P_OK;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    do {
      if (!step(dpx.alloc(n_p)) || !step(dpy.alloc(n_p)) || !step(dpz.alloc(n_p)) || !step(cell_count.alloc(nc)) ||
          !step(word_pop.alloc(nwords)) || !step(totals.alloc(2)) || !step(scan_tmp.alloc(size_t(nc / kScanTile) + 8)) || !step(c->greach.alloc(nwords)) || !step(c->gcoarse.alloc(c->hgrid.coarse_words))) break;
      step(hipMemcpyAsync(dpx.p, c->hpx.data(), n_p * 4, hipMemcpyHostToDevice, st));
      step(hipMemcpyAsync(dpy.p, c->hpy.data(), n_p * 4, hipMemcpyHostToDevice, st));
      step(hipMemcpyAsync(dpz.p, c->hpz.data(), n_p * 4, hipMemcpyHostToDevice, st));
      step(hipMemsetAsync(cell_count.p, 0, nc * 4, st));
      step(hipMemsetAsync(c->gcoarse.p, 0, size_t(c->hgrid.coarse_words) * 4, st));
      if (e != hipSuccess) break;
      if (!step(c->p4o.alloc(size_t(n_p)))) break;
      hipLaunchKernelGGL(k_pack_points, dim3(1024), dim3(256), 0, st, dpx.p, dpy.p, dpz.p, uint32_t(n_p), c->p4o.p);
      GridBuildParams G{};
      G.px = dpx.p; G.py = dpy.p; G.pz = dpz.p; G.n_p = uint32_t(n_p);
      G.ox = c->hgrid.ox; G.oy = c->hgrid.oy; G.oz = c->hgrid.oz; G.h = c->hgrid.h; G.inv_h = c->hgrid.inv_h;
      G.nx = c->hgrid.nx; G.ny = c->hgrid.ny; G.nz = c->hgrid.nz; G.reach2 = c->hgrid.reach * c->hgrid.reach;
      G.cell_count = cell_count.p; G.reach = c->greach.p; G.n_words = nwords;
      G.coarse = c->gcoarse.p; G.cshift = c->hgrid.cshift; G.cnx = c->hgrid.cnx; G.cny = c->hgrid.cny;
      hipLaunchKernelGGL(k_grid_count, dim3(2048), dim3(256), 0, st, G);
      hipLaunchKernelGGL(k_grid_words, dim3(1024), dim3(256), 0, st, G, word_pop.p);
      launch_scan(word_pop.p, nwords, totals.p, scan_tmp.p, st);
      uint32_t n_reach = 0;
      step(hipMemcpyAsync(&n_reach, totals.p, 4, hipMemcpyDeviceToHost, st));
      if (!step(hipStreamSynchronize(st))) break;
      if (n_reach == 0) { c->err = "LCP grid: no reachable cell"; rc = S4P_ERR_STATE; break; }
      if (!step(hdr_count.alloc(n_reach)) || !step(cell_id.alloc(n_reach)) || !step(cursor.alloc(n_reach)) || !step(c->glist_hdr.alloc(size_t(n_reach) * 2u))) break;
      G.list_hdr = c->glist_hdr.p; G.cell_id = cell_id.p; G.cursor = cursor.p;
      hipLaunchKernelGGL(k_grid_headers, dim3(1024), dim3(256), 0, st, G, word_pop.p, hdr_count.p);
      // first line of every list = exclusive scan of the per-cell LINE counts (8 points per 128-byte line)
      DevBuf<uint32_t> starts;
      if (!step(starts.alloc(n_reach))) break;
      step(hipMemcpyAsync(starts.p, hdr_count.p, size_t(n_reach) * 4, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(k_lines_of, dim3(1024), dim3(256), 0, st, starts.p, n_reach);
      launch_scan(starts.p, n_reach, totals.p + 1, scan_tmp.p, st);
      uint32_t n_lines = 0;
      step(hipMemcpyAsync(&n_lines, totals.p + 1, 4, hipMemcpyDeviceToHost, st));
      if (!step(hipStreamSynchronize(st))) { starts.free(); break; }
      if (uint64_t(n_lines) * 8u >= (1ull << 31)) { c->err = "LCP grid: point lists beyond 2^31 records"; rc = S4P_ERR_CAPACITY; starts.free(); break; }
      if (!step(c->gnbr.alloc(size_t(n_lines) * 8u))) { starts.free(); break; }
      G.nbr = c->gnbr.p;
      { // k_verify block size: see kVerifyThreadsCached (s4p_kernels.hip.hpp)
        const char* vt = getenv("S4P_VERIFY_THREADS");
        const int v = vt ? atoi(vt) : 0;
        c->verify_threads = (v >= 256 && v <= kVerifyMaxThreads && v % 64 == 0) ? v : kVerifyThreadsCached;
        // k_verify workgroups: one per CU while the point lines stay in the Infinity Cache -- with the early exit the kernel is
        // short and the other lanes' pair / quad kernels need CU slots next to it (measured 256 / 384 / 512 / 768 workgroups:
        // 115.4 / 113.5 / 112.2 / 105.3 M candidates/s, profiles/r03_lanes_blocks_sweep.log); two per CU when the lines
        // stream from HBM, where more waves in flight carry the bandwidth
        if (!c->verify_blocks_fixed) c->verify_blocks = size_t(n_lines) * 128u > (size_t(192) << 20) ? 512u : 256u; }
      hipLaunchKernelGGL(k_lines_clear, dim3(2048), dim3(256), 0, st, c->gnbr.p, uint64_t(n_lines));
      hipLaunchKernelGGL(k_grid_hdr_pack, dim3(1024), dim3(256), 0, st, G, starts.p, hdr_count.p, n_reach);
      hipLaunchKernelGGL(k_grid_fill, dim3(2048), dim3(256), 0, st, G);
      MaskParams M{};
      M.list_hdr = c->glist_hdr.p; M.nbr = c->gnbr.p; M.cell_id = cell_id.p; M.n_reach = n_reach;
      M.ox = c->hgrid.ox; M.oy = c->hgrid.oy; M.oz = c->hgrid.oz; M.h = c->hgrid.h; M.nx = c->hgrid.nx; M.ny = c->hgrid.ny;
      M.reach2 = G.reach2;
      hipLaunchKernelGGL(k_build_masks, dim3(std::min<uint32_t>((n_reach + 3u) / 4u, 65536u)), dim3(256), 0, st, M);   // a wave per cell
      step(hipGetLastError());
      step(hipStreamSynchronize(st));
      starts.free();
    } while (0);
    dpx.free(); dpy.free(); dpz.free(); cell_count.free(); word_pop.free(); hdr_count.free(); cell_id.free(); cursor.free(); totals.free(); scan_tmp.free();
    if (rc != S4P_OK) return rc;
    HIPCHK(c, e);
  }
  c->set_clouds_s[1] = sc_since(sc_t1);
  const auto sc_t2 = std::chrono::steady_clock::now();
  {
    std::vector<float4> q4((size_t)n_q);
    for (int64_t i = 0; i < n_q; ++i) q4[size_t(i)] = make_float4(qx[i], qy[i], qz[i], 0.f);
    HIPCHK(c, c->q4.alloc(size_t(n_q)));
    HIPCHK(c, hipMemcpy(c->q4.p, q4.data(), size_t(n_q) * sizeof(float4), hipMemcpyHostToDevice));
    // Second copy for the LCP sweep, in Morton order of the unit-cube coordinates: Verify only counts inliers, so
    // the order of the queries is free, and spatially sorted queries keep the 64 lanes of a wave (and the 64
    // survivors of a phase-2 batch) in neighbouring grid cells -> shared LDS words and shared cache lines.
    std::vector<uint32_t> ord((size_t)n_q);
    std::vector<uint32_t> key((size_t)n_q);
    auto spread = [](uint32_t v) { v &= 0x3FFu; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu;
                                   v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u; return v; };
    for (int64_t i = 0; i < n_q; ++i) {
      ord[size_t(i)] = uint32_t(i);
      const uint32_t a = uint32_t(std::min(std::max(c->hux[size_t(i)], 0.f), 0.999f) * 1024.f);
      const uint32_t b = uint32_t(std::min(std::max(c->huy[size_t(i)], 0.f), 0.999f) * 1024.f);
      const uint32_t d = uint32_t(std::min(std::max(c->huz[size_t(i)], 0.f), 0.999f) * 1024.f);
      key[size_t(i)] = spread(a) | (spread(b) << 1) | (spread(d) << 2);
    }
    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });
    // (padded to a multiple of a sweep step with far-away points: the lean sweep of k_verify reads whole steps)
    const size_t n_pad_q = size_t((n_q + int64_t(kSweepStep) - 1) & ~(int64_t(kSweepStep) - 1));
    std::vector<float4> qv(n_pad_q, make_float4(kLeanPad, kLeanPad, kLeanPad, 0.f));
    for (int64_t i = 0; i < n_q; ++i) qv[size_t(i)] = q4[ord[size_t(i)]];
    HIPCHK(c, c->q4v.alloc(n_pad_q));
    HIPCHK(c, hipMemcpy(c->q4v.p, qv.data(), n_pad_q * sizeof(float4), hipMemcpyHostToDevice));
    // 16-bit quantisation of the same points over their bounding box, for the sweep's LDS copy (s4p_kernels.hip.hpp,
    // "LCP scoring").  Used when the sample fits the LDS budget and half a quantisation step stays below 0.004 cell
    // (the structure's slack is 0.01 cell); otherwise the sweep reads the float points from global memory.
    float lo[3] = {qv[0].x, qv[0].y, qv[0].z}, hi[3] = {qv[0].x, qv[0].y, qv[0].z};
    for (const float4& p : qv) {
      lo[0] = std::min(lo[0], p.x); hi[0] = std::max(hi[0], p.x);
      lo[1] = std::min(lo[1], p.y); hi[1] = std::max(hi[1], p.y);
      lo[2] = std::min(lo[2], p.z); hi[2] = std::max(hi[2], p.z);
    }
    bool fine_enough = true;
    for (int k = 0; k < 3; ++k) {
      c->qq.lo[k] = lo[k];
      c->qq.step[k] = (hi[k] > lo[k]) ? (hi[k] - lo[k]) / 65535.0f : 1.0f;
      if (!(0.5f * c->qq.step[k] * c->hgrid.inv_h < 0.004f)) fine_enough = false;
    }
    const size_t lds_room = size_t(c->verify_blocks <= 256u ? kVerifyLdsOnePerCu : kVerifyLdsBudget);      // (verify_blocks was settled by the structure build above)
    c->qlds = fine_enough && n_q <= int64_t(kLdsQueries) &&
              c->gcoarse.n * 4 + size_t((n_q + int64_t(kSweepStep) - 1) & ~(int64_t(kSweepStep) - 1)) * 8 + size_t(c->verify_threads / 64) * kQueueWordsPerWave * 4 <= lds_room;
    std::vector<uint2> packed((size_t)n_q);
    for (int64_t i = 0; i < n_q; ++i) {
      uint32_t u[3];
      const float v[3] = {qv[size_t(i)].x, qv[size_t(i)].y, qv[size_t(i)].z};
      for (int k = 0; k < 3; ++k) {
        const double t = std::floor((double(v[k]) - double(lo[k])) / double(c->qq.step[k]) + 0.5);
        u[k] = uint32_t(std::min(65535.0, std::max(0.0, t)));
      }
      packed[size_t(i)] = make_uint2(u[0] | (u[1] << 16), u[2]);
    }
    HIPCHK(c, c->qquant.alloc(size_t(n_q)));
    HIPCHK(c, hipMemcpy(c->qquant.p, packed.data(), size_t(n_q) * sizeof(uint2), hipMemcpyHostToDevice));
    c->qq.packed = c->qquant.p;
    // float copy for the lean sweep (early-exit mode): x | y | z, each padded to a multiple of a sweep step with far-away points
    c->lean = false; c->lean_lds = false; c->qsoa.free();
    {
      const size_t n_pad = n_pad_q;
      const size_t fixed = c->gcoarse.n * 4 + size_t(c->verify_threads / 64) * kLeanQueue * 2;
      if (n_q <= int64_t(kLeanMaxQueries) && fixed + n_pad * 12 <= lds_room) {
        std::vector<float> soa(3 * n_pad, kLeanPad);
        for (int64_t i = 0; i < n_q; ++i) { soa[size_t(i)] = qv[size_t(i)].x; soa[n_pad + size_t(i)] = qv[size_t(i)].y; soa[2 * n_pad + size_t(i)] = qv[size_t(i)].z; }
        HIPCHK(c, c->qsoa.alloc(3 * n_pad));
        HIPCHK(c, hipMemcpy(c->qsoa.p, soa.data(), 3 * n_pad * sizeof(float), hipMemcpyHostToDevice));
        c->lean = c->lean_lds = true;
      } else if (n_q <= 65535 && fixed <= size_t(kVerifyLdsBudget)) {
        c->lean = true;                                      // queries from global memory (samples that do not fit LDS: the 20 000-point sample)
      }
    }
    // k_sweep's tiles: the same points in the same (Morton) order, tile by tile x | y | z, the last tile padded with far-away points;
    // one tile while the sample fits kSweepTileMax, tiles of 2048 beyond (any sample size goes through LDS)
    c->qtiles.free(); c->tile_q = 0; c->n_tiles = 0;
    if (c->lean) {
      const uint32_t n_pad = uint32_t(n_pad_q);
      // as few tiles as fit kSweepTileMax queries each, all of the same size (a multiple of a sweep step): 5000 points are two tiles of 2560
      c->n_tiles = (n_pad + kSweepTileMax - 1u) / kSweepTileMax;
      c->tile_q = ((uint32_t(n_q) + c->n_tiles - 1u) / c->n_tiles + kSweepStep - 1u) & ~(kSweepStep - 1u);
      if (c->gcoarse.n * 4 + size_t(c->tile_q) * 12 + sizeof(SweepShared) <= size_t(kVerifyLdsBudget)) {
        std::vector<float> tl(size_t(c->n_tiles) * 3u * c->tile_q, kLeanPad);
        for (int64_t i = 0; i < n_q; ++i) {
          const size_t t = size_t(i) / c->tile_q, o = size_t(i) % c->tile_q, b0 = t * 3u * c->tile_q;
          tl[b0 + o] = qv[size_t(i)].x; tl[b0 + c->tile_q + o] = qv[size_t(i)].y; tl[b0 + 2u * c->tile_q + o] = qv[size_t(i)].z;
        }
        HIPCHK(c, c->qtiles.alloc(tl.size()));
        HIPCHK(c, hipMemcpy(c->qtiles.p, tl.data(), tl.size() * sizeof(float), hipMemcpyHostToDevice));
      }
    }
  }
  auto up = [&](DevBuf<float>& d, const float* src) -> hipError_t {
    hipError_t e = d.alloc(n_q); if (e != hipSuccess) return e;
    return hipMemcpy(d.p, src, n_q * 4, hipMemcpyHostToDevice);
  };
  HIPCHK(c, up(c->qx, qx)); HIPCHK(c, up(c->qy, qy)); HIPCHK(c, up(c->qz, qz));
  HIPCHK(c, up(c->ux, c->hux.data())); HIPCHK(c, up(c->uy, c->huy.data())); HIPCHK(c, up(c->uz, c->huz.data()));
  c->has_normals = (qnx && qny && qnz); c->has_rgb = (qr && qg && qb);
  if (c->has_normals) { HIPCHK(c, up(c->qnx, qnx)); HIPCHK(c, up(c->qny, qny)); HIPCHK(c, up(c->qnz, qnz)); }
  if (c->has_rgb) { HIPCHK(c, up(c->qcr, qr)); HIPCHK(c, up(c->qcg, qg)); HIPCHK(c, up(c->qcb, qb)); }
  for (int li = 0; li < c->n_lanes; ++li) HIPCHK(c, c->lane[li].seqbuf.alloc(2 * s4p_ctx::StageSlot::blob_words(n_q) + 16));      // both sets of a base
  for (auto& st : c->stage) HIPCHK(c, st.blob.alloc(2 * s4p_ctx::StageSlot::blob_words(n_q) + 16));
  c->est_m1 = c->est_m2 = 0u;                            // a new registration: no grid estimates yet
  c->clouds_set = true;
  c->set_clouds_s[2] = sc_since(sc_t2); c->set_clouds_s[3] = sc_since(sc_t0);
  return S4P_OK;
}

// Wall time of the last s4p_set_clouds, seconds: {host copies + unit frame + grid plan, device build of the LCP structure,
// Q-side uploads, total}.  Measurement aid.
int32_t s4p_set_clouds_timing(const s4p_ctx* c, double* out4) {
  if (!c || !out4) return S4P_ERR_BAD_ARG;
  for (int k = 0; k < 4; ++k) out4[k] = c->set_clouds_s[k];
  return S4P_OK;
}

int32_t s4p_set_base(s4p_ctx* c, const float* xyz, const float* nrm, const float* rgb) {
  if (!c || !xyz) return S4P_ERR_BAD_ARG;
  std::memcpy(c->base_xyz, xyz, sizeof c->base_xyz);
  if (nrm) std::memcpy(c->base_nrm, nrm, sizeof c->base_nrm); else std::memset(c->base_nrm, 0, sizeof c->base_nrm);
  if (rgb) std::memcpy(c->base_rgb, rgb, sizeof c->base_rgb); else for (int k = 0; k < 12; ++k) c->base_rgb[k] = -1.f;
  return S4P_OK;
}

int32_t s4p_extract_pairs(s4p_ctx* c, float pair_distance, float pair_normals_angle, float pair_distance_epsilon,
                          int32_t bp1, int32_t bp2, int32_t* out_pairs, int64_t cap, int64_t* n_out) {
  if (!c || !n_out) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds not called");
  if (bp1 < 0 || bp1 > 3 || bp2 < 0 || bp2 > 3) S4P_FAIL(c, S4P_ERR_BAD_ARG, "base_point index out of [0,3]");
  S4P_NEED_IDLE(c);
  HIPCHK(c, hipSetDevice(c->device));
  if (int32_t rc = reset_counters(c)) return rc;
  c->lane[c->cur].dirty = true;
  if (int32_t rc = launch_pairs(c, 0, pair_distance, pair_normals_angle, pair_distance_epsilon, bp1, bp2)) return rc;
  HIPCHK(c, hipMemcpyAsync(c->hctr[c->cur].p, c->lane[c->cur].ctr.p, sizeof(DevCounters), hipMemcpyDeviceToHost, c->lane[c->cur].stream));
  HIPCHK(c, hipStreamSynchronize(c->lane[c->cur].stream));
  if (int32_t rc = check_overflow(c, *c->hctr[c->cur].p)) return rc;
  const uint32_t m = c->hctr[c->cur].p->m1;
  *n_out = m;
  if (m == 0) return S4P_OK;
  if (!out_pairs || cap < int64_t(m)) S4P_FAIL(c, S4P_ERR_CAPACITY, "s4p_extract_pairs: output buffer too small");
  std::vector<int2> ab(m); std::vector<uint32_t> ok(m);
  HIPCHK(c, hipMemcpy(ab.data(), c->lane[c->cur].ab1.p, size_t(m) * sizeof(int2), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(ok.data(), c->lane[c->cur].okey1.p, size_t(m) * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> order(m);
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ok[a] < ok[b]; });
  for (uint32_t i = 0; i < m; ++i) { out_pairs[2 * i] = ab[order[i]].x; out_pairs[2 * i + 1] = ab[order[i]].y; }
  return S4P_OK;
}

int32_t s4p_find_congruent(s4p_ctx* c, float inv1, float inv2, float /*thr1*/, float thr2,
                           const int32_t* pairs1, int64_t m1, const int32_t* pairs2, int64_t m2,
                           int32_t* out_quads, int64_t cap, int64_t* n_out) {
  if (!c || !n_out) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds not called");
  *n_out = 0;
  if (m1 <= 0 || m2 <= 0) return S4P_OK;
  if (!pairs1 || !pairs2) S4P_FAIL(c, S4P_ERR_BAD_ARG, "null pair list");
  if (uint64_t(m1) > c->lane[0].cap_pairs || uint64_t(m2) > c->lane[0].cap_pairs) S4P_FAIL(c, S4P_ERR_CAPACITY, "pair list longer than max_pairs");
  for (int64_t i = 0; i < 2 * m1; ++i) if (pairs1[i] < 0 || uint32_t(pairs1[i]) >= c->n_q) S4P_FAIL(c, S4P_ERR_BAD_ARG, "pair index out of range");
  for (int64_t i = 0; i < 2 * m2; ++i) if (pairs2[i] < 0 || uint32_t(pairs2[i]) >= c->n_q) S4P_FAIL(c, S4P_ERR_BAD_ARG, "pair index out of range");
  S4P_NEED_IDLE(c);
  HIPCHK(c, hipSetDevice(c->device));
  if (int32_t rc = reset_counters(c)) return rc;
  std::vector<uint32_t> idx((size_t)std::max(m1, m2));
  std::iota(idx.begin(), idx.end(), 0u);
  HIPCHK(c, hipMemcpyAsync(c->lane[c->cur].ab1.p, pairs1, size_t(m1) * 8, hipMemcpyHostToDevice, c->lane[c->cur].stream));
  HIPCHK(c, hipMemcpyAsync(c->lane[c->cur].ab2.p, pairs2, size_t(m2) * 8, hipMemcpyHostToDevice, c->lane[c->cur].stream));
  HIPCHK(c, hipMemcpyAsync(c->lane[c->cur].okey1.p, idx.data(), size_t(m1) * 4, hipMemcpyHostToDevice, c->lane[c->cur].stream));
  HIPCHK(c, hipMemcpyAsync(c->lane[c->cur].okey2.p, idx.data(), size_t(m2) * 4, hipMemcpyHostToDevice, c->lane[c->cur].stream));
  const uint32_t mm[2] = {uint32_t(m1), uint32_t(m2)};
  HIPCHK(c, hipMemcpyAsync(&c->lane[c->cur].ctr.p->m1, mm, 8, hipMemcpyHostToDevice, c->lane[c->cur].stream));
  c->lane[c->cur].dirty = true;
  PrepParams P1; QuadParams Q;
  if (int32_t rc = quad_params(c, inv1, inv2, thr2, P1, Q)) return rc;
  launch_prep_kernel(c, P1);
  launch_quads_kernel(c, Q);
  HIPCHK(c, hipGetLastError());
  s4p_ctx::Lane& L = c->lane[c->cur];
  HIPCHK(c, hipMemcpyAsync(c->hctr[c->cur].p, L.ctr.p, sizeof(DevCounters), hipMemcpyDeviceToHost, L.stream));
  HIPCHK(c, hipStreamSynchronize(L.stream));
  if (c->hctr[c->cur].p->overflow == 4u && c->chunking) {
    // More congruent quads than the lane's quad buffers hold (the reference's std::vector<Quadrilateral> simply grows,
    // super4pcs.cc:166-174): the counter kept counting, so the size of the list is known; it is enumerated again in ranges of
    // the FIRST pair set (a quad's tag is (index in P_pairs) << 32 | index in Q_pairs: ranges in ascending order are chunks
    // of the std::set order), each chunk sorted and appended to the caller's buffer.
    const uint64_t Ktot = c->hctr[c->cur].p->K;
    *n_out = int64_t(Ktot);
    if (!out_quads || uint64_t(cap) < Ktot) S4P_FAIL(c, S4P_ERR_CAPACITY, "s4p_find_congruent: output buffer too small");
    const uint64_t target = std::max<uint64_t>(L.cap_quads * 6 / 10, 1), nch = (Ktot + target - 1) / target;
    const uint64_t step = std::max<uint64_t>(1, (uint64_t(m1) + nch - 1) / nch);
    std::vector<std::pair<uint64_t, uint64_t>> todo;
    for (uint64_t a = 0; a < uint64_t(m1); a += step) todo.emplace_back(a, std::min<uint64_t>(a + step, uint64_t(m1)));
    std::reverse(todo.begin(), todo.end());
    uint64_t at = 0;
    while (!todo.empty()) {
      const std::pair<uint64_t, uint64_t> rg = todo.back(); todo.pop_back();
      const unsigned long long zero = 0ull; const uint32_t z32 = 0u;
      HIPCHK(c, hipMemcpyAsync(&L.ctr.p->K, &zero, 8, hipMemcpyHostToDevice, L.stream));
      HIPCHK(c, hipMemcpyAsync(&L.ctr.p->overflow, &z32, 4, hipMemcpyHostToDevice, L.stream));
      Q.k1_all = 0; Q.k1_lo = uint32_t(rg.first); Q.k1_hi = uint32_t(rg.second);
      launch_quads_kernel(c, Q);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(c->hctr[c->cur].p, L.ctr.p, sizeof(DevCounters), hipMemcpyDeviceToHost, L.stream));
      HIPCHK(c, hipStreamSynchronize(L.stream));
      const DevCounters& d = *c->hctr[c->cur].p;
      if (d.overflow & 4u) {
        if (rg.second - rg.first < 2u) S4P_FAIL(c, S4P_ERR_CAPACITY, "one pair has more congruent quads than max_quads: raise s4p_limits.max_quads");
        const uint64_t mid = rg.first + (rg.second - rg.first) / 2u;
        todo.emplace_back(mid, rg.second); todo.emplace_back(rg.first, mid);
        continue;
      }
      const uint64_t Kc = d.K;
      if (at + Kc > Ktot) S4P_FAIL(c, S4P_ERR_STATE, "chunked quad enumeration disagrees with the counting pass");
      std::vector<int4> q(Kc); std::vector<unsigned long long> t(Kc);
      if (Kc) {
        HIPCHK(c, hipMemcpy(q.data(), L.quads.p, size_t(Kc) * 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(t.data(), L.tags.p, size_t(Kc) * 8, hipMemcpyDeviceToHost));
      }
      std::vector<uint32_t> order(Kc);
      std::iota(order.begin(), order.end(), 0u);
      std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t[a] < t[b]; });
      for (uint64_t i = 0; i < Kc; ++i) {
        const int4 v = q[order[i]];
        int32_t* o = out_quads + 4 * (at + i);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
      }
      at += Kc;
    }
    if (at != Ktot) S4P_FAIL(c, S4P_ERR_STATE, "chunked quad enumeration disagrees with the counting pass");
    return S4P_OK;
  }
  if (int32_t rc = check_overflow(c, *c->hctr[c->cur].p)) return rc;
  const uint32_t K = c->hctr[c->cur].p->K;
  *n_out = K;
  if (K == 0) return S4P_OK;
  if (!out_quads || cap < int64_t(K)) S4P_FAIL(c, S4P_ERR_CAPACITY, "s4p_find_congruent: output buffer too small");
  std::vector<int4> q(K); std::vector<unsigned long long> t(K);
  HIPCHK(c, hipMemcpy(q.data(), c->lane[c->cur].quads.p, size_t(K) * 16, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(t.data(), c->lane[c->cur].tags.p, size_t(K) * 8, hipMemcpyDeviceToHost));
  std::vector<uint32_t> order(K);
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t[a] < t[b]; });   // std::set<(id,i)> order
  for (uint32_t i = 0; i < K; ++i) {
    const int4 v = q[order[i]];
    out_quads[4 * i] = v.x; out_quads[4 * i + 1] = v.y; out_quads[4 * i + 2] = v.z; out_quads[4 * i + 3] = v.w;
  }
  return S4P_OK;
}

int32_t s4p_try_congruent_set(s4p_ctx* c, const int32_t* base_ids, const int32_t* quads, int64_t K,
                              int32_t* per_candidate, s4p_base_result* result) {
  if (!c || !base_ids || !result) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds not called");
  for (int i = 0; i < 4; ++i) if (base_ids[i] < 0 || uint32_t(base_ids[i]) >= c->n_p) S4P_FAIL(c, S4P_ERR_BAD_ARG, "base id out of range");
  if (K < 0) S4P_FAIL(c, S4P_ERR_BAD_ARG, "negative quad count");
  if (K > 0 && !quads) S4P_FAIL(c, S4P_ERR_BAD_ARG, "null quads");
  for (int64_t i = 0; i < 4 * K; ++i) if (quads[i] < 0 || uint32_t(quads[i]) >= c->n_q) S4P_FAIL(c, S4P_ERR_BAD_ARG, "quad index out of range");
  S4P_NEED_IDLE(c);
  HIPCHK(c, hipSetDevice(c->device));
  const BaseFrame bf = make_base_frame(c, base_ids);
  s4p_ctx::Lane& L = c->lane[c->cur];
  // The caller's list may be longer than the lane's quad buffers (the reference's std::vector has no such limit,
  // match4pcsBase.hpp:340-351): it is scored in slices, each slice's best folded with the rule of the single pass
  // (greatest count, then smallest position in the list), the per-candidate records kept on the host if somebody listens.
  const uint64_t cap = std::max<uint64_t>(L.cap_quads, 1);
  const bool sliced = uint64_t(K) > cap;
  if (sliced || c->capturing()) c->kept.clear();
  DevCounters best{}; bool have = false;
  uint64_t Csum = 0, csum = 0;
  for (uint64_t off = 0; off == 0 || off < uint64_t(K); off += cap) {
    const uint64_t n = std::min<uint64_t>(cap, uint64_t(K) - off);
    if (int32_t rc = reset_counters(c)) return rc;
    std::vector<unsigned long long> tg((size_t)n);
    std::iota(tg.begin(), tg.end(), (unsigned long long)off);
    if (n > 0) {
      HIPCHK(c, hipMemcpyAsync(L.quads.p, quads + 4 * off, size_t(n) * 16, hipMemcpyHostToDevice, L.stream));
      HIPCHK(c, hipMemcpyAsync(L.tags.p, tg.data(), size_t(n) * 8, hipMemcpyHostToDevice, L.stream));
    }
    const unsigned long long k64 = (unsigned long long)n;
    HIPCHK(c, hipMemcpyAsync(&L.ctr.p->K, &k64, 8, hipMemcpyHostToDevice, L.stream));
    launch_gate_kernel(c, gate_params(c, bf));
    if (int32_t rc = launch_verify(c, bf)) return rc;
    s4p_base_result part;
    if (int32_t rc = fetch_result(c, bf, &part)) return rc;        // (waits: tg may go out of scope)
    if (per_candidate && n > 0) {
      std::vector<uint32_t> cnt((size_t)n);
      HIPCHK(c, hipMemcpy(cnt.data(), L.counts.p, size_t(n) * 4, hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < n; ++i) per_candidate[off + i] = cnt[i] == kGateFailed ? -1 : int32_t(cnt[i]);
    }
    if (sliced || c->capturing()) {
      DevCounters d = *c->hctr[c->cur].p;
      d.C = uint32_t(part.n_verified);
      if (int32_t rc = capture_pass(c, d, true, true)) return rc;
    }
    Csum += part.n_verified; csum += part.cand_checksum;
    if (!sliced) { *result = part; if (c->capturing()) { c->kept.valid = true; c->last_chunked = true; } return S4P_OK; }
    if (part.has_best && (!have || part.best_count > best.best_count || (part.best_count == best.best_count && part.best_rank < best.best_tag))) {
      have = true; best.best_count = part.best_count; best.best_tag = part.best_rank;
      for (int i = 0; i < 4; ++i) best.best_quad[i] = part.best_quad[i];
      for (int i = 0; i < 16; ++i) best.best_T[i] = part.best_transform[i];
      for (int i = 0; i < 3; ++i) best.best_c2[i] = part.best_centroid2[i];
    }
  }
  std::memset(result, 0, sizeof(*result));
  result->n_quads = uint64_t(K); result->n_verified = Csum; result->cand_checksum = csum;
  fill_winner(best, have, bf, result);
  c->kept.valid = true; c->last_chunked = true; c->last_K = 0;
  return S4P_OK;
}

namespace {
int32_t verify_transforms_impl(s4p_ctx* c, const float* T, int64_t B, uint32_t* counts, uint64_t* stats4) {
  if (!c || (B > 0 && (!T || !counts))) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds not called");
  if (B <= 0) return S4P_OK;
  S4P_NEED_IDLE(c);
  HIPCHK(c, hipSetDevice(c->device));
  s4p_ctx::Lane& L = c->lane[c->cur];
  hipStream_t st = L.stream;
  DevBuf<float> dT; DevBuf<uint32_t> dC;
  HIPCHK(c, dT.alloc(size_t(B) * 16));
  hipError_t e = dC.alloc(size_t(B));
  if (e != hipSuccess) { dT.free(); HIPCHK(c, e); }
  int32_t rc = S4P_OK;
  do {
    if (stats4) { if ((rc = reset_counters(c)) != S4P_OK) break; L.dirty = true; }
    if ((e = hipMemcpyAsync(dT.p, T, size_t(B) * 64, hipMemcpyHostToDevice, st)) != hipSuccess) break;
    VerifyTParams V{};
    V.grid = c->dev_grid(); V.q4 = c->q4v.p; V.qq = c->qq; V.n_q = c->n_q; V.T = dT.p; V.B = uint32_t(B);
    V.counts = dC.p; V.ctr = L.ctr.p;
    const uint32_t wpb = uint32_t(c->verify_threads) / 64u;
    const uint32_t blocks = uint32_t(std::min<int64_t>((B + wpb - 1) / wpb, 512));
    const size_t lds = c->verify_lds_bytes();
    if (stats4) { if (c->qlds) hipLaunchKernelGGL((k_verify_T<true, true>), dim3(blocks), dim3(c->verify_threads), lds, st, V); else hipLaunchKernelGGL((k_verify_T<true, false>), dim3(blocks), dim3(c->verify_threads), lds, st, V); }
    else { if (c->qlds) hipLaunchKernelGGL((k_verify_T<false, true>), dim3(blocks), dim3(c->verify_threads), lds, st, V); else hipLaunchKernelGGL((k_verify_T<false, false>), dim3(blocks), dim3(c->verify_threads), lds, st, V); }
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipMemcpyAsync(counts, dC.p, size_t(B) * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    if (stats4 && (e = hipMemcpyAsync(c->hctr[c->cur].p, L.ctr.p, sizeof(DevCounters), hipMemcpyDeviceToHost, st)) != hipSuccess) break;
    e = hipStreamSynchronize(st);
    if (stats4 && e == hipSuccess) {
      const DevCounters& d = *c->hctr[c->cur].p;
      stats4[0] = d.point_tests; stats4[1] = d.l0_pass; stats4[2] = d.l1_pass; stats4[3] = d.l2_pass;
    }
  } while (0);
  dT.free(); dC.free();
  if (rc != S4P_OK) return rc;
  if (e != hipSuccess) { c->err = std::string("s4p_verify_transforms: ") + hipGetErrorString(e); rc = S4P_ERR_HIP; }
  return rc;
}
}  // namespace

int32_t s4p_verify_transforms(s4p_ctx* c, const float* T, int64_t B, uint32_t* counts) {
  return verify_transforms_impl(c, T, B, counts, nullptr);
}

int32_t s4p_verify_transforms_counted(s4p_ctx* c, const float* T, int64_t B, uint32_t* counts, uint64_t* stats4) {
  if (!stats4) return S4P_ERR_BAD_ARG;
  return verify_transforms_impl(c, T, B, counts, stats4);
}

#if defined(S4P_PROF)
// lab build only (-DS4P_PROF=1): the per-wave stamps of the last launches of kernel `which` (0 k_pairs2, 1 k_quads, 2 k_verify);
// the device copy is cleared after the read
int32_t s4p_debug_prof(int32_t which, uint64_t* out, int32_t n_words) {
  if (which < 0 || which > 2 || !out) return S4P_ERR_BAD_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return S4P_ERR_HIP;
  const size_t all = size_t(kProfWords) * kProfWaves, bytes = std::min<size_t>(size_t(n_words), all) * 8, off = size_t(which) * all * 8;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), bytes, off, hipMemcpyDeviceToHost) != hipSuccess) return S4P_ERR_HIP;
  static std::vector<unsigned long long> zeros(all, 0ull);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zeros.data(), all * 8, off, hipMemcpyHostToDevice) == hipSuccess ? S4P_OK : S4P_ERR_HIP;
}
#endif
int32_t s4p_stage_slots(const s4p_ctx*) { return s4p_ctx::kStageSlots; }
int32_t s4p_pipeline_depth(const s4p_ctx* c) { return c ? c->n_lanes : 0; }

int32_t s4p_stage_base(s4p_ctx* c, const float* base_xyz, const float* base_nrm, int32_t want_device_data, int32_t slot) {
  if (!c || !base_xyz) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds not called");
  if (want_device_data && (slot < 0 || slot >= s4p_ctx::kStageSlots)) S4P_FAIL(c, S4P_ERR_BAD_ARG, "bad staging slot");
  float zero[12] = {0};
  const float* nrm = base_nrm ? base_nrm : zero;
  const float eps = 2.0f * c->opt.delta;                                          // distance_factor * delta
  stage_pairs(c, slot, 0, seg_len(base_xyz, 0, 1), seg_len(nrm, 0, 1), eps, want_device_data != 0);
  stage_pairs(c, slot, 1, seg_len(base_xyz, 2, 3), seg_len(nrm, 2, 3), eps, want_device_data != 0);
  return S4P_OK;
}

int32_t s4p_try_base_staged_async(s4p_ctx* c, int32_t slot, const int32_t* base_ids, float inv1, float inv2) {
  if (!c || !base_ids) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds not called");
  if (slot < 0 || slot >= s4p_ctx::kStageSlots) S4P_FAIL(c, S4P_ERR_BAD_ARG, "bad staging slot");
  if (c->broken) S4P_FAIL(c, S4P_ERR_STATE, "context unusable: a buffer growth failed half-way");
  for (int i = 0; i < 4; ++i) if (base_ids[i] < 0 || uint32_t(base_ids[i]) >= c->n_p) S4P_FAIL(c, S4P_ERR_BAD_ARG, "base id out of range");
  if (c->q_tail - c->q_head >= uint32_t(c->n_lanes)) S4P_FAIL(c, S4P_ERR_STATE, "all lanes busy: call s4p_try_base_wait first");
  HIPCHK(c, hipSetDevice(c->device));
  c->cur = int(c->q_tail % uint32_t(c->n_lanes));
  s4p_ctx::Lane& L = c->lane[c->cur];
  L.sv_slot = slot; L.sv_inv1 = inv1; L.sv_inv2 = inv2; L.sv_gen = c->stage[slot].gen; L.sv_nseq1 = c->stage[slot].n_seq[0];
  for (int i = 0; i < 4; ++i) L.sv_ids[i] = base_ids[i];
  std::memcpy(L.sv_bx, c->base_xyz, sizeof L.sv_bx); std::memcpy(L.sv_brgb, c->base_rgb, sizeof L.sv_brgb);
  if (int32_t rc = prepare_base(c, slot, base_ids, inv1, inv2)) return rc;
  L.pending = true;
  c->q_tail++;
  // the group goes to the device with its last base (the last lane of the group, or of the context); a wait for one of its
  // bases launches it earlier with what it has (s4p_try_base_wait)
  const int g = c->cur / c->group;
  if (c->cur == std::min(c->n_lanes, (g + 1) * c->group) - 1) return flush_group(c, g);
  return S4P_OK;
}

int32_t s4p_try_base_async(s4p_ctx* c, const int32_t* base_ids, float inv1, float inv2) {
  if (!c || !base_ids) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set) S4P_FAIL(c, S4P_ERR_STATE, "s4p_set_clouds not called");
  if (c->q_tail - c->q_head >= uint32_t(c->n_lanes)) S4P_FAIL(c, S4P_ERR_STATE, "all lanes busy: call s4p_try_base_wait first");
  // self-staging: a private round-robin slot (pipeline depth + 1 slots are always safe to rotate through)
  const int slot = int(c->stage_rr++ % uint32_t(c->n_lanes + 1));
  if (int32_t rc = s4p_stage_base(c, c->base_xyz, c->base_nrm, 1, slot)) return rc;
  return s4p_try_base_staged_async(c, slot, base_ids, inv1, inv2);
}

int32_t s4p_try_base_wait(s4p_ctx* c, s4p_base_result* result) {
  if (!c || !result) return S4P_ERR_BAD_ARG;
  if (c->q_head == c->q_tail) S4P_FAIL(c, S4P_ERR_STATE, "no asynchronous base in flight");
  HIPCHK(c, hipSetDevice(c->device));
  c->cur = int(c->q_head % uint32_t(c->n_lanes));
  c->q_head++;
  if (c->lane[c->cur].pending) if (int32_t rc = flush_group(c, c->cur / c->group)) return rc;      // its group was still filling up
  return finish_result(c, result, true);
}

int32_t s4p_try_base(s4p_ctx* c, const int32_t* base_ids, float inv1, float inv2, s4p_base_result* result) {
  if (!c || !base_ids || !result) return S4P_ERR_BAD_ARG;
  if (c->q_head != c->q_tail) S4P_FAIL(c, S4P_ERR_STATE, "asynchronous bases outstanding: call s4p_try_base_wait first");
  if (int32_t rc = s4p_try_base_async(c, base_ids, inv1, inv2)) return rc;
  return s4p_try_base_wait(c, result);
}

int32_t s4p_pair_state_words(const s4p_ctx* c) { return c ? int32_t(c->tree.ids.size()) : 0; }
int32_t s4p_pair_state_save(const s4p_ctx* c, uint32_t* out) {
  if (!c || !out) return S4P_ERR_BAD_ARG;
  std::memcpy(out, c->tree.ids.data(), c->tree.ids.size() * sizeof(uint32_t));
  return S4P_OK;
}
int32_t s4p_pair_state_restore(s4p_ctx* c, const uint32_t* in) {
  if (!c || !in) return S4P_ERR_BAD_ARG;
  std::memcpy(c->tree.ids.data(), in, c->tree.ids.size() * sizeof(uint32_t));
  c->tree.forget_splits();          // the remembered cell boundaries belong to the permutation being replaced
  return S4P_OK;
}

int32_t s4p_skip_base(s4p_ctx* c) {
  if (!c) return S4P_ERR_BAD_ARG;
  return s4p_stage_base(c, c->base_xyz, c->base_nrm, 0, 0);
}

namespace {
// The last base took several device passes and nobody kept its records: run it once more in reference-ordered chunks with the
// records kept.  Possible while the base's launch record is intact (its staging slot has not been rewritten).
int32_t replay_for_records(s4p_ctx* c) {
  if (c->kept.valid) return S4P_OK;
  s4p_ctx::Lane& L = c->lane[c->cur];
  if (L.sv_slot < 0 || c->stage[L.sv_slot].gen != L.sv_gen)
    S4P_FAIL(c, S4P_ERR_STATE, "the per-candidate records of a base processed in chunks were not kept and the base can no longer be replayed: "
                               "call s4p_keep_candidate_records(ctx, 1) (or set a sink) before the base");
  const bool keep = c->keep_records;
  c->keep_records = true;
  s4p_base_result again;
  int32_t rc = relaunch_base(c);
  if (rc == S4P_OK) rc = finish_result(c, &again, true);
  c->keep_records = keep;
  return rc;
}
}  // namespace

int32_t s4p_keep_candidate_records(s4p_ctx* c, int32_t enable) {
  if (!c) return S4P_ERR_BAD_ARG;
  c->keep_records = enable != 0;
  return S4P_OK;
}
int32_t s4p_set_candidate_sink(s4p_ctx* c, s4p_candidate_sink sink, void* user) {
  if (!c) return S4P_ERR_BAD_ARG;
  c->sink = sink; c->sink_user = user;
  return S4P_OK;
}

int32_t s4p_last_candidates(s4p_ctx* c, int32_t* quads, int32_t* counts, int64_t cap, int64_t* n_out) {
  if (!c || !n_out) return S4P_ERR_BAD_ARG;
  if (c->last_chunked) {
    HIPCHK(c, hipSetDevice(c->device));
    if (int32_t rc = replay_for_records(c)) return rc;
  }
  // (the replay may have fitted ONE pass -- the lane has grown towards quad_grow_cap since the base was chunked -- and then left
  // its records on the device like any single-pass base: last_chunked says which of the two read-backs applies NOW)
  if (c->last_chunked) {
    const int64_t K = int64_t(c->kept.qcounts.size());
    *n_out = K;
    if (K == 0) return S4P_OK;
    if (cap < K || !quads || !counts) S4P_FAIL(c, S4P_ERR_CAPACITY, "s4p_last_candidates: output buffer too small");
    std::memcpy(quads, c->kept.quads.data(), size_t(K) * 16);
    std::memcpy(counts, c->kept.qcounts.data(), size_t(K) * 4);
    return S4P_OK;
  }
  const uint64_t K = c->last_K;
  *n_out = int64_t(K);
  if (K == 0) return S4P_OK;
  if (cap < int64_t(K) || !quads || !counts) S4P_FAIL(c, S4P_ERR_CAPACITY, "s4p_last_candidates: output buffer too small");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(c->done[c->cur]));
  std::vector<int4> q(K); std::vector<unsigned long long> t(K); std::vector<uint32_t> cn(K);
  HIPCHK(c, hipMemcpy(q.data(), c->lane[c->cur].quads.p, K * 16, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(t.data(), c->lane[c->cur].tags.p, K * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(cn.data(), c->lane[c->cur].counts.p, K * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> order(K);
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t[a] < t[b]; });
  for (uint64_t i = 0; i < K; ++i) {
    const int4 v = q[order[i]];
    quads[4 * i] = v.x; quads[4 * i + 1] = v.y; quads[4 * i + 2] = v.z; quads[4 * i + 3] = v.w;
    counts[i] = cn[order[i]] == kGateFailed ? -1 : int32_t(cn[order[i]]);
  }
  return S4P_OK;
}

// Verified candidates of the base whose s4p_try_base_wait returned last, in reference candidate order:
// inlier count and the 3x4 transform each was scored with (cand_T, kept in HBM by k_gate).
int32_t s4p_last_verified(s4p_ctx* c, uint32_t* counts, float* transforms16, int64_t cap, int64_t* n_out) {
  if (!c || !n_out) return S4P_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->last_chunked) if (int32_t rc = replay_for_records(c)) return rc;
  if (c->last_chunked) {                                   // (else: the replay fitted one pass, see s4p_last_candidates)
    const int64_t C = int64_t(c->kept.counts.size());
    *n_out = C;
    if (C == 0) return S4P_OK;
    if (cap < C || !counts || !transforms16) S4P_FAIL(c, S4P_ERR_CAPACITY, "s4p_last_verified: output buffer too small");
    std::memcpy(counts, c->kept.counts.data(), size_t(C) * 4);
    std::memcpy(transforms16, c->kept.T16.data(), size_t(C) * 64);
    return S4P_OK;
  }
  const s4p_ctx::Lane& L = c->lane[c->cur];
  const uint32_t Cdev = c->hctr[c->cur].p->C;               // as the device counted them (incl. candidates the host rejected afterwards)
  *n_out = 0;
  if (Cdev == 0) return S4P_OK;
  HIPCHK(c, hipEventSynchronize(c->done[c->cur]));
  std::vector<uint32_t> idx(Cdev); std::vector<float4> T(size_t(Cdev) * kCandStride);
  HIPCHK(c, hipMemcpy(idx.data(), L.cand_idx.p, size_t(Cdev) * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(T.data(), L.cand_T.p, size_t(Cdev) * 16 * kCandStride, hipMemcpyDeviceToHost));
  for (auto& k : idx) k &= ~kBorderFlag;
  const uint64_t K = c->last_K;
  std::vector<unsigned long long> t(K); std::vector<uint32_t> cn(K);
  HIPCHK(c, hipMemcpy(t.data(), L.tags.p, K * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(cn.data(), L.counts.p, K * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> order;
  order.reserve(Cdev);
  for (uint32_t a = 0; a < Cdev; ++a) if (cn[idx[a]] != kGateFailed) order.push_back(a);   // (an undecided Euler-angle gate the host rejected)
  const uint32_t C = uint32_t(order.size());
  *n_out = int64_t(C);
  if (C == 0) return S4P_OK;
  if (cap < int64_t(C) || !counts || !transforms16) S4P_FAIL(c, S4P_ERR_CAPACITY, "s4p_last_verified: output buffer too small");
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t[idx[a]] < t[idx[b]]; });
  for (uint32_t i = 0; i < C; ++i) {
    const uint32_t a = order[i];
    counts[i] = cn[idx[a]];
    float* o = transforms16 + 16 * size_t(i);
    std::memcpy(o, &T[size_t(a) * kCandStride], 48);
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
  }
  return S4P_OK;
}

namespace {
void launch_apply(s4p_ctx* c, const float* M, float* dx, float* dy, float* dz, uint64_t n, int variant, hipStream_t st) {
  ApplyParams A{};
  for (int i = 0; i < 12; ++i) A.M[i] = M[i];
  A.x = dx; A.y = dy; A.z = dz; A.n = n;
  // 16 B per lane would need AoS; SoA with 4 B per lane coalesces to full 256 B wave requests, 8192 workgroups keep
  // every CU's queue full for the grid-stride loop
  const uint32_t blocks = uint32_t(std::min<uint64_t>((n + 255) / 256, 8192));
  if (variant == 1) hipLaunchKernelGGL(k_apply_mfma, dim3(blocks), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(k_apply, dim3(blocks), dim3(256), 0, st, A);
}
}  // namespace

// Device-resident form: x, y, z are DEVICE pointers (SoA), transformed in place on the context's stream 0; returns after
// the kernel has completed.  For callers that keep the full-resolution cloud in HBM.
int32_t s4p_transform_points_device(s4p_ctx* c, const float* M, float* dx, float* dy, float* dz, int64_t n) {
  if (!c || !M || (n > 0 && (!dx || !dy || !dz))) return S4P_ERR_BAD_ARG;
  if (n <= 0) return S4P_OK;
  HIPCHK(c, hipSetDevice(c->device));
  launch_apply(c, M, dx, dy, dz, uint64_t(n), 0, c->lane[0].stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->lane[0].stream));
  return S4P_OK;
}

// Host-pointer form (Match4PCSBase::Perform_N_steps tail): the cloud is streamed through two pinned staging buffers in
// chunks, so that the host->pinned copy of chunk k+1, the PCIe transfers and k_apply of chunk k and the pinned->host
// copy of chunk k-1 overlap (a single pageable hipMemcpy per array serialises all of them).
int32_t s4p_transform_points(s4p_ctx* c, const float* M, float* x, float* y, float* z, int64_t n) {
  if (!c || !M || (n > 0 && (!x || !y || !z))) return S4P_ERR_BAD_ARG;
  if (n <= 0) return S4P_OK;
  HIPCHK(c, hipSetDevice(c->device));
  // points per chunk: 1.5 MB per staging buffer.  A whole 1 M-point cloud in ONE chunk (the size until round 4) serialised
  // copy-in, upload, kernel, download and copy-out and paid for 24 MB of pinned memory on a context's first call; with eight
  // chunks the host copies of one chunk run beside the DMA of its neighbours.
  constexpr size_t kChunk = size_t(1) << 17;
  const size_t chunk = std::min<size_t>(kChunk, size_t(n));
  if (c->tbuf_cap < 2 * 3 * chunk) { HIPCHK(c, c->tbuf.alloc(2 * 3 * chunk)); c->tbuf_cap = 2 * 3 * chunk; }
  if (c->tpin.n < 2 * 3 * chunk) HIPCHK(c, c->tpin.alloc(2 * 3 * chunk));
  if (!c->tev[0]) for (auto& e : c->tev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipStream_t st = c->lane[0].stream;
  const size_t nchunks = (size_t(n) + chunk - 1) / chunk;
  auto drain = [&](size_t k) -> hipError_t {                         // pinned -> caller, once chunk k's D2H has landed
    const int b = int(k & 1);
    hipError_t e = hipEventSynchronize(c->tev[b]);
    if (e != hipSuccess) return e;
    const size_t off = k * chunk, m = std::min(chunk, size_t(n) - off);
    const float* p = c->tpin.p + size_t(b) * 3 * chunk;
    std::memcpy(x + off, p, m * 4); std::memcpy(y + off, p + chunk, m * 4); std::memcpy(z + off, p + 2 * chunk, m * 4);
    return hipSuccess;
  };
  for (size_t k = 0; k < nchunks; ++k) {
    const int b = int(k & 1);
    if (k >= 2) HIPCHK(c, drain(k - 2));                             // frees staging buffer b
    const size_t off = k * chunk, m = std::min(chunk, size_t(n) - off);
    float* p = c->tpin.p + size_t(b) * 3 * chunk;
    float* d = c->tbuf.p + size_t(b) * 3 * chunk;
    std::memcpy(p, x + off, m * 4); std::memcpy(p + chunk, y + off, m * 4); std::memcpy(p + 2 * chunk, z + off, m * 4);
    HIPCHK(c, hipMemcpyAsync(d, p, 3 * chunk * 4, hipMemcpyHostToDevice, st));
    launch_apply(c, M, d, d + chunk, d + 2 * chunk, uint64_t(m), 0, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(p, d, 3 * chunk * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->tev[b], st));
  }
  if (nchunks >= 2) HIPCHK(c, drain(nchunks - 2));
  HIPCHK(c, drain(nchunks - 1));
  return S4P_OK;
}

// Measurement aid (bench.py): k_apply on n device-resident synthetic points, `reps` timed launches
// per variant (0 = VALU, the product path; 1 = v_mfma_f32_4x4x1 chain).  out_ms[variant] = mean HIP-event time per
// launch; *mismatch = coordinates where the MFMA result is not bit-identical to the VALU result; *max_abs = their
// largest absolute difference.
// SelectRandomTriangle + the 4th-point scan of SelectQuadrilateral as device reductions (k_select_*): one attempt.
// Safe to call from a thread of its own while bases are in flight: it touches only the selection buffers and stream.
int32_t s4p_select_base_points_batch(s4p_ctx* c, const uint32_t* draws, int32_t n_attempts, float limit_sq, float too_small,
                                     int32_t* ids, float* xyz, int32_t* status) {
  if (!c || !draws || !ids || !xyz || !status || n_attempts < 1 || n_attempts > kSelectBatch) return S4P_ERR_BAD_ARG;
  if (!c->clouds_set || !c->p4o.p) S4P_FAIL(c, S4P_ERR_STATE, "s4p_select_base_points: call s4p_set_clouds first");
  const size_t nd = size_t(kSelectDraws) * size_t(n_attempts);
  for (size_t k = 0; k < nd; ++k)
    if (draws[k] >= c->n_p) S4P_FAIL(c, S4P_ERR_BAD_ARG, "s4p_select_base_points: draw outside the sampled P");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->sel_stream;
  std::memcpy(c->sel_hdraws.p, draws, sizeof(uint32_t) * nd);
  HIPCHK(c, hipMemcpyAsync(c->sel_draws.p, c->sel_hdraws.p, sizeof(uint32_t) * nd, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_select_triangle, dim3(uint32_t(n_attempts)), dim3(1024), 0, st, c->p4o.p, c->sel_draws.p, limit_sq, c->sel_rec.p);
  // (one scan of P serves the whole batch: one point per thread in the first trip, kSelectTile per thread and trip after it; at
  // most one workgroup per CU, each ending in one atomic per attempt)
  const uint32_t blocks = std::max<uint32_t>(1u, std::min<uint32_t>(256u, (c->n_p + uint32_t(kSelectThreads) - 1u) / uint32_t(kSelectThreads)));
  hipLaunchKernelGGL(k_select_fourth, dim3(blocks), dim3(kSelectThreads), 0, st, c->p4o.p, c->n_p, too_small, c->sel_rec.p, n_attempts);
  hipLaunchKernelGGL(k_select_finish, dim3(uint32_t(n_attempts)), dim3(64), 0, st, c->p4o.p, c->sel_rec.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->sel_hrec.p, c->sel_rec.p, sizeof(SelectRecord) * size_t(n_attempts), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  for (int32_t a = 0; a < n_attempts; ++a) {
    const SelectRecord& r = c->sel_hrec.p[a];
    for (int k = 0; k < 4; ++k) ids[4 * a + k] = r.ids[k];
    for (int k = 0; k < 12; ++k) xyz[12 * a + k] = r.xyz[k];
    status[a] = r.status;
  }
  return S4P_OK;
}

int32_t s4p_select_base_points(s4p_ctx* c, const uint32_t* draws, float limit_sq, float too_small,
                               int32_t* ids, float* xyz, int32_t* status) {
  return s4p_select_base_points_batch(c, draws, 1, limit_sq, too_small, ids, xyz, status);
}

int32_t s4p_select_batch_max(void) { return kSelectBatch; }

int32_t s4p_apply_bench(s4p_ctx* c, int64_t n, int32_t reps, double* out_ms, uint64_t* mismatch, float* max_abs) {
  if (!c || n <= 0 || reps <= 0 || !out_ms || !mismatch || !max_abs) return S4P_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t un = size_t(n);
  std::vector<float> h(3 * un);
  uint32_t sd = 12345u;
  for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = (float(sd >> 8) / 16777216.f - 0.5f) * 4.f; }
  const float M[12] = {0.36f, 0.48f, -0.8f, 0.125f, -0.8f, 0.6f, 0.f, -0.75f, 0.48f, 0.64f, 0.6f, 0.3125f};   // a rotation | t
  DevBuf<float> src, a, b;
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = c->lane[0].stream;
  std::vector<float> ra, rb;
  do {
    if (!ok(src.alloc(3 * un)) || !ok(a.alloc(3 * un)) || !ok(b.alloc(3 * un))) break;
    if (!ok(hipEventCreate(&e0)) || !ok(hipEventCreate(&e1))) break;
    if (!ok(hipMemcpy(src.p, h.data(), 3 * un * 4, hipMemcpyHostToDevice))) break;
    for (int variant = 0; variant < 2 && e == hipSuccess; ++variant) {
      float* d = variant == 0 ? a.p : b.p;
      ok(hipMemcpyAsync(d, src.p, 3 * un * 4, hipMemcpyDeviceToDevice, st));
      launch_apply(c, M, d, d + un, d + 2 * un, uint64_t(n), variant, st);                 // result kept for the comparison (and warm-up)
      DevBuf<float> scratch;
      if (!ok(scratch.alloc(3 * un))) break;
      ok(hipMemcpyAsync(scratch.p, src.p, 3 * un * 4, hipMemcpyDeviceToDevice, st));
      ok(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) launch_apply(c, M, scratch.p, scratch.p + un, scratch.p + 2 * un, uint64_t(n), variant, st);
      ok(hipEventRecord(e1, st));
      ok(hipStreamSynchronize(st));
      float ms = 0.f;
      if (e == hipSuccess) ok(hipEventElapsedTime(&ms, e0, e1));
      out_ms[variant] = double(ms) / reps;
      scratch.free();
    }
    if (e != hipSuccess) break;
    ra.resize(3 * un); rb.resize(3 * un);
    if (!ok(hipMemcpy(ra.data(), a.p, 3 * un * 4, hipMemcpyDeviceToHost)) || !ok(hipMemcpy(rb.data(), b.p, 3 * un * 4, hipMemcpyDeviceToHost))) break;
    uint64_t mm = 0; float mx = 0.f;
    for (size_t i = 0; i < 3 * un; ++i) if (std::memcmp(&ra[i], &rb[i], 4) != 0) { ++mm; mx = std::max(mx, std::fabs(ra[i] - rb[i])); }
    *mismatch = mm; *max_abs = mx;
  } while (0);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  src.free(); a.free(); b.free();
  HIPCHK(c, e);
  return S4P_OK;
}

int32_t s4p_profile_enable(s4p_ctx* c, int32_t enable_events, int32_t count_point_tests) {
  if (!c) return S4P_ERR_BAD_ARG;
  c->prof_events = enable_events != 0; c->prof_stages = enable_events == 1; c->prof_points = count_point_tests != 0;
  return S4P_OK;
}
int32_t s4p_profile_get(s4p_ctx* c, s4p_profile* out, int32_t reset) {
  if (!c || !out) return S4P_ERR_BAD_ARG;
  (void)hipSetDevice(c->device);
  for (int li = 0; li < c->n_lanes; ++li) harvest_events(c, li);      // event times of the launches not read yet
  c->prof.host_octree_s = c->host_octree_s; c->prof.host_wait_s = c->host_wait_s;
  if (reset) { c->host_octree_s = 0; c->host_wait_s = 0; }
  *out = c->prof;
  if (reset) c->prof = s4p_profile{};
  return S4P_OK;
}

int32_t s4p_selftest_ieee(s4p_ctx* c, const float* a, const float* b, int64_t n, float* o_sqrt, float* o_div, float* o_ma) {
  if (!c || !a || !b || !o_sqrt || !o_div || !o_ma || n <= 0) return S4P_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf<float> d;
  HIPCHK(c, d.alloc(size_t(n) * 5));
  hipError_t e;
  do {
    if ((e = hipMemcpy(d.p, a, size_t(n) * 4, hipMemcpyHostToDevice)) != hipSuccess) break;
    if ((e = hipMemcpy(d.p + n, b, size_t(n) * 4, hipMemcpyHostToDevice)) != hipSuccess) break;
    hipLaunchKernelGGL(k_selftest, dim3(256), dim3(256), 0, c->lane[c->cur].stream, d.p, d.p + n, uint64_t(n), d.p + 2 * n, d.p + 3 * n, d.p + 4 * n);
    if ((e = hipStreamSynchronize(c->lane[c->cur].stream)) != hipSuccess) break;
    if ((e = hipMemcpy(o_sqrt, d.p + 2 * n, size_t(n) * 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
    if ((e = hipMemcpy(o_div, d.p + 3 * n, size_t(n) * 4, hipMemcpyDeviceToHost)) != hipSuccess) break;
    e = hipMemcpy(o_ma, d.p + 4 * n, size_t(n) * 4, hipMemcpyDeviceToHost);
  } while (0);
  d.free();
  HIPCHK(c, e);
  return S4P_OK;
}

}  // extern "C"
