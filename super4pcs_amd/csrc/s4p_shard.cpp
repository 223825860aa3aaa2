// s4p_shard.cpp -- sharding of RANSAC bases over the GPUs of one node, in C++ behind the C ABI (include/s4p_matcher.h,
// "multi-GPU" section).
//
// One process per GPU.  Every rank walks the SAME sequence of bases (same RNG, same pair-octree state:
// match4pcsBase.cc:185-351 never reads results); the rank that owns a trial runs the fused device pass, the others only
// advance host state.  After a window of `world` consecutive trials (one per rank) ONE all-reduce(MAX) of one packed
// 64-bit key selects the winner exactly as the sequential reference would (match4pcsBase.hpp:467-484: the first strictly
// greater LCP wins; :255: stop at the first trial whose best LCP exceeds the terminate threshold); the winner's result
// record travels by one broadcast, only when the window improved the best LCP.
//
// The collective is 8 bytes: latency-bound, xGMI link bandwidth is irrelevant.  The built-in provider calls RCCL
// (rccl.h: ncclAllReduce / ncclBroadcast on a private HIP stream, key staged through pinned memory, completion by event,
// so the launch thread never blocks on a collective it has just issued).  RCCL is bound at run time (dlopen), reusing the
// copy the process has already loaded if any, so that a process that also uses torch.distributed holds ONE RCCL.
// A caller-supplied provider (s4p_collective: MPI, gloo through callbacks, the tests) plugs into the same loop.
//
// The window loop is written against a small operations table, bound either to a matcher (the public entry points
// next_base / next_base_async / wait_base / commit) or to recorded outcomes (s4p_shard_replay: the host-only
// self-check the CPU tests drive over gloo with 2 and 4 ranks).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <string>
#include <vector>

#include "s4p_matcher.h"

namespace {

constexpr uint64_t kCrossBit = 1ull << 62;
// A rank that failed locally (device error, refused buffer growth, ...) still owes the other ranks its all-reduce of the
// window: it posts this key, which outranks every real one, so that every rank sees the failure in the same window and
// leaves the loop together instead of blocking in a collective for ever.
constexpr uint64_t kErrorKey = 0x7FFFFFFFFFFFFFFFull;      // (below 2^63: providers may carry keys as signed 64-bit integers)

// Packs one trial's outcome so that max() over the window reproduces the sequential reference.
// usable: pairs1, pairs2 and quads all non-empty (otherwise TryOneBase returned before TryCongruentSet).  A trial whose
// count exceeds the terminate threshold outranks everything and, among those, the earliest wins; otherwise the higher
// count wins and ties go to the earliest trial.
uint64_t window_key(uint32_t count, bool has_best, bool usable, uint32_t trial_in_window, uint32_t threshold_count) {
  if (!(has_best && usable)) return 0;
  const uint64_t inv_t = 0xFFFFull - trial_in_window;
  if (count > threshold_count) return kCrossBit | (inv_t << 32) | uint64_t(count);
  return (uint64_t(count) << 16) | inv_t;
}
struct Decoded { bool any; uint32_t trial; uint32_t count; bool crossed; };
Decoded decode_key(uint64_t key) {
  if (key == 0) return {false, 0, 0, false};
  if (key & kCrossBit) return {true, uint32_t(0xFFFFull - ((key >> 32) & 0xFFFFull)), uint32_t(key & 0xFFFFFFFFull), true};
  return {true, uint32_t(0xFFFFull - (key & 0xFFFFull)), uint32_t(key >> 16), false};
}

// Largest inlier count that does NOT cross the terminate threshold: lcp = float(count) / float(n_q) > threshold is
// evaluated in float exactly as commit does (match4pcsBase.cc:566, match4pcsBase.hpp:496).
uint32_t threshold_count_for(uint32_t n_q, float thr) {
  long c = long(std::floor(double(thr) * double(n_q)));
  if (c < 0) c = 0;
  if (c > long(n_q)) c = long(n_q);
  while (c < long(n_q) && !(float(c + 1) / float(n_q) > thr)) ++c;
  while (c >= 0 && (float(c) / float(n_q) > thr)) --c;
  return c < 0 ? 0u : uint32_t(c);
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;              // (optional: what the communicator itself says about its size and this rank)
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;      // the copy already in the process
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) for (const char* n : {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
    CommUserRank = reinterpret_cast<decltype(CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !Broadcast) { err = "librccl lacks an expected symbol"; lib = nullptr; return false; }
    return true;
  }
};
Rccl g_rccl;
static_assert(sizeof(ncclUniqueId) == 128, "s4p_rccl_unique_id hands out 128 bytes");

// ---- collective providers: post() / result() are split so that the RCCL one can be asynchronous ---------------------
struct Collective {
  virtual ~Collective() {}
  virtual int32_t post(int slot, uint64_t key) = 0;          // all-reduce(MAX) of one key; slot in {0, 1, 2}
  virtual int32_t result(int slot, uint64_t* key) = 0;
  virtual int32_t broadcast(void* buf, size_t bytes, int root) = 0;
  std::string err;
};

struct CallbackCollective : Collective {                     // caller-supplied, synchronous
  s4p_collective c;
  uint64_t val[3] = {0, 0, 0};
  explicit CallbackCollective(const s4p_collective& cc) : c(cc) {}
  int32_t post(int slot, uint64_t key) override {
    val[slot] = key;
    const int32_t rc = c.allreduce_max_u64(c.user, &val[slot]);
    if (rc) err = "caller's allreduce_max_u64 failed";
    return rc ? S4P_ERR_STATE : S4P_OK;
  }
  int32_t result(int slot, uint64_t* key) override { *key = val[slot]; return S4P_OK; }
  int32_t broadcast(void* buf, size_t bytes, int root) override {
    const int32_t rc = c.broadcast(c.user, buf, int64_t(bytes), root);
    if (rc) err = "caller's broadcast failed";
    return rc ? S4P_ERR_STATE : S4P_OK;
  }
};

// Measurement aid (tools/sim_world.py): this process plays ONE rank of a `world`-rank job on one GPU -- it selects and
// stages every trial of every window and runs its own device passes, as a real rank does; the reduction returns its own
// key and the broadcast is a no-op.  What it measures is the per-window cost of a rank, i.e. whether the replicated host
// chain or the GPU pass bounds a rank at that world size.  Never a product path.
struct NullCollective : Collective {
  uint64_t val[3] = {0, 0, 0};
  int32_t post(int slot, uint64_t key) override { val[slot] = key; return S4P_OK; }
  int32_t result(int slot, uint64_t* key) override { *key = val[slot]; return S4P_OK; }
  int32_t broadcast(void*, size_t, int) override { return S4P_OK; }
};

struct RcclCollective : Collective {                         // RCCL over xGMI, one communicator per matcher
  int device = 0, rank = 0, world = 1;
  int comm_count = -1, comm_rank = -1;                       // as RCCL reports them (ncclCommCount / ncclCommUserRank); -1: symbol not bound
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  uint64_t* host = nullptr;                                  // pinned: [0..2] keys (two window slots + the time-budget vote), then a 256-byte record
  uint64_t* dev = nullptr;
  static constexpr size_t kRecBytes = 256, kKeyBytes = 32, kBytes = kKeyBytes + kRecBytes;
  bool hip_ok(hipError_t e, const char* what) { if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; }
  bool nccl_ok(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) { err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"); return false; }
    return true;
  }
  int32_t init(int dev_index, int r, int w, const uint8_t* id128) {
    device = dev_index; rank = r; world = w;
    if (!g_rccl.load()) { err = g_rccl.err; return S4P_ERR_STATE; }
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return S4P_ERR_HIP;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    if (!nccl_ok(g_rccl.CommInitRank(&comm, world, id, rank), "ncclCommInitRank")) return S4P_ERR_STATE;
    // the communicator's own view of the job: a line that says "8 ranks" must be RCCL's statement, not the launcher's
    if (g_rccl.CommCount && !nccl_ok(g_rccl.CommCount(comm, &comm_count), "ncclCommCount")) return S4P_ERR_STATE;
    if (g_rccl.CommUserRank && !nccl_ok(g_rccl.CommUserRank(comm, &comm_rank), "ncclCommUserRank")) return S4P_ERR_STATE;
    if ((comm_count >= 0 && comm_count != world) || (comm_rank >= 0 && comm_rank != rank)) {
      err = "the RCCL communicator reports " + std::to_string(comm_count) + " ranks / rank " + std::to_string(comm_rank) + ", the shard was created as rank " + std::to_string(rank) + " of " + std::to_string(world);
      return S4P_ERR_STATE;
    }
    if (!hip_ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate")) return S4P_ERR_HIP;
    for (auto& e : ev) if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return S4P_ERR_HIP;
    if (!hip_ok(hipHostMalloc((void**)&host, kBytes, hipHostMallocDefault), "hipHostMalloc")) return S4P_ERR_HIP;
    if (!hip_ok(hipMalloc((void**)&dev, kBytes), "hipMalloc")) return S4P_ERR_HIP;
    // First use of a communicator sets up its channels (and loads the collective's kernel): tens of milliseconds, against a
    // window of ~0.1 ms.  Both collectives of the loop run once here, where every rank is anyway (communicator set-up is
    // collective), so that the first window and the first improved result of a registration do not pay for it.
    if (!hip_ok(hipMemsetAsync(dev, 0, kBytes, stream), "hipMemsetAsync")) return S4P_ERR_HIP;
    if (!nccl_ok(g_rccl.AllReduce(dev, dev, 1, ncclUint64, ncclMax, comm, stream), "ncclAllReduce (set-up)")) return S4P_ERR_STATE;
    char* drec = reinterpret_cast<char*>(dev) + kKeyBytes;
    if (!nccl_ok(g_rccl.Broadcast(drec, drec, sizeof(s4p_base_result), ncclChar, 0, comm, stream), "ncclBroadcast (set-up)")) return S4P_ERR_STATE;
    if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return S4P_ERR_HIP;
    return S4P_OK;
  }
  ~RcclCollective() override {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (host) (void)hipHostFree(host);
    if (dev) (void)hipFree(dev);
    if (stream) (void)hipStreamDestroy(stream);
  }
  int32_t post(int slot, uint64_t key) override {            // pinned -> device -> ncclAllReduce -> pinned, closed by an event
    host[slot] = key;
    if (!hip_ok(hipMemcpyAsync(dev + slot, host + slot, 8, hipMemcpyHostToDevice, stream), "hipMemcpyAsync")) return S4P_ERR_HIP;
    if (!nccl_ok(g_rccl.AllReduce(dev + slot, dev + slot, 1, ncclUint64, ncclMax, comm, stream), "ncclAllReduce")) return S4P_ERR_STATE;
    if (!hip_ok(hipMemcpyAsync(host + slot, dev + slot, 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync")) return S4P_ERR_HIP;
    if (!hip_ok(hipEventRecord(ev[slot], stream), "hipEventRecord")) return S4P_ERR_HIP;
    return S4P_OK;
  }
  int32_t result(int slot, uint64_t* key) override {
    if (!hip_ok(hipEventSynchronize(ev[slot]), "hipEventSynchronize")) return S4P_ERR_HIP;
    *key = host[slot];
    return S4P_OK;
  }
  int32_t broadcast(void* buf, size_t bytes, int root) override {
    if (bytes > kRecBytes) { err = "broadcast record too large"; return S4P_ERR_BAD_ARG; }
    char* hrec = reinterpret_cast<char*>(host) + kKeyBytes;
    char* drec = reinterpret_cast<char*>(dev) + kKeyBytes;
    if (rank == root) std::memcpy(hrec, buf, bytes);
    if (!hip_ok(hipMemcpyAsync(drec, hrec, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync")) return S4P_ERR_HIP;
    if (!nccl_ok(g_rccl.Broadcast(drec, drec, bytes, ncclChar, root, comm, stream), "ncclBroadcast")) return S4P_ERR_STATE;
    if (!hip_ok(hipMemcpyAsync(hrec, drec, bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync")) return S4P_ERR_HIP;
    if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return S4P_ERR_HIP;
    std::memcpy(buf, hrec, bytes);
    return S4P_OK;
  }
};

// ---- the window loop -----------------------------------------------------------------------------------------------
struct TrialOps {                                            // what the loop needs from "a matcher"
  std::function<int32_t(bool run_device, bool* found, int32_t ids[4])> prepare;     // own trial: enqueue; other ranks': advance host state
  std::function<int32_t(s4p_base_result*)> wait_own;                                // result of the oldest enqueued own trial
  std::function<int32_t(const int32_t ids[4], const s4p_base_result*, bool* ok)> commit;
  int depth = 2;
};

struct BaseId { bool found = false; int32_t ids[4] = {0, 0, 0, 0}; int32_t failed_rc = 0; };      // failed_rc: (split loop) this rank could not even enqueue the trial
struct Window { std::vector<BaseId> bases; bool mine_found = false; s4p_base_result r{}; uint64_t verified = 0; int slot = -1; };

struct Loop {
  int rank = 0, world = 1;
  Collective* coll = nullptr;
  TrialOps ops;
  uint32_t threshold_count = 0, best_count = 0;
  bool terminated = false;
  uint64_t trials_done = 0, local_candidates = 0;
  uint64_t trials_prepared = 0, trial_limit = ~0ull;     // trials of the common sequence handed out so far / never run from here on
  int slot_rr = 0;
  std::string err;
  bool coll_failed = false;               // the collective itself failed (not "some rank posted the error key"): nothing further is posted
  int32_t fail(int32_t rc, const std::string& m) { err = m; return rc; }
  int32_t cfail(int32_t rc) { coll_failed = true; return fail(rc, coll->err); }

  int32_t prepare_window(Window& w) {
    w.bases.resize(size_t(world));
    for (int j = 0; j < world; ++j) {
      bool found = false;
      // trials past the limit (the tail of the last window of a registration) are not run by anybody
      if (trials_prepared + uint64_t(j) < trial_limit)
        if (int32_t rc = ops.prepare(j == rank, &found, w.bases[size_t(j)].ids)) return rc;
      w.bases[size_t(j)].found = found;
      if (j == rank) w.mine_found = found;
    }
    trials_prepared += uint64_t(world);
    return S4P_OK;
  }
  int32_t post_window(Window& w) {
    std::memset(&w.r, 0, sizeof w.r);
    if (w.mine_found) if (int32_t rc = ops.wait_own(&w.r)) return rc;
    w.verified = w.r.n_verified;
    local_candidates += w.verified;
    w.slot = -1;
    if (!terminated) {
      const bool usable = w.mine_found && w.r.n_pairs1 && w.r.n_pairs2 && w.r.n_quads;
      const uint64_t key = window_key(w.r.best_count, w.r.has_best != 0, usable, uint32_t(rank), threshold_count);
      w.slot = slot_rr;
      slot_rr ^= 1;
      if (int32_t rc = coll->post(w.slot, key)) return cfail(rc);
    }
    return S4P_OK;
  }
  bool commit_failed = false;             // complete_window failed in THIS rank's commit: a local failure (the collectives of the window are done)
  int32_t complete_window(Window& w) {
    trials_done += uint64_t(world);
    if (w.slot < 0) return S4P_OK;
    uint64_t key = 0;
    if (int32_t rc = coll->result(w.slot, &key)) return cfail(rc);      // always consumed: the slot is reused
    if (key == kErrorKey) return fail(S4P_ERR_STATE, "a rank of the sharded job failed in this window (see that rank's error)");
    if (terminated) return S4P_OK;                           // a window posted before the threshold was crossed: not committed
    const Decoded d = decode_key(key);
    if (d.any && d.count > best_count) {                     // the window improved the best LCP: fetch the winner's record
      if (d.trial >= uint32_t(world)) return fail(S4P_ERR_STATE, "corrupt window key");
      s4p_base_result wr = w.r;                              // (the owner's own record; everybody else receives it)
      if (int32_t rc = coll->broadcast(&wr, sizeof wr, int(d.trial))) return cfail(rc);
      // the crossing is read off the KEY, before the commit: a rank whose commit fails must still agree with the others on
      // whether the job has stopped posting
      terminated = terminated || d.crossed;
      best_count = wr.best_count;
      bool ok = false;
      if (int32_t rc = ops.commit(w.bases[d.trial].ids, &wr, &ok)) { commit_failed = true; return rc; }
      terminated = terminated || ok;
    }
    return S4P_OK;
  }
  // Closes every call of run() on every rank: one more 8-byte all-reduce(MAX) of "did this rank fail locally".  A failure the
  // window protocol could not carry any more -- the commit of the LAST window of a call, a failure at depth 1 -- reaches the
  // others here instead of leaving them to pair their next call's collectives with a rank that has left.  The
  // ranks are aligned when they get here: a local failure has posted the error keys it owed (leave), a rank that READ an
  // error key has posted exactly what the failing rank answered.  Not after a failure of the collective itself.
  int32_t close_call(int32_t rc, bool local_failure, bool collective_broken) {
    if (!coll || collective_broken) return rc;
    const std::string keep = err;
    uint64_t g = 0;
    if (coll->post(2, local_failure ? kErrorKey : 0ull) != S4P_OK || coll->result(2, &g) != S4P_OK) { if (rc == S4P_OK) return fail(S4P_ERR_STATE, coll->err); err = keep; return rc; }
    if (rc == S4P_OK && g == kErrorKey) return fail(S4P_ERR_STATE, "a rank of the sharded job failed at the end of this call (see that rank's error)");
    err = keep;
    return rc;
  }
  // n windows (n * world trials) through a three-stage software pipeline: window w+d is PREPARED (own device pass
  // enqueued, the other ranks' bases advanced on the host) while the passes of windows w+1..w+d-1 are in flight;
  // window w's result is waited for and its key POSTED; only then is the reduction of window w-1 COMPLETED.
  int32_t run(int n) {
    std::deque<Window> prepared;
    Window posted; bool have_posted = false;
    int posted_n = 0;                                        // windows of this call whose key this rank has posted
    bool remote_error = false;
    commit_failed = false; coll_failed = false;
    auto advance = [&]() -> int32_t {
      Window nxt = std::move(prepared.front());
      prepared.pop_front();
      if (int32_t rc = post_window(nxt)) return rc;
      if (nxt.slot >= 0) ++posted_n;
      if (have_posted) if (int32_t rc = complete_window(posted)) {
        if (commit_failed) { posted = std::move(nxt); have_posted = true; }      // local: the window just posted is still owed its completion (leave)
        else remote_error = true;
        return rc;
      }
      posted = std::move(nxt);
      have_posted = true;
      return S4P_OK;
    };
    // A LOCAL failure while window W (= posted_n) was being prepared or waited for.  The healthy ranks go on with
    //   reduce(W), complete(W-1) [result, and a BROADCAST of its winner if it improved the best], reduce(W+1), complete(W): error
    // so this rank owes them exactly that sequence: the error key as its reduction of W, the completion of the window it has
    // already posted (the broadcast is a collective too: leaving it out pairs the others' broadcast with this rank's next
    // all-reduce -- ADVICE r03), then the error key once more for W+1.  If W-1 turns out to have crossed the terminate
    // threshold nobody posts W+1; the others still read the error out of W's reduction.  A failed COMMIT (of window W-1, with W
    // already posted) is the same situation one window later: `posted` is then W, the error keys go out for W+1 and W+2.
    auto leave = [&](int32_t rc) -> int32_t {
      if (coll_failed) return rc;                            // the collective is gone: nothing can be told to anybody
      if (rc == S4P_OK || remote_error || !coll) return close_call(rc, false, false);      // (remote: this rank read the error key)
      if (terminated) return close_call(rc, true, false);
      const std::string keep = err;
      bool broken = false;
      for (int k = 0; k < 2 && posted_n < n; ++k, ++posted_n) {
        const int slot = slot_rr; slot_rr ^= 1;
        uint64_t dummy = 0;
        if (coll->post(slot, kErrorKey) != S4P_OK) { broken = true; break; }
        if (k == 0 && have_posted) {
          have_posted = false;
          if (complete_window(posted) != S4P_OK && !commit_failed) { (void)coll->result(slot, &dummy); broken = true; break; }      // (another rank failed too, or the collective did)
        }
        if (coll->result(slot, &dummy) != S4P_OK) { broken = true; break; }
        if (terminated) break;
      }
      // the window this rank had already posted when nothing further is posted in this call (the last window): the others
      // complete it -- result and, if it improved the best, a broadcast -- and so does this rank
      if (!broken && have_posted) { have_posted = false; const bool cf = commit_failed; if (complete_window(posted) != S4P_OK && !commit_failed) broken = true; commit_failed = cf || commit_failed; }
      err = keep;
      return close_call(rc, true, broken || coll_failed);
    };
    for (int w = 0; w < n; ++w) {
      if (terminated) break;                                 // threshold crossed: no further bases are selected or launched
      prepared.emplace_back();
      if (int32_t rc = prepare_window(prepared.back())) return leave(rc);
      if (int(prepared.size()) >= ops.depth) if (int32_t rc = advance()) return leave(rc);
    }
    while (!prepared.empty()) if (int32_t rc = advance()) return leave(rc);
    if (have_posted) {
      have_posted = false;
      if (int32_t rc = complete_window(posted)) { if (!commit_failed) remote_error = true; return leave(rc); }
    }
    return close_call(S4P_OK, false, false);
  }
};

// ---- SURVEY 8e level 2: every base over ALL ranks ---------------------------------------------------------------------
// Every rank runs every trial, but enumerates, gates and scores only its share of the base's second pair set
// (s4p_set_quad_slice).  Per trial two 8-byte all-reduce(MAX) pick the base's winner among the shares exactly as a single
// pass would -- greatest inlier count, then smallest order tag (match4pcsBase.hpp:467-484):
//   key A = (count + 1) << 32 | (0xFFFFFFFF - tag.hi)      (0 = this share verified nothing)
//   key B = (0xFFFFFFFF - tag.lo) << 16 | (rank + 1)       from the ranks whose (count, tag.hi) won A, else 0
// and one broadcast from the rank B names carries the winner's record when it improves the registration's best.  For
// bases that take seconds (the 20 000-point sample) the three collectives are noise; for cheap bases sharding by base
// (the window loop above) is the mode to use.
struct SplitOps {
  std::function<int32_t(bool* found, int32_t ids[4])> prepare;           // enqueue this rank's share of the next trial
  std::function<int32_t(s4p_base_result*)> wait_own;
  std::function<int32_t(const int32_t ids[4], const s4p_base_result*, bool* ok)> commit;
  int depth = 2;
};
struct SplitLoop {
  int rank = 0, world = 1;
  Collective* coll = nullptr;
  SplitOps ops;
  uint32_t best_count = 0;
  bool terminated = false;
  uint64_t trials_done = 0, local_candidates = 0;
  int32_t local_fail = 0;                 // a local failure the other ranks have not been told about yet (a failed commit): the next reduction carries the error key
  bool commit_failed = false;
  std::string err;
  int32_t fail(int32_t rc, const std::string& m) { err = m; return rc; }

  uint32_t threshold_count = 0xFFFFFFFFu; // largest inlier count that does not cross the terminate threshold (threshold_count_for)
  bool remote_error = false, coll_failed = false;
  int32_t cfail(int32_t rc) { coll_failed = true; return fail(rc, coll->err); }

  int32_t reduce_and_commit(const BaseId& b, const s4p_base_result& r) {
    const bool usable = b.found && r.n_pairs1 && r.n_pairs2 && r.n_quads && r.has_best;
    const uint32_t thi = uint32_t(r.best_rank >> 32), tlo = uint32_t(r.best_rank);
    uint64_t a = usable ? ((uint64_t(r.best_count) + 1ull) << 32) | uint64_t(0xFFFFFFFFu - thi) : 0ull, ga = 0;
    if (b.found && !usable && r.n_pairs1 == ~0ull) a = kErrorKey;             // (a rank whose pass failed: see run())
    if (local_fail) a = kErrorKey;
    if (int32_t rc = coll->post(0, a)) return cfail(rc);
    if (int32_t rc = coll->result(0, &ga)) return cfail(rc);
    if (ga == kErrorKey) { remote_error = a != kErrorKey; return fail(S4P_ERR_STATE, "a rank of the sharded job failed on this base (see that rank's error)"); }
    const uint64_t kb = (ga != 0 && a == ga) ? (uint64_t(0xFFFFFFFFu - tlo) << 16) | uint64_t(rank + 1) : 0ull;
    uint64_t gb = 0;
    if (int32_t rc = coll->post(1, kb)) return cfail(rc);
    if (int32_t rc = coll->result(1, &gb)) return cfail(rc);
    if (ga == 0 || gb == 0) return S4P_OK;                                     // no share of this base verified a candidate
    const uint32_t win_count = uint32_t(ga >> 32) - 1u;
    const int root = int(gb & 0xFFFFull) - 1;
    if (root < 0 || root >= world) return fail(S4P_ERR_STATE, "corrupt split-base key");
    if (win_count > best_count) {                                              // only an improvement travels (and is committed)
      s4p_base_result wr = r;
      if (int32_t rc = coll->broadcast(&wr, sizeof wr, root)) return cfail(rc);
      wr.n_quads = std::max<uint64_t>(wr.n_quads, 1);                          // (some share had quads: TryOneBase went on to TryCongruentSet)
      // the crossing is read off the reduced count, before the commit: a rank whose commit fails must still agree with the
      // others on whether the job has stopped reducing
      terminated = terminated || win_count > threshold_count;
      best_count = wr.best_count;
      bool ok = false;
      if (int32_t rc = ops.commit(b.ids, &wr, &ok)) { commit_failed = true; return rc; }
      terminated = terminated || ok;
    }
    return S4P_OK;
  }
  // Closes every call of run() on every rank (as Loop::close_call): one 8-byte all-reduce(MAX) of "did this rank fail
  // locally", so that a failure the per-trial reductions could not carry any more -- the commit of the LAST trial of a call --
  // still reaches every rank inside this call.
  int32_t close_call(int32_t rc, bool local_failure) {
    if (!coll || coll_failed) return rc;
    const std::string keep = err;
    uint64_t g = 0;
    if (coll->post(2, local_failure ? kErrorKey : 0ull) != S4P_OK || coll->result(2, &g) != S4P_OK) { if (rc == S4P_OK) return fail(S4P_ERR_STATE, coll->err); err = keep; return rc; }
    if (rc == S4P_OK && g == kErrorKey) return fail(S4P_ERR_STATE, "a rank of the sharded job failed at the end of this call (see that rank's error)");
    err = keep;
    return rc;
  }
  // n trials, `depth` of them in flight on the device; the reduction of trial t runs while t+1 .. t+depth-1 compute.
  // Every local failure -- a trial that cannot be enqueued (HIP error, device selection error: NOT symmetric over the ranks),
  // a pass that fails, a commit that fails -- reaches the other ranks as the error key in the reduction they are waiting in
  //  or, when this call has no reduction left, in the status reduction that closes the call; no device
  // pass is left un-waited when the loop is left.
  int32_t run(int n) {
    std::deque<BaseId> inflight;
    remote_error = false; coll_failed = false; commit_failed = false; local_fail = 0;
    auto drain_one = [&]() -> int32_t {
      BaseId b = inflight.front(); inflight.pop_front();
      s4p_base_result r; std::memset(&r, 0, sizeof r);
      int32_t rc = b.failed_rc;
      if (rc == S4P_OK && b.found) rc = ops.wait_own(&r);
      if (rc != S4P_OK) { std::memset(&r, 0, sizeof r); r.n_pairs1 = ~0ull; b.found = true; }     // tell the others, then leave
      local_candidates += rc == S4P_OK ? r.n_verified : 0;
      ++trials_done;
      if (terminated) return rc;                            // drained, not committed (the sequential loop would not have run it)
      const std::string keep = err;
      const int32_t rc2 = reduce_and_commit(b, r);
      if (rc != S4P_OK) { err = keep; return rc; }
      return rc2;
    };
    // leaving with an error: the shares still in flight are waited for (their results are dropped); after a failed commit
    // the next reduction of this call, if there is one, carries the error key; the status reduction closes the call
    // more: this call still has trials the OTHER ranks will reduce (they committed the window this rank failed to commit and go
    // on).  With nothing in flight here (depth 1, or the failing commit drained the last share in flight) their next per-trial
    // reduction must still find a partner: collectives pair by order, not by slot, and it would otherwise pair with this rank's
    // closing status reduction and leave their own closing reduction without one.
    auto leave = [&](int32_t rc, bool more) -> int32_t {
      const std::string keep = err;
      const bool local = !remote_error;
      if (commit_failed && !inflight.empty() && !terminated) { local_fail = rc; (void)drain_one(); }
      else if (commit_failed && inflight.empty() && !terminated && more && !coll_failed) {
        uint64_t g = 0;
        if (coll->post(0, kErrorKey) == S4P_OK) (void)coll->result(0, &g);
      }
      while (!inflight.empty()) {
        const BaseId b = inflight.front(); inflight.pop_front();
        s4p_base_result r;
        if (b.failed_rc == S4P_OK && b.found) (void)ops.wait_own(&r);
      }
      local_fail = 0; commit_failed = false;
      err = keep;
      return close_call(rc, local);
    };
    for (int t = 0; t < n && !terminated; ++t) {
      BaseId b;
      const int32_t prc = ops.prepare(&b.found, b.ids);
      if (prc != S4P_OK) { b.found = true; b.failed_rc = prc; }
      inflight.push_back(b);
      if (prc != S4P_OK) break;                              // the earlier trials are reduced in order, then this one posts the error key
      if (int(inflight.size()) >= ops.depth) if (int32_t rc = drain_one()) return leave(rc, t + 1 < n);
    }
    while (!inflight.empty()) if (int32_t rc = drain_one()) return leave(rc, false);
    return close_call(S4P_OK, false);
  }
};

}  // namespace

struct s4p_shard {
  s4p_matcher* m = nullptr;
  int mode = 0;                        // 0: trials sharded by base (window loop), 1: every base split over all ranks
  int producer_threads = 2;            // the helper-thread policy given to s4p_shard_create (0 never, 1 always, 2 where they pay): set_mode keeps it
  SplitLoop split;
  int64_t init_generation = -1;        // the matcher initialisation the loop state belongs to
  Loop loop;
  Collective* coll = nullptr;
  std::string err;
  ~s4p_shard() { delete coll; }
};

extern "C" {

int32_t s4p_rccl_unique_id(uint8_t* out128) {
  if (!out128) return S4P_ERR_BAD_ARG;
  if (!g_rccl.load()) return S4P_ERR_STATE;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return S4P_ERR_STATE;
  std::memcpy(out128, &id, sizeof id);
  return S4P_OK;
}

int32_t s4p_shard_create(s4p_matcher* m, int32_t rank, int32_t world, int32_t producer_threads, s4p_shard** out) {
  if (!m || !out || world < 1 || rank < 0 || rank >= world || world > 0xFFFF) return S4P_ERR_BAD_ARG;
  *out = nullptr;
  if (int32_t rc = s4p_matcher_set_sharding(m, rank, world, producer_threads)) return rc;
  s4p_shard* s = new s4p_shard();
  s->m = m;
  s->producer_threads = producer_threads;
  s->loop.rank = rank; s->loop.world = world;
  *out = s;
  return S4P_OK;
}

void s4p_shard_destroy(s4p_shard* s) { delete s; }
const char* s4p_shard_last_error(const s4p_shard* s) { return s ? s->err.c_str() : "null shard"; }

int32_t s4p_shard_use_rccl(s4p_shard* s, int32_t device, const uint8_t* unique_id128) {
  if (!s || !unique_id128) return S4P_ERR_BAD_ARG;
  RcclCollective* c = new RcclCollective();
  const int32_t rc = c->init(device, s->loop.rank, s->loop.world, unique_id128);
  if (rc) { s->err = c->err; delete c; return rc; }
  delete s->coll;
  s->coll = c;
  return S4P_OK;
}

int32_t s4p_shard_comm_info(const s4p_shard* s, int32_t* n_ranks, int32_t* rank) {
  if (!s || !n_ranks || !rank) return S4P_ERR_BAD_ARG;
  *n_ranks = -1; *rank = -1;
  if (const RcclCollective* c = dynamic_cast<const RcclCollective*>(s->coll)) { *n_ranks = c->comm_count; *rank = c->comm_rank; }
  return S4P_OK;
}

int32_t s4p_shard_use_null_collective(s4p_shard* s) {
  if (!s) return S4P_ERR_BAD_ARG;
  delete s->coll;
  s->coll = new NullCollective();
  return S4P_OK;
}

int32_t s4p_shard_use_collective(s4p_shard* s, const s4p_collective* coll) {
  if (!s || !coll || !coll->allreduce_max_u64 || !coll->broadcast) return S4P_ERR_BAD_ARG;
  delete s->coll;
  s->coll = new CallbackCollective(*coll);
  return S4P_OK;
}

int32_t s4p_shard_set_mode(s4p_shard* s, int32_t mode) {
  if (!s || (mode != 0 && mode != 1)) return S4P_ERR_BAD_ARG;
  s4p_matcher* m = s->m;
  // split mode: this rank runs EVERY trial (it owns them all as far as base selection and staging go) on its share of the quads
  if (int32_t rc = s4p_matcher_set_sharding(m, mode ? 0 : s->loop.rank, mode ? 1 : s->loop.world, s->producer_threads)) return rc;
  if (int32_t rc = s4p_set_quad_slice(s4p_matcher_ctx(m), mode ? uint32_t(s->loop.rank) : 0u, mode ? uint32_t(s->loop.world) : 0u)) return rc;
  s->mode = mode;
  return S4P_OK;
}

static int32_t shard_run_split(s4p_shard* s, int32_t n_trials, uint64_t* candidates_local, int32_t* terminated) {
  s4p_matcher* m = s->m;
  s4p_matcher_info info;
  if (int32_t rc = s4p_matcher_get_info(m, &info)) { s->err = s4p_matcher_last_error(m); return rc; }
  SplitLoop& L = s->split;
  if (s->init_generation != s4p_matcher_init_generation(m)) { s->init_generation = s4p_matcher_init_generation(m); L.terminated = false; L.trials_done = 0; }
  L.rank = s->loop.rank; L.world = s->loop.world; L.coll = s->coll;
  L.best_count = info.best_count;
  L.threshold_count = threshold_count_for(uint32_t(info.n_sampled_q), s4p_matcher_terminate_threshold(m));
  L.ops.depth = std::max(1, s4p_pipeline_depth(s4p_matcher_ctx(m)));
  L.ops.prepare = [m, s](bool* found, int32_t ids[4]) -> int32_t {
    int32_t f = 0;
    const int32_t rc = s4p_matcher_next_base_async(m, 1, &f, ids);
    if (rc) s->err = s4p_matcher_last_error(m);
    *found = f != 0;
    return rc;
  };
  L.ops.wait_own = [m, s](s4p_base_result* r) -> int32_t {
    const int32_t rc = s4p_matcher_wait_base(m, r);
    if (rc) s->err = s4p_matcher_last_error(m);
    return rc;
  };
  L.ops.commit = [m, s](const int32_t ids[4], const s4p_base_result* r, bool* ok) -> int32_t {
    int32_t o = 0;
    const int32_t rc = s4p_matcher_commit(m, 1, ids, r, &o);
    if (rc) s->err = s4p_matcher_last_error(m);
    *ok = o != 0;
    return rc;
  };
  const uint64_t before = L.local_candidates;
  (void)s4p_matcher_loop_begin(m);
  const int32_t rc = L.run(n_trials);
  (void)s4p_matcher_loop_end(m);
  if (rc && s->err.empty()) s->err = L.err;
  if (candidates_local) *candidates_local = L.local_candidates - before;
  if (terminated) *terminated = L.terminated ? 1 : 0;
  return rc;
}

int32_t s4p_shard_run_windows(s4p_shard* s, int32_t n_windows, uint64_t* candidates_local, int32_t* terminated) {
  if (!s || n_windows < 0) return S4P_ERR_BAD_ARG;
  if (!s->coll) { s->err = "no collective: call s4p_shard_use_rccl or s4p_shard_use_collective first"; return S4P_ERR_STATE; }
  if (s->mode == 1) return shard_run_split(s, n_windows, candidates_local, terminated);      // (split mode: a "window" is one trial)
  s4p_matcher* m = s->m;
  s4p_matcher_info info;
  if (int32_t rc = s4p_matcher_get_info(m, &info)) { s->err = s4p_matcher_last_error(m); return rc; }
  Loop& L = s->loop;
  if (s->init_generation != s4p_matcher_init_generation(m)) {      // the matcher was (re-)initialised: a new registration
    s->init_generation = s4p_matcher_init_generation(m);
    L.terminated = false; L.trials_done = 0; L.trials_prepared = 0; L.slot_rr = 0;
  }
  L.coll = s->coll;
  L.best_count = info.best_count;
  L.threshold_count = threshold_count_for(uint32_t(info.n_sampled_q), s4p_matcher_terminate_threshold(m));
  L.ops.depth = std::max(1, s4p_pipeline_depth(s4p_matcher_ctx(m)));
  L.ops.prepare = [m, s](bool run_device, bool* found, int32_t ids[4]) -> int32_t {
    int32_t f = 0;
    int32_t rc;
    if (run_device) rc = s4p_matcher_next_base_async(m, 1, &f, ids);
    else { s4p_base_result dummy; rc = s4p_matcher_next_base(m, 0, &f, ids, &dummy); }
    if (rc) s->err = s4p_matcher_last_error(m);
    *found = f != 0;
    return rc;
  };
  L.ops.wait_own = [m, s](s4p_base_result* r) -> int32_t {
    const int32_t rc = s4p_matcher_wait_base(m, r);
    if (rc) s->err = s4p_matcher_last_error(m);
    return rc;
  };
  L.ops.commit = [m, s](const int32_t ids[4], const s4p_base_result* r, bool* ok) -> int32_t {
    int32_t o = 0;
    const int32_t rc = s4p_matcher_commit(m, 1, ids, r, &o);
    if (rc) s->err = s4p_matcher_last_error(m);
    *ok = o != 0;
    return rc;
  };
  const uint64_t before = L.local_candidates;
  (void)s4p_matcher_loop_begin(m);                          // commits refresh the device's best-count hint (early exit)
  const int32_t rc = L.run(n_windows);
  (void)s4p_matcher_loop_end(m);
  if (rc && s->err.empty()) s->err = L.err;
  if (candidates_local) *candidates_local = L.local_candidates - before;
  if (terminated) *terminated = L.terminated ? 1 : 0;
  return rc;
}

// Whole ComputeTransformation over `world` GPUs: every rank initialises identically (same clouds, same seed), the trial
// loop runs in windows until the trial budget is spent or the terminate threshold is crossed, and every rank ends with
// the same best transform (commit runs everywhere); the caller's Q is transformed on every rank that passes buffers.
int32_t s4p_shard_compute_transformation(s4p_shard* s, const s4p_cloud_view* P, const s4p_cloud_view* Q,
                                         float* qx, float* qy, float* qz, float* M, float* lcp) {
  if (!s || !M || !lcp) return S4P_ERR_BAD_ARG;
  s4p_matcher* m = s->m;
  *lcp = 1e9f;                                                     // kLargeNumber, match4pcsBase.hpp:69-70
  if (!P || !Q || P->n == 0 || Q->n == 0) return S4P_OK;
  if (int32_t rc = s4p_matcher_init_full(m, P, Q)) { s->err = s4p_matcher_last_error(m); return rc; }
  s->init_generation = s4p_matcher_init_generation(m);
  s->loop.terminated = false; s->loop.trials_done = 0; s->loop.trials_prepared = 0; s->loop.slot_rr = 0;
  s4p_matcher_info info;
  if (int32_t rc = s4p_matcher_get_info(m, &info)) return rc;
  const float lcp0 = info.best_lcp;
  if (info.best_lcp != 1.f) {
    // Perform_N_steps (match4pcsBase.hpp:236-256) runs trial i = 0, 1, ... and leaves the loop after the first trial with
    // max(time fraction, float(i) / float(number_of_trials)) >= 0.99, or when the terminate threshold is crossed:
    const int N = info.number_of_trials;
    int i_stop = N - 1;
    for (int i = 0; i < N; ++i) if (float(i) / float(N) >= 0.99f) { i_stop = i; break; }
    s->loop.trial_limit = uint64_t(i_stop) + 1u;
    const int world = s->mode == 1 ? 1 : s->loop.world;       // split mode: one trial per "window", all ranks on it
    const int windows = (i_stop + 1 + world - 1) / world;
    const auto t0 = std::chrono::system_clock::now();
    const long max_seconds = long(s4p_matcher_max_time_seconds(m));
    // in slices, so that a crossed threshold or the time budget stops the job within a few windows
    s->split.terminated = false; s->split.trials_done = 0;
    for (int done = 0; done < windows && !s->loop.terminated && !s->split.terminated;) {
      const int n = std::min(windows - done, 8 * std::max(1, s4p_pipeline_depth(s4p_matcher_ctx(m))));
      int32_t term = 0;
      if (int32_t rc = s4p_shard_run_windows(s, n, nullptr, &term)) return rc;
      done += n;
      // time budget (integer seconds / integer max_time_seconds: the reference's quirk, :240-243), agreed on by all ranks
      const long el = long(std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now() - t0).count());
      const uint64_t mine = (max_seconds > 0 && float(el / max_seconds) >= 0.99f) ? 1u : 0u;
      uint64_t any = 0;
      if (int32_t rc = s->coll->post(2, mine)) { s->err = s->coll->err; return rc; }      // a slot of its own: never a window's
      if (int32_t rc = s->coll->result(2, &any)) { s->err = s->coll->err; return rc; }
      if (any) break;
    }
    s->loop.trial_limit = ~0ull;
  }
  if (int32_t rc = s4p_matcher_get_info(m, &info)) return rc;
  *lcp = info.best_lcp;
  if (info.best_lcp > lcp0) {
    if (int32_t rc = s4p_matcher_global_transform(m, M)) return rc;
    if (qx && qy && qz) {                                           // match4pcsBase.hpp:259-268
      if (qx != Q->x) std::memcpy(qx, Q->x, size_t(Q->n) * 4);
      if (qy != Q->y) std::memcpy(qy, Q->y, size_t(Q->n) * 4);
      if (qz != Q->z) std::memcpy(qz, Q->z, size_t(Q->n) * 4);
      if (int32_t rc = s4p_transform_points(s4p_matcher_ctx(m), M, qx, qy, qz, Q->n)) { s->err = s4p_last_error(s4p_matcher_ctx(m)); return rc; }
    }
  } else {
    std::memcpy(M, info.transform, sizeof(float) * 16);
  }
  return S4P_OK;
}

// Test aid of the two replay harnesses below: S4P_TEST_FAIL_COMMIT="<rank>:<k>" makes the k-th commit (0-based) of that rank fail,
// the one local failure the recorded outcomes cannot express (tests/test_shard_native_gloo.py).
static bool replay_commit_fails(int rank, int n_commits_so_far) {
  const char* e = std::getenv("S4P_TEST_FAIL_COMMIT");
  if (!e) return false;
  int r = -1, k = -1;
  return std::sscanf(e, "%d:%d", &r, &k) == 2 && r == rank && k == n_commits_so_far;
}

// Host-only self-check of the window loop (no matcher, no GPU): this rank replays recorded outcomes of ITS trials
// (found[w], results[w] for window w) through the same Loop and the given collective, and logs every commit.
// The CPU tests run it over gloo with 2 and 4 ranks against the sequential semantics.
int32_t s4p_shard_replay(int32_t rank, int32_t world, const s4p_collective* coll, int32_t n_windows, int32_t depth,
                         uint32_t threshold_count, uint32_t start_best_count, const int32_t* found,
                         const s4p_base_result* results, int32_t* commit_trials, uint32_t* commit_counts, int32_t commit_cap,
                         int32_t* n_commits, int32_t* terminated, uint64_t* trials_done) {
  if (!coll || !coll->allreduce_max_u64 || !coll->broadcast || world < 1 || rank < 0 || rank >= world || !found || !results || !n_commits)
    return S4P_ERR_BAD_ARG;
  CallbackCollective cc(*coll);
  Loop L;
  L.rank = rank; L.world = world; L.coll = &cc; L.threshold_count = threshold_count; L.best_count = start_best_count;
  L.ops.depth = std::max(1, depth);
  int prepared = 0, waited = 0, trial = 0;
  std::deque<int> own;                                               // windows whose own trial is "in flight"
  *n_commits = 0;
  L.ops.prepare = [&](bool run_device, bool* f, int32_t ids[4]) -> int32_t {
    const int t = trial++;                                           // global trial index = window * world + position
    ids[0] = t; ids[1] = ids[2] = ids[3] = 0;
    if (run_device) { *f = found[prepared] != 0; if (*f) own.push_back(prepared); ++prepared; }
    else *f = true;                                                  // what other ranks found does not enter the key
    return S4P_OK;
  };
  L.ops.wait_own = [&](s4p_base_result* r) -> int32_t {
    if (own.empty()) return S4P_ERR_STATE;
    *r = results[own.front()];
    own.pop_front(); ++waited;
    if (r->n_quads == ~0ull) { L.err = "injected failure of this rank's device pass"; return S4P_ERR_CAPACITY; }   // (tests: the collective abort)
    return S4P_OK;
  };
  L.ops.commit = [&](const int32_t ids[4], const s4p_base_result* r, bool* ok) -> int32_t {
    if (replay_commit_fails(rank, *n_commits)) { L.err = "injected failure of this rank's commit"; return S4P_ERR_CAPACITY; }
    if (*n_commits < commit_cap) { if (commit_trials) commit_trials[*n_commits] = ids[0]; if (commit_counts) commit_counts[*n_commits] = r->best_count; }
    ++*n_commits;
    *ok = r->best_count > threshold_count;
    return S4P_OK;
  };
  const int32_t rc = L.run(n_windows);
  if (terminated) *terminated = L.terminated ? 1 : 0;
  if (trials_done) *trials_done = L.trials_done;
  return rc;
}

// Host-only self-check of the split-base loop (no matcher, no GPU): this rank replays the recorded results of ITS share of
// every trial (found[t], results[t]) through the same SplitLoop and the given collective, and logs every commit.
int32_t s4p_shard_replay_split(int32_t rank, int32_t world, const s4p_collective* coll, int32_t n_trials, int32_t depth,
                               uint32_t threshold_count, uint32_t start_best_count, const int32_t* found,
                               const s4p_base_result* results, int32_t* commit_trials, uint32_t* commit_counts, uint64_t* commit_tags,
                               int32_t commit_cap, int32_t* n_commits, int32_t* terminated, uint64_t* trials_done) {
  if (!coll || !coll->allreduce_max_u64 || !coll->broadcast || world < 1 || rank < 0 || rank >= world || !found || !results || !n_commits)
    return S4P_ERR_BAD_ARG;
  CallbackCollective cc(*coll);
  SplitLoop L;
  L.rank = rank; L.world = world; L.coll = &cc; L.best_count = start_best_count; L.threshold_count = threshold_count;
  L.ops.depth = std::max(1, depth);
  int prepared = 0;
  std::deque<int> own;
  *n_commits = 0;
  L.ops.prepare = [&](bool* f, int32_t ids[4]) -> int32_t {
    ids[0] = prepared; ids[1] = ids[2] = ids[3] = 0;
    *f = found[prepared] != 0;
    if (*f) own.push_back(prepared);
    ++prepared;
    return S4P_OK;
  };
  L.ops.wait_own = [&](s4p_base_result* r) -> int32_t {
    if (own.empty()) return S4P_ERR_STATE;
    *r = results[own.front()];
    own.pop_front();
    if (r->n_quads == ~0ull) { L.err = "injected failure of this rank's device pass"; return S4P_ERR_CAPACITY; }
    return S4P_OK;
  };
  L.ops.commit = [&](const int32_t ids[4], const s4p_base_result* r, bool* ok) -> int32_t {
    if (replay_commit_fails(rank, *n_commits)) { L.err = "injected failure of this rank's commit"; return S4P_ERR_CAPACITY; }
    if (*n_commits < commit_cap) {
      if (commit_trials) commit_trials[*n_commits] = ids[0];
      if (commit_counts) commit_counts[*n_commits] = r->best_count;
      if (commit_tags) commit_tags[*n_commits] = r->best_rank;
    }
    ++*n_commits;
    *ok = r->best_count > threshold_count;
    return S4P_OK;
  };
  const int32_t rc = L.run(n_trials);
  if (terminated) *terminated = L.terminated ? 1 : 0;
  if (trials_done) *trials_done = L.trials_done;
  return rc;
}

}  // extern "C"
