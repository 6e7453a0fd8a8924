// Native driver of the pipelined DYNAMIC control step (include/deer_model.h, "step plan").
//
// The step is a set of pre-captured HIP-graph pieces (captured once by the host binding from the spine's piece functions):
//   * per vision chain a short head graph (begin + patch embedding + first ViT blocks) and a tail graph (rest + Perceiver),
//     chain 0 on the caller's stream, the others on their own streams;
//   * per trunk layer of the dynamic plan one graph on the caller's stream (piece 0 also holds the media K/V projection and
//     the token embedding);
//   * per pseudo / exit-check layer one head-evaluation graph on the head stream, forked off after that layer.
// The exit decision is taken on the device; the last kernel of every exit check publishes (sequence number, checks done,
// verdict) into pinned host memory (csrc/head.hip::check_done).  This driver submits the pieces, keeps at most `lookahead`
// trunk layers in flight beyond an undecided check, polls the mirror and stops submitting at the exit - the loop that
// engine.py::_step_segmented runs in Python, as one GIL-free native call (the Python loop spent ~40 us per step before the
// first graph was submitted and ~25 us per piece; submission here is one hipGraphLaunch per piece).
#include "common.h"
#include "../../include/deer_hip.h"
#include "../../include/deer_model.h"
#include <chrono>
#include <sched.h>
#include <cstring>
#include <vector>

struct deer_step_plan {
  int n_chains = 0, n_pieces = 0, lookahead = 1, n_envs = 1;
  std::vector<hipGraphExec_t> chain_head, chain_tail, main_g, head_g;   // head_g[i] == nullptr: no head evaluation after piece i
  std::vector<int> exits;                                             // piece index of exit check k
  hipEvent_t ev_in = nullptr;
  std::vector<hipEvent_t> ev_join, ev_head, ev_head_done;
  int* step_info = nullptr;                                           // pinned: {hold, seq, mirror ptr lo, hi} read by ctl_begin_step
  int* mirror = nullptr;                                              // pinned: [0] progress, [1] done, then one ctl block per env
};

namespace {
inline int load_acquire(const int* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
#define SD_HIP(expr)                               \
  do {                                             \
    if ((expr) != hipSuccess) return DEER_ERR_LAUNCH; \
  } while (0)
}  // namespace

extern "C" int deer_step_plan_create(int n_chains, void* const* chain_head, void* const* chain_tail, int n_pieces, void* const* main_graphs,
                                     void* const* head_graphs, const int* is_exit, int lookahead, int n_envs, int* step_info_pinned,
                                     int* host_mirror, deer_step_plan** out) {
  if (out == nullptr || n_chains <= 0 || n_chains > 8 || n_pieces <= 0 || chain_head == nullptr || chain_tail == nullptr ||
      main_graphs == nullptr || head_graphs == nullptr || is_exit == nullptr || lookahead < 0 || n_envs <= 0 || step_info_pinned == nullptr ||
      host_mirror == nullptr)
    return DEER_ERR_SHAPE;
  deer_step_plan* p = new deer_step_plan();
  p->n_chains = n_chains; p->n_pieces = n_pieces; p->lookahead = lookahead; p->n_envs = n_envs;
  p->step_info = step_info_pinned; p->mirror = host_mirror;
  for (int c = 0; c < n_chains; ++c) {
    if (chain_head[c] == nullptr || chain_tail[c] == nullptr) { delete p; return DEER_ERR_SHAPE; }
    p->chain_head.push_back(reinterpret_cast<hipGraphExec_t>(chain_head[c]));
    p->chain_tail.push_back(reinterpret_cast<hipGraphExec_t>(chain_tail[c]));
  }
  for (int i = 0; i < n_pieces; ++i) {
    if (main_graphs[i] == nullptr || (is_exit[i] && head_graphs[i] == nullptr)) { delete p; return DEER_ERR_SHAPE; }
    p->main_g.push_back(reinterpret_cast<hipGraphExec_t>(main_graphs[i]));
    p->head_g.push_back(reinterpret_cast<hipGraphExec_t>(head_graphs[i]));
    if (is_exit[i]) p->exits.push_back(i);
  }
  if (p->exits.empty()) { delete p; return DEER_ERR_SHAPE; }          // a dynamic step without an exit check never ends
  bool ok = hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming) == hipSuccess;
  p->ev_join.resize(n_chains, nullptr);
  p->ev_head.resize(n_pieces, nullptr);
  p->ev_head_done.resize(n_pieces, nullptr);
  for (int c = 1; c < n_chains && ok; ++c) ok = hipEventCreateWithFlags(&p->ev_join[c], hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < n_pieces && ok; ++i)
    if (p->head_g[i] != nullptr)
      ok = hipEventCreateWithFlags(&p->ev_head[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&p->ev_head_done[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { deer_step_plan_destroy(p); return DEER_ERR_LAUNCH; }
  *out = p;
  return DEER_OK;
}

extern "C" void deer_step_plan_destroy(deer_step_plan* p) {
  if (p == nullptr) return;
  if (p->ev_in) (void)hipEventDestroy(p->ev_in);
  for (hipEvent_t e : p->ev_join) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_head) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_head_done) if (e) (void)hipEventDestroy(e);
  delete p;
}

// One dynamic control step.  hold: 1 iff cur_step % steps_per_stage != 0 (value_net.py:285-286); seq: the step's sequence number
// (1 .. 2^24-1, increasing; the caller clears the mirror when it wraps).  chain_streams: streams of chains 1 .. n_chains-1.
// ctl_out: host int32 [n_envs * 64] <- the environments' control blocks as published with their verdicts.  pieces_out (NULL ok):
// number of trunk pieces submitted.  Returns DEER_OK, DEER_ERR_LAUNCH, or 3 when no verdict arrived within 20 s.
extern "C" int deer_step_plan_run(deer_step_plan* p, int hold, int seq, void* main_stream, void* const* chain_streams, void* head_stream,
                                  int* ctl_out, int* pieces_out) {
  if (p == nullptr || seq <= 0 || ctl_out == nullptr || (p->n_chains > 1 && chain_streams == nullptr)) return DEER_ERR_SHAPE;
  hipStream_t ms = reinterpret_cast<hipStream_t>(main_stream);
  hipStream_t hs = head_stream != nullptr ? reinterpret_cast<hipStream_t>(head_stream) : ms;
  // the device reads the step info when ctl_begin_step RUNS; the previous step's verdict has been seen, so that kernel is long gone
  p->step_info[0] = hold;
  p->step_info[1] = seq;
  __atomic_thread_fence(__ATOMIC_RELEASE);

  // ---- vision tower: chain 0 here, the others on their streams once the inputs are in place; join before the trunk ----
  auto cs = [&](int c) {                                              // a NULL chain stream = the caller's stream
    return (c == 0 || chain_streams[c - 1] == nullptr) ? ms : reinterpret_cast<hipStream_t>(chain_streams[c - 1]);
  };
  bool forked = false;
  for (int c = 1; c < p->n_chains; ++c) forked = forked || cs(c) != ms;
  if (forked) {
    SD_HIP(hipEventRecord(p->ev_in, ms));
    for (int c = 1; c < p->n_chains; ++c)
      if (cs(c) != ms) SD_HIP(hipStreamWaitEvent(cs(c), p->ev_in, 0));
  }
  for (int part = 0; part < 2; ++part)
    for (int c = 0; c < p->n_chains; ++c) SD_HIP(hipGraphLaunch(part == 0 ? p->chain_head[c] : p->chain_tail[c], cs(c)));
  for (int c = 1; c < p->n_chains; ++c)
    if (cs(c) != ms) {
      SD_HIP(hipEventRecord(p->ev_join[c], cs(c)));
      SD_HIP(hipStreamWaitEvent(ms, p->ev_join[c], 0));
    }

  // ---- trunk pieces with head evaluations forked off; stop at the exit ----
  const int* mirror = p->mirror;
  auto poll = [&](int n_checks) -> int {                             // 1: every environment exited, 0: n_checks decided, -1: timeout
    const int want = seq * 64 + n_checks;
    unsigned spins = 0;
    std::chrono::steady_clock::time_point dead;
    bool armed = false;
    while (load_acquire(mirror + HOSTM_DONE) != seq && load_acquire(mirror + HOSTM_PROGRESS) < want) {
      // one poller per engine (four per GPU in the batched_groups leg, 32 on an 8-GPU node): `pause` keeps the spin off the
      // sibling hyperthread's issue slots and the memory pipeline; a wait longer than ~100 us (the vision tower: ~2 ms before the
      // first verdict) also yields the core to a runnable launch thread (sched_yield returns at once when there is none)
      __builtin_ia32_pause();
      if (spins > 4096u) sched_yield();
      if ((++spins & 0xFFFFFu) == 0) {
        const auto now = std::chrono::steady_clock::now();
        if (!armed) { dead = now + std::chrono::seconds(20); armed = true; }
        else if (now > dead) return -1;
      }
    }
    return load_acquire(mirror + HOSTM_DONE) == seq ? 1 : 0;
  };
  const int n_exits = (int)p->exits.size();
  int decided = 0, done = 0, launched = 0;
  for (int i = 0; i < p->n_pieces && !done; ++i) {
    while (decided < n_exits && p->exits[decided] + p->lookahead < i) {
      done = poll(decided + 1);
      if (done < 0) return 3;
      ++decided;
      if (done) break;
    }
    if (done) break;
    // env batches with compaction (csrc/model.hip): piece i gathers the rows of the environments still active after the exit check of
    // layer i - 2.  With the default look-ahead the host has already SEEN that verdict (poll above); the stream dependency makes it
    // hold for every look-ahead
    if (p->n_envs > 1 && hs != ms && i >= 2 && p->ev_head_done[i - 2] != nullptr) SD_HIP(hipStreamWaitEvent(ms, p->ev_head_done[i - 2], 0));
    SD_HIP(hipGraphLaunch(p->main_g[i], ms));
    ++launched;
    if (p->head_g[i] != nullptr) {
      if (hs != ms) {
        SD_HIP(hipEventRecord(p->ev_head[i], ms));
        SD_HIP(hipStreamWaitEvent(hs, p->ev_head[i], 0));
      }
      SD_HIP(hipGraphLaunch(p->head_g[i], hs));
      if (p->n_envs > 1 && hs != ms) SD_HIP(hipEventRecord(p->ev_head_done[i], hs));
    }
  }
  if (!done) {
    done = poll(n_exits);
    if (done < 0) return 3;
  }
  if (pieces_out != nullptr) *pieces_out = launched;
  if (!done) return DEER_ERR_LAUNCH;                                  // the forced exit at the last check always fires
  std::memcpy(ctl_out, const_cast<const int*>(mirror) + CTL_WORDS, sizeof(int) * CTL_WORDS * p->n_envs);
  return DEER_OK;
}
