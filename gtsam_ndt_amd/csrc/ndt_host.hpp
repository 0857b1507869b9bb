// Host-side helpers shared by the C-ABI translation units: per-thread error text, the HIP_TRY
// macro that turns a hipError_t into NDT_ERR_HIP without throwing, launch-chain graphs and their
// cache, and the host half of the converged-mode protocol (flags in pinned host memory).
#pragma once
#include <hip/hip_runtime.h>

#include <sched.h>

#include <atomic>
#include <string>

#include "ndt_dyn.hpp"

namespace ndt {

inline std::string& last_error() {
  thread_local std::string e;
  return e;
}
inline void set_error(const char* msg) { last_error() = msg ? msg : ""; }

}  // namespace ndt

namespace ndt {

// the sharded build counters (ndt_device.hpp: kCountShards pairs of valid / overflowed cells) read back -> their totals
inline void sum_count_shards(const int* hc, int* valid, int* over) {
  int v = 0, o = 0;
  for (int k = 0; k < 16; ++k) { v += hc[2 * k]; o += hc[2 * k + 1]; }
  *valid = v; *over = o;
}

// roctx range over an API call (SURVEY.md section 5 "tracing"): shows up in rocprofv3 --marker-trace around
// the kernels the call enqueues; a few nanoseconds when no profiler is attached, nothing at all when the marker
// library is not on the machine (ndt_dyn.hpp).
struct TraceRange {
  explicit TraceRange(const char* name) { if (roctx().RangePushA) (void)roctx().RangePushA(name); }
  ~TraceRange() { if (roctx().RangePop) (void)roctx().RangePop(); }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

// Orders everything enqueued on `consumer` from now on behind the work that is in `producer` now
// (event record + stream wait; the host does not block).
// `cached` (optional): an event the caller keeps for this purpose, created on first use - a wait that
// is already enqueued refers to the record that preceded it, so re-recording the event is safe.
inline hipError_t order_after(hipStream_t consumer, hipStream_t producer, hipEvent_t* cached = nullptr) {
  if (producer == consumer) return hipSuccess;
  hipEvent_t ev = cached ? *cached : nullptr;
  hipError_t e = hipSuccess;
  if (!ev) {
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    if (cached) *cached = ev;
  }
  e = hipEventRecord(ev, producer);
  if (e == hipSuccess) e = hipStreamWaitEvent(consumer, ev, 0);
  if (!cached) (void)hipEventDestroy(ev);            // released by the runtime once the wait has been satisfied
  return e;
}

// A linear chain of `launches` launches of one kernel whose last argument is the launch parity
// (k & 1), built with explicit graph nodes.  No stream capture: capture state is process-wide
// in the HIP runtime and this library's handles may be driven from several threads at once.
inline hipError_t build_chain_graph(const void* func, dim3 grid, dim3 block, void* a0, void* a1, void* a2,
                                    int launches, hipGraph_t* graph_out, hipGraphExec_t* exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipGraphCreate(&g, 0);
  if (e != hipSuccess) return e;
  hipGraphNode_t prev = nullptr;
  for (int k = 0; k < launches && e == hipSuccess; ++k) {
    int parity = k & 1;
    void* args[4] = {&a0, &a1, &a2, &parity};
    hipKernelNodeParams p{};
    p.func = const_cast<void*>(func);
    p.gridDim = grid;
    p.blockDim = block;
    p.sharedMemBytes = 0;
    p.kernelParams = args;
    p.extra = nullptr;
    hipGraphNode_t node = nullptr;
    e = hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &p);
    prev = node;
  }
  if (e == hipSuccess) e = hipGraphInstantiate(exec_out, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); return e; }
  *graph_out = g;
  return hipSuccess;
}

// The same for chains that alternate two kernels per step (A then B, both taking the step's parity).
inline hipError_t build_chain_graph2(const void* fa, dim3 ga, dim3 ba, const void* fb, dim3 gb, dim3 bb, void* a0, void* a1,
                                     void* a2, int steps, hipGraph_t* graph_out, hipGraphExec_t* exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipGraphCreate(&g, 0);
  if (e != hipSuccess) return e;
  hipGraphNode_t prev = nullptr;
  for (int k = 0; k < 2 * steps && e == hipSuccess; ++k) {
    int parity = (k >> 1) & 1;
    void* args[4] = {&a0, &a1, &a2, &parity};
    hipKernelNodeParams p{};
    p.func = const_cast<void*>((k & 1) ? fb : fa);
    p.gridDim = (k & 1) ? gb : ga;
    p.blockDim = (k & 1) ? bb : ba;
    p.sharedMemBytes = 0;
    p.kernelParams = args;
    p.extra = nullptr;
    hipGraphNode_t node = nullptr;
    e = hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &p);
    prev = node;
  }
  if (e == hipSuccess) e = hipGraphInstantiate(exec_out, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); return e; }
  *graph_out = g;
  return hipSuccess;
}

// A handle alternates between a few chain lengths (the converged-mode chunk, the fixed-K chain of a
// timing run, the 2-launch chain of an evaluation): keep the last few instantiated graphs instead
// of rebuilding one on every change.
struct ChainGraphCache {
  static constexpr int kSlots = 16;
  struct Slot { int launches = 0, blocks = 0, mode = -1; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; unsigned long stamp = 0; };
  Slot slot[kSlots];
  unsigned long clock = 0;

  hipGraphExec_t find(int launches, int blocks, int mode) {
    for (Slot& s : slot)
      if (s.exec && s.launches == launches && s.blocks == blocks && s.mode == mode) { s.stamp = ++clock; return s.exec; }
    return nullptr;
  }
  // the slot to build into: an empty one, else the least recently used (destroyed first).  Replays
  // of the evicted exec may still be queued (a converged-mode loop leaves up to two chunks of no-op
  // launches behind): the owner's stream is drained before the exec is destroyed.
  Slot* victim(hipStream_t stream) {
    Slot* v = &slot[0];
    for (Slot& s : slot) {
      if (!s.exec) { v = &s; break; }
      if (s.stamp < v->stamp) v = &s;
    }
    if (v->exec) (void)hipStreamSynchronize(stream);
    release(*v);
    return v;
  }
  void release(Slot& s) {
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
    if (s.graph) (void)hipGraphDestroy(s.graph);
    s = Slot{};
  }
  void clear() { for (Slot& s : slot) release(s); }
  hipError_t get(const void* func, dim3 grid, dim3 block, void* a0, void* a1, void* a2, int launches, int mode,
                 hipStream_t stream, hipGraphExec_t* out) {
    if (hipGraphExec_t e = find(launches, (int)grid.x, mode)) { *out = e; return hipSuccess; }
    Slot* s = victim(stream);
    const hipError_t err = build_chain_graph(func, grid, block, a0, a1, a2, launches, &s->graph, &s->exec);
    if (err != hipSuccess) { *s = Slot{}; return err; }
    s->launches = launches; s->blocks = (int)grid.x; s->mode = mode; s->stamp = ++clock;
    *out = s->exec;
    return hipSuccess;
  }
  // two-kernel chains (build_chain_graph2); `blocks` of the key is the first kernel's grid
  hipError_t get2(const void* fa, dim3 ga, dim3 ba, const void* fb, dim3 gb, dim3 bb, void* a0, void* a1, void* a2, int steps,
                  int mode, hipStream_t stream, hipGraphExec_t* out) {
    if (hipGraphExec_t e = find(steps, (int)ga.x, mode)) { *out = e; return hipSuccess; }
    Slot* s = victim(stream);
    const hipError_t err = build_chain_graph2(fa, ga, ba, fb, gb, bb, a0, a1, a2, steps, &s->graph, &s->exec);
    if (err != hipSuccess) { *s = Slot{}; return err; }
    s->launches = steps; s->blocks = (int)ga.x; s->mode = mode; s->stamp = ++clock;
    *out = s->exec;
    return hipSuccess;
  }
};

// Converged mode of the launch-chain paths: replay `exec` (an even-length chunk of launches),
// always one chunk ahead of the one being waited for, until the finishing launch raises flag[0]
// in pinned host memory (it has written its state next to it first).  Nothing but kernel
// launches goes into the stream: no copy, no event, no sync inside the loop; launches enqueued
// past the end exit on the `done` state (about 1.7 us each).
//   flag[0]  raised by the finishing launch
//   flag[1]  index of the last launch that ran its prologue (the host's view of progress)
//   flag[2]  call number, written by the first launch past the end: the source arrays are free
struct ChunkRun {          // a converged-mode loop in flight (begin ... finish)
  hipGraphExec_t exec = nullptr;
  int chunk = 0, max_launches = 0, launched = 0;
  bool active = false;
  int seq = 0;             // number of this call: flag[2] == seq means "nothing reads the sources any more"
  bool drain = true;       // before returning, wait until nothing reads the source arrays any more (the
                           // first launch past the end says so); a caller that owns those arrays may skip it
};

// Enqueue the first two chunks and return: the asynchronous half.
inline hipError_t chunk_run_begin(ChunkRun& r, hipGraphExec_t exec, hipStream_t stream, int chunk, int max_launches) {
  r.exec = exec; r.chunk = chunk; r.max_launches = max_launches; r.launched = 0; r.active = true;
  hipError_t e = hipGraphLaunch(exec, stream);
  ++r.launched;
  if (e == hipSuccess) { e = hipGraphLaunch(exec, stream); ++r.launched; }
  return e;
}

// Spin on pinned host memory until `ready()`; the stream is only consulted for errors and, after a
// long time without progress, for a real synchronisation.  (hipStreamQuery is NOT used to decide
// that the stream has drained: with graph replays in flight it was seen to report hipSuccess in
// the middle of an alignment, which made an earlier version of this loop return a state a few
// chunks short of convergence - about one alignment in a hundred with eight busy host threads,
// tools/soak_threads.py.)  Returns hipSuccess with *ok = ready(), or the stream's error.
// How a host thread waits for a flag in pinned memory (ndt_set_host_wait): 0 = spin on the core (default: an
// alignment takes 60-300 us, shorter than a sleep's granularity), 1 = spin for the first ~20 us of a wait, then give the
// core away between polls (sched_yield) - for a SLAM process that drives several handles from several threads on
// fewer cores than threads.  Process-wide; results do not depend on it.
inline std::atomic<int>& host_wait_mode() {
  static std::atomic<int> mode{0};
  return mode;
}

template <class Ready>
inline hipError_t spin_until(hipStream_t stream, Ready&& ready, bool* ok) {
  unsigned long spins = 0;
  int syncs = 0;
  const bool yielding = host_wait_mode().load(std::memory_order_relaxed) == 1;
  for (;;) {
    if (ready()) { *ok = true; return hipSuccess; }
    if (yielding && spins > 2048) sched_yield();
    else __builtin_ia32_pause();                         // a polite spin: yields the core's issue slots to its sibling thread
    if ((++spins & 0xfffff) != 0) continue;
    const hipError_t q = hipStreamQuery(stream);
    if (q != hipSuccess && q != hipErrorNotReady) { *ok = false; return q; }
    if ((spins >> 20) % 48 == 0) {                      // roughly every second of (paused) spinning: a real sync
      const hipError_t es = hipStreamSynchronize(stream);
      if (es != hipSuccess) { *ok = false; return es; }
      if (ready()) { *ok = true; return hipSuccess; }
      if (++syncs >= 2) { *ok = false; return hipSuccess; }   // everything enqueued has run and it is still not ready
    }
  }
}

// Keep one chunk ahead until the flag is raised, then wait until the source arrays are free.
// *seen = true on success; a loop that does not report its end is an error, never a result.
inline hipError_t chunk_run_finish(ChunkRun& r, hipStream_t stream, int* flag, bool* seen) {
  auto raised = [&]() { return __atomic_load_n(&flag[0], __ATOMIC_ACQUIRE) != 0; };
  int waited = 0;
  hipError_t e = hipSuccess;
  *seen = false;
  while (e == hipSuccess) {
    const int need = (waited + 1) * r.chunk - 1;         // chunk `waited` is through when progress reaches this
    bool ok = false;
    e = spin_until(stream, [&]() { return raised() || __atomic_load_n(&flag[1], __ATOMIC_ACQUIRE) >= need; }, &ok);
    if (e != hipSuccess || raised()) break;
    // ok == false: the stream is idle and the progress counter did not move - feed it anyway; the
    // launch cap below ends a loop that can never finish
    ++waited;
    if ((r.launched - 1) * r.chunk > r.max_launches + 2 * r.chunk) { e = hipErrorLaunchFailure; break; }
    e = hipGraphLaunch(r.exec, stream);
    ++r.launched;
  }
  if (e != hipSuccess) {
    (void)hipStreamSynchronize(stream);
  } else if (r.drain) {
    // flag[2]: raised by the first launch past the end - the finishing launch (whose other workgroups
    // may still have had point loads in flight when the flag went up) is complete by then, and it left
    // n = 0 behind, so no later launch touches the source arrays.
    // flag[1] now holds the finishing launch's index: if it was the last launch enqueued there is
    // nothing behind it to raise flag[2], and the stream's end is what to wait for.
    const int last = __atomic_load_n(&flag[1], __ATOMIC_ACQUIRE);
    if (r.launched * r.chunk - 1 > last) {
      bool ok = false;
      e = spin_until(stream, [&]() { return __atomic_load_n(&flag[2], __ATOMIC_ACQUIRE) == r.seq; }, &ok);
      if (e == hipSuccess && !ok) e = hipStreamSynchronize(stream);
    } else {
      e = hipStreamSynchronize(stream);
    }
  }
  *seen = e == hipSuccess && raised();
  r.active = false;
  return e;
}

inline hipError_t run_chunks_until_flag(hipGraphExec_t exec, hipStream_t stream, int* flag, int chunk, int max_launches,
                                        int seq, bool* seen) {
  ChunkRun r;
  r.seq = seq;
  const hipError_t e = chunk_run_begin(r, exec, stream, chunk, max_launches);
  if (e != hipSuccess) { *seen = false; (void)hipStreamSynchronize(stream); return e; }
  return chunk_run_finish(r, stream, flag, seen);
}

}  // namespace ndt

#define HIP_TRY(expr)                                                            \
  do {                                                                           \
    const hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) {                                                      \
      ::ndt::last_error() = std::string(#expr) + ": " + hipGetErrorString(_e);   \
      (void)hipGetLastError();                                                   \
      return NDT_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)
