// Host-side helpers shared by the C-ABI translation units: per-thread error text and the
// HIP_TRY macro that turns a hipError_t into NDT_ERR_HIP without throwing.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace ndt {

inline std::string& last_error() {
  thread_local std::string e;
  return e;
}
inline void set_error(const char* msg) { last_error() = msg ? msg : ""; }

}  // namespace ndt

namespace ndt {

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
