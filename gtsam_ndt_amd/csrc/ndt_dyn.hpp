// Optional run-time dependencies of libndt_hip.so, loaded on first use with dlopen instead of being link-time
// NEEDED entries: a single-GPU caller (or the C adapter) must be able to load the matcher on a machine that has
// neither RCCL nor the roctx marker library, and a process that already carries a copy of RCCL (PyTorch bundles one
// under the same soname) must not get a second one mapped beside it - dlopen by soname returns the copy that is loaded.
//   RCCL   only the multi-device gather (ndt2d_multi_align_dev / ndt3d_multi_align_dev) calls it;
//          without it those two entry points return NDT_ERR_RCCL, everything else works.
//   roctx  marker ranges around API calls (rocprofv3 --marker-trace); without it the ranges are no-ops.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and enums only: no RCCL symbol is referenced at link time

namespace ndt {

inline void* dlopen_first(const char* const* names) {
  for (; *names; ++names)
    if (void* h = dlopen(*names, RTLD_NOW | RTLD_GLOBAL)) return h;
  return nullptr;
}

struct RcclApi {
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

inline const RcclApi& rccl() {
  static const RcclApi api = [] {
    RcclApi a;
    static const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
    void* h = dlopen_first(names);
    if (!h) return a;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.AllGather && a.GetErrorString;
    return a;
  }();
  return api;
}

struct RoctxApi {
  int (*RangePushA)(const char*) = nullptr;
  int (*RangePop)() = nullptr;
};

inline const RoctxApi& roctx() {
  static const RoctxApi api = [] {
    RoctxApi a;
    static const char* const names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                                        "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", nullptr};
    if (void* h = dlopen_first(names)) {
      a.RangePushA = reinterpret_cast<decltype(a.RangePushA)>(dlsym(h, "roctxRangePushA"));
      a.RangePop = reinterpret_cast<decltype(a.RangePop)>(dlsym(h, "roctxRangePop"));
      if (!a.RangePushA || !a.RangePop) a.RangePushA = nullptr, a.RangePop = nullptr;
    }
    return a;
  }();
  return api;
}

}  // namespace ndt
