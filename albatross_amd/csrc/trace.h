// trace.h — roctx ranges around the stages of the hot path (Gram / factorisation / substitutions / predict), so that
// `rocprofv3 --marker-trace` timelines show the stages next to the kernels.  The reference has no tracing of its own
// (SURVEY.md section 5); this is the ROCm-native equivalent of what an operator would add.
//
// The roctx library is bound at run time and only when AGP_ROCTX=1: without it a range is two predictable branches.
#pragma once
#include <dlfcn.h>
#include <cstdlib>

namespace agp {

struct RoctxApi {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};

inline const RoctxApi &roctx_api() {
  static RoctxApi api = [] {
    RoctxApi a;
    const char *e = getenv("AGP_ROCTX");
    if (!e || e[0] != '1') return a;
    // rocprofv3 (--marker-trace) listens to the rocprofiler-sdk roctx library; libroctx64 is the roctracer one
    for (const char *nm : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      if (void *h = dlopen(nm, RTLD_NOW | RTLD_LOCAL)) {
        a.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (a.push && a.pop) break;
        a.push = nullptr; a.pop = nullptr;
      }
    }
    return a;
  }();
  return api;
}

struct TraceRange {
  bool on;
  explicit TraceRange(const char *name) : on(roctx_api().push != nullptr) {
    if (on) (void)roctx_api().push(name);
  }
  ~TraceRange() {
    if (on) (void)roctx_api().pop();
  }
  TraceRange(const TraceRange &) = delete;
  TraceRange &operator=(const TraceRange &) = delete;
};

}  // namespace agp
