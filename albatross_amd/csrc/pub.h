// pub.h - values handed from one workgroup to another INSIDE a launch (the fused panel / step kernels of chol.hip, the
// one-launch back substitution of solve.hip): sentinel-filled slots, device-scope stores and loads, bounded polling.
#pragma once
#include <hip/hip_runtime.h>

namespace agp {

// ---- hand-over of the factored diagonal block INSIDE one launch (panel_fused_kernel) ----------------------------
// The workgroup that factors the 128 x 128 diagonal block emits its tile image tile by tile while it runs; the
// workgroups that solve the rows below consume those tiles as they appear.  No flags: the image (and the slot of
// z_b) is filled with a sentinel bit pattern before the launch, the producer writes every value ONCE with a
// device-scope store (global_store ... sc1: written through to the memory side, past the XCD-private L2), and a
// consumer re-reads a fragment with device-scope loads (sc1) until none of its values is the sentinel.  8-byte
// accesses are single-copy atomic, so a value is either the sentinel or final.  The producer never waits for an
// acknowledgement - nothing is added to the serial pivot chain - and a consumer waits for exactly the values it is
// about to multiply.  The sentinel is a NaN payload no arithmetic produces (hardware NaNs are 0x7FF8000000000000).
constexpr unsigned long long PUB_SENTINEL = 0xFFF8A5A5DEADBEEFull;

__device__ __forceinline__ void store_pub(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_pub(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool is_unpublished(double v) {
  return (unsigned long long)__double_as_longlong(v) == PUB_SENTINEL;
}

// A consumer never spins for ever: after ~2 s (s_memrealtime, 100 MHz) without the values it waits for it records the
// failure in flags[2] and carries on with whatever it has read - the host turns that flag into AGP_ERR_HIP instead of
// the launch hanging the GPU (that can only happen if the producer workgroup died).
constexpr unsigned long long PUB_TIMEOUT_TICKS = 200000000ull;

__device__ __forceinline__ bool poll_expired(unsigned long long t0, int *flags) {
  if (__builtin_amdgcn_s_memrealtime() - t0 < PUB_TIMEOUT_TICKS) return false;
  if (flags && (threadIdx.x & 63) == 0) atomicExch(flags + 2, 1);
  return true;
}



// Several small fills / copies of 8-byte words in ONE launch (what precedes a fit: sentinel-filled hand-over buffers,
// zeroed flags and counters, the copies of the training features and targets).  A launch is ~4-5 us of stream time
// whatever it moves; a fit of a few hundred points made ten of them.
constexpr int PREP_MAX = 14;
struct PrepArgs {
  unsigned long long *dst[PREP_MAX];
  const unsigned long long *src[PREP_MAX];  // nullptr: fill with pattern
  unsigned long long pattern[PREP_MAX];
  long long first_block[PREP_MAX + 1];      // blocks of 1024 words
  long long count[PREP_MAX];
  int n = 0;
  void add(void *d, const void *sr, unsigned long long pat, long long words) {
    if (words <= 0 || n >= PREP_MAX) return;
    if (n == 0) first_block[0] = 0;
    dst[n] = static_cast<unsigned long long *>(d);
    src[n] = static_cast<const unsigned long long *>(sr);
    pattern[n] = pat;
    count[n] = words;
    first_block[n + 1] = first_block[n] + (words + 1023) / 1024;
    ++n;
  }
  void fill(void *d, unsigned long long pat, long long words) { add(d, nullptr, pat, words); }
  void sentinel(double *d, long long words) { add(d, nullptr, PUB_SENTINEL, words); }
  void copy(void *d, const void *sr, long long words) { add(d, sr, 0, words); }
  bool full() const { return n >= PREP_MAX; }
};
void launch_prep(hipStream_t s, const PrepArgs &a);  // chol.hip

// Many small device-to-device copies of 8-byte words in ONE launch, described by a table in DEVICE memory (the training
// features of the problems of a batch: one copy kernel each was half of a batched fit's host time)
struct CopyItem {
  unsigned long long *dst;
  const unsigned long long *src;
  long long words;
};
void launch_copy_table(hipStream_t s, const CopyItem *table_dev, long long count, long long max_words);  // chol.hip

}  // namespace agp
