// az_host.h -- host-side helpers shared by the translation units of libaz_mcts.so:
// the thread-local error channel, argument checks and the (AMAX, KMAX) dispatch.
#ifndef AZ_HOST_H_
#define AZ_HOST_H_

#include <hip/hip_runtime.h>

#include "az_device.h"

int az_fail(int code, const char* fmt, ...);
int az_check_desc(const az_search_desc* d, const az_tree* t);
int az_check_inputs(const az_search_desc* d, const az_search_inputs* in);
bool az_coop_applicable(const az_search_desc* d, size_t extra_lds_bytes, bool borrowed);

#define AZ_HIP(call)                                                                 \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess)                                                            \
      return az_fail(AZ_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
  } while (0)

#define AZ_LAUNCH_CHECK()                                                            \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess)                                                            \
      return az_fail(AZ_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e_)); \
  } while (0)

// Compile-time bounds (AMAX, KMAX) for a run-time (A, K).  Bounds up to 8 keep the
// per-action / per-atom vectors in registers (fully unrolled loops); the largest
// bound is a generic path with run-time loops.
template <typename F>
static int az_dispatch_ak(int A, int K, F&& f) {
#ifdef AZ_NARROW_BUILD
  // diagnostic builds only: one instantiation (A <= AZ_NARROW_A, K == 1) to keep compile times short
#ifndef AZ_NARROW_A
#define AZ_NARROW_A 2
#endif
#ifndef AZ_NARROW_K
#define AZ_NARROW_K 1
#endif
  if (A <= AZ_NARROW_A && K <= AZ_NARROW_K) return f.template operator()<AZ_NARROW_A, AZ_NARROW_K>();
  return az_fail(AZ_ERR_UNSUPPORTED, "narrow diagnostic build: A <= AZ_NARROW_A and K <= AZ_NARROW_K only");
#else
#define AZ_K_CASES(AM)                                  \
  if (K == 1) return f.template operator()<AM, 1>();    \
  if (K <= 8) return f.template operator()<AM, 8>();    \
  return f.template operator()<AM, AZ_MAX_ATOMS>();
  if (A <= 2) { AZ_K_CASES(2) }
  if (A <= 3) { AZ_K_CASES(3) }
  if (A <= 4) { AZ_K_CASES(4) }
  if (A <= 8) { AZ_K_CASES(8) }
  AZ_K_CASES(AZ_MAX_ACTIONS)
#undef AZ_K_CASES
#endif
}

// Instantiation of the cooperative kernels for a run-time (A, K): the shapes of the built-in
// environments and value heads run with A and K as compile-time constants (EXACT: every
// per-action / per-atom loop is straight-line code), everything else up to 8 x 8 with run-time
// bounds.  Calls f.template operator()<AMAX, KMAX, EXACT>().
template <typename F>
static int az_dispatch_coop(int A, int K, F&& f) {
#ifdef AZ_NARROW_BUILD
  if (A == AZ_NARROW_A && K == AZ_NARROW_K) return f.template operator()<AZ_NARROW_A, AZ_NARROW_K, true>();
  return az_fail(AZ_ERR_UNSUPPORTED, "narrow diagnostic build: A == AZ_NARROW_A and K == AZ_NARROW_K only");
#else
  if (K == 1) {
    if (A == 2) return f.template operator()<2, 1, true>();
    if (A == 3) return f.template operator()<3, 1, true>();
    if (A == 4) return f.template operator()<4, 1, true>();
  } else if (K == 8) {
    if (A == 2) return f.template operator()<2, 8, true>();
    if (A == 3) return f.template operator()<3, 8, true>();
  } else if (K == 16) {  // 16-atom quantile heads on the built-in environments
    if (A == 2) return f.template operator()<2, 16, true>();
    if (A == 3) return f.template operator()<3, 16, true>();
  }
  if (K > 8) return az_fail(AZ_ERR_UNSUPPORTED, "cooperative search dispatched beyond its bounds");
  return f.template operator()<8, 8, false>();
#endif
}

// wave-level sum of a per-thread counter, one atomic per wave
AZ_DEV void az_add_depth(unsigned long long* dst, long long v) {
  if (!dst) return;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(dst, (unsigned long long)v);
}

#endif  // AZ_HOST_H_
