// kloop_probe.hip -- how fast does the networks' k-loop (az_mlp_unit_loop) issue matrix
// instructions when a wave has a SIMD to itself, and with a partner (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I alphazero_amd/csrc -I include \
//         -o tools/probes/kloop_probe.bin tools/probes/kloop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "az_mlp.h"

// VARIANT 0: the library's loop; 1: a lean loop (16 k-blocks, no clamps, one running pointer)
template <int VARIANT>
__global__ __launch_bounds__(512) void k_loop(const float* w, int units, int rider, float* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  for (int i = tid; i < 16 * 260; i += blockDim.x) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  MlpUnit u{};
  u.nkb = 16;
  u.nkb2 = rider;
  u.phase = 1;
  MlpPre next;
  const az_f32x4* wb = reinterpret_cast<const az_f32x4*>(w) + (size_t)(blockIdx.x % 4) * 8 * 2 * 16 * 64;
  next.w0 = wb + (size_t)wave * 2 * 16 * 64;
  next.w1 = next.w0 + 16 * 64;
  next.w2 = wb;
  next.nkb = 16;
  next.nkb2 = rider;
  MlpRegs R;
#pragma unroll
  for (int kb = 0; kb < AZ_SLOTS; ++kb) {
    R.W0[kb] = az_ldg4(next.w0 + kb * 64 + lane);
    R.W1[kb] = az_ldg4(next.w1 + kb * 64 + lane);
    if (kb < AZ_RIDER_SLOTS) R.W2[kb] = az_ldg4(next.w2 + kb * 64 + lane);
  }
  az_f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  const float* ap = lds + (lane & 15) * 260 + (lane >> 4) * 64;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < units; ++it) {
    if (VARIANT == 0) {
      az_mlp_unit_loop<false>(u, next, R, acc0, acc1, acc2, ap, ap, lane);
    } else {
      az_f32x4 a[2];
      a[0] = *reinterpret_cast<const az_f32x4*>(ap);
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        if (kb + 1 < 16) a[(kb + 1) & 1] = *reinterpret_cast<const az_f32x4*>(ap + (kb + 1) * 4);
        const az_f32x4 a4 = a[kb & 1], b0 = R.W0[kb], b1 = R.W1[kb];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        R.W0[kb] = az_ldg4_lane(next.w0 + kb * 64, lane);
        R.W1[kb] = az_ldg4_lane(next.w1 + kb * 64, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = acc0.x + acc1.y + acc2.z;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;
  const int units = 200;
  float *w, *out;
  long long* cyc;
  const size_t wfloats = (size_t)4 * 8 * 2 * 16 * 64 * 4 + 4096;
  (void)hipMalloc(&w, wfloats * 4);
  (void)hipMemset(w, 0, wfloats * 4);
  (void)hipMalloc(&out, (size_t)wgs * 512 * 4);
  (void)hipMalloc(&cyc, 64);
  const size_t lds = (16 * 260 + 16) * 4;
  auto run = [&](auto kern, const char* name, int waves, int rider) {
    for (int rep = 0; rep < 2; ++rep) kern<<<wgs, waves * 64, lds>>>(w, units, rider, out, cyc);
    long long h[8];
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-8s waves=%d rider=%d: wave 0 %.0f cycles per unit (128 MFMAs: floor %d), per SIMD %.1f cycles per MFMA\n", name, waves,
           rider, (double)h[0] / units, waves > 4 ? 8192 : 4096, (double)h[0] / units / (128.0 * (waves > 4 ? 2 : 1)));
  };
  run(k_loop<0>, "library", 4, 0);
  run(k_loop<0>, "library", 4, 2);
  run(k_loop<0>, "library", 8, 0);
  run(k_loop<0>, "library", 8, 2);
  run(k_loop<1>, "lean", 4, 0);
  run(k_loop<1>, "lean", 8, 0);
  return 0;
}
