// overlap.hip -- do MFMA (wave A) and VALU work (wave B) of two waves on one SIMD overlap?
// (diagnostic).  Workgroup of 8 waves: waves 0-3 run MFMA chains, waves 4-7 run fma chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef __attribute__((ext_vector_type(4))) float f4;

// mode bit0: waves 0-3 do MFMA; bit1: waves 4-7 do VALU; bit2: set priorities (MFMA 3, VALU 0)
// bit3: VALU waves use 4 independent chains
template <int mode>
__global__ __launch_bounds__(512) void k_ov(int n, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode & 1) {
      if (mode & 4) __builtin_amdgcn_s_setprio(3);
      for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, a1, 0, 0, 0);
        }
      }
    }
  } else {
    if (mode & 2) {
      if (mode & 4) __builtin_amdgcn_s_setprio(0);
      for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (mode & 8) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b);
            x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
          } else {
            x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
            x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
          }
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a0.x + a1.y + x0 + x1 + x2 + x3;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64);
  const int n = 2000;
  auto run = [&](auto tag) {
    constexpr int mode = decltype(tag)::value;
    k_ov<mode><<<1, 512>>>(n, out, cyc);
    k_ov<mode><<<1, 512>>>(n, out, cyc);
    long long h[8];
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("mode=%2d: MFMA wave %.1f cycles/MFMA, VALU wave %.2f cycles/fma\n", mode,
           (double)h[0] / (n * 16.0), (double)h[4] / (n * 64.0));
  };
  run(std::integral_constant<int, 1>{});
  run(std::integral_constant<int, 2>{});
  run(std::integral_constant<int, 3>{});
  run(std::integral_constant<int, 7>{});
  run(std::integral_constant<int, 10>{});
  run(std::integral_constant<int, 11>{});
  run(std::integral_constant<int, 15>{});
  return 0;
}
