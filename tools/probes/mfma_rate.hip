// mfma_rate.hip -- issue rate / dependent latency of v_mfma_f32_16x16x4_f32 on gfx950
// (diagnostic).  One workgroup; `waves` waves; each wave runs CH independent accumulator
// chains of N MFMAs and reports cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f4;

template <int CH>
__global__ void k_rate(int n, float* out, long long* cyc) {
  f4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f4{0, 0, 0, 0};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  out[threadIdx.x] = s;
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int CH>
void run(int waves, int n) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 4096);
  hipMalloc(&cyc, 16 * 8);
  k_rate<CH><<<1, waves * 64>>>(n, out, cyc);
  k_rate<CH><<<1, waves * 64>>>(n, out, cyc);
  long long h[16];
  hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  const double per = (double)h[0] / ((double)n * 8 * CH);
  const int waves_per_simd = (waves + 3) / 4;
  printf("waves=%2d chains=%d: %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD\n", waves, CH, per,
         per / waves_per_simd);
}

int main() {
  const int n = 2000;
  for (int waves : {1, 4, 8, 16}) {
    run<1>(waves, n);
    run<2>(waves, n);
    run<4>(waves, n);
  }
  return 0;
}
