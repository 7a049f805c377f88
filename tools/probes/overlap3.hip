// overlap3.hip -- cost of vector instructions placed between the matrix instructions of the
// same wave, with one or two such waves per SIMD (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;

// K vector instructions after every matrix instruction; PK: packed fma, else plain fma; DEP: the
// vector instructions form one dependent chain, else four independent ones
template <int K, bool PK, bool DEP>
__global__ __launch_bounds__(512) void k_ov(int n, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  f2 x[4];
  for (int c = 0; c < 4; ++c) x[c] = f2{threadIdx.x * 1e-3f + c, 1.0f + c};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  const f2 pa = {a, a}, pb = {b, b};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int cc = DEP ? 0 : (k & 3);
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[cc]) : "v"(pa), "v"(pb));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[cc].x) : "v"(a), "v"(b));
      }
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, a1, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int cc = DEP ? 0 : (k & 3);
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[cc]) : "v"(pa), "v"(pb));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[cc].x) : "v"(a), "v"(b));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0.x + a1.y;
  for (int c = 0; c < 4; ++c) s += x[c].x + x[c].y;
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int K, bool PK, bool DEP>
void run(float* out, long long* cyc) {
  const int n = 1000;
  for (int waves : {4, 8}) {
    k_ov<K, PK, DEP><<<1, waves * 64>>>(n, out, cyc);
    k_ov<K, PK, DEP><<<1, waves * 64>>>(n, out, cyc);
    long long h[8] = {};
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("K=%d %s %s waves/SIMD=%d: %.1f cycles per matrix instruction per SIMD (slowest wave), wave 0 %.1f\n", K,
           PK ? "pk_fma" : "fma   ", DEP ? "dep  " : "indep", waves / 4, (double)mx / (n * 16.0) / (waves / 4),
           (double)h[0] / (n * 16.0) / (waves / 4));
  }
}

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&cyc, 64);
  run<0, true, false>(out, cyc);
  run<1, true, false>(out, cyc);
  run<2, true, false>(out, cyc);
  run<3, true, false>(out, cyc);
  run<4, true, false>(out, cyc);
  run<2, true, true>(out, cyc);
  run<2, false, false>(out, cyc);
  run<4, false, false>(out, cyc);
  return 0;
}
