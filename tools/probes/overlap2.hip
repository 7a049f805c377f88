// overlap2.hip -- which vector instructions of wave B fit under a saturated f32 MFMA stream of
// wave A on the same SIMD (diagnostic).  Workgroup of 8 waves: waves 0-3 run MFMA chains, waves
// 4-7 the instruction under test, as one dependent chain or four independent ones.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;

// KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_rcp_f32, 3: v_mul_f32 by an inline constant, 4: v_add_u32
template <int KIND, int CHAINS, bool MFMA, bool PRIO>
__global__ __launch_bounds__(512) void k_ov(int n, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  f2 x[4];
  for (int c = 0; c < 4; ++c) x[c] = f2{threadIdx.x * 1e-3f + c, 1.0f + c};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  const f2 pa = {a, a}, pb = {b, b};
  int iv[4] = {(int)threadIdx.x, 1, 2, 3};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (MFMA) {
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, a1, 0, 0, 0);
        }
      }
    }
  } else {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int cc = CHAINS == 4 ? c : 0;
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[cc].x) : "v"(a), "v"(b));
          if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[cc]) : "v"(pa), "v"(pb));
          if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[cc].x));
          if (KIND == 3) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x[cc].x));
          if (KIND == 4) asm volatile("v_add_u32 %0, 3, %0" : "+v"(iv[cc]));
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0.x + a1.y;
  for (int c = 0; c < 4; ++c) s += x[c].x + x[c].y + iv[c];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int KIND, int CHAINS, bool MFMA, bool PRIO>
void run(const char* name, float* out, long long* cyc) {
  const int n = 1000;
  k_ov<KIND, CHAINS, MFMA, PRIO><<<1, 512>>>(n, out, cyc);
  k_ov<KIND, CHAINS, MFMA, PRIO><<<1, 512>>>(n, out, cyc);
  long long h[8];
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-12s chains=%d mfma=%d prio=%d: MFMA wave %.1f cycles/MFMA, other wave %.2f cycles/instr\n", name, CHAINS,
         (int)MFMA, (int)PRIO, (double)h[0] / (n * 16.0), (double)h[4] / (n * 64.0));
}

#define ALL(KIND, NAME)                      \
  run<KIND, 1, false, false>(NAME, out, cyc); \
  run<KIND, 4, false, false>(NAME, out, cyc); \
  run<KIND, 1, true, false>(NAME, out, cyc);  \
  run<KIND, 4, true, false>(NAME, out, cyc);  \
  run<KIND, 4, true, true>(NAME, out, cyc);

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&cyc, 64);
  ALL(0, "v_fma_f32")
  ALL(1, "v_pk_fma_f32")
  ALL(2, "v_rcp_f32")
  ALL(3, "v_mul_f32")
  ALL(4, "v_add_u32")
  return 0;
}
