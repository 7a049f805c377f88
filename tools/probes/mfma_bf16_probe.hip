// mfma_bf16_probe.hip -- operand layout of v_mfma_f32_16x16x32_bf16 on gfx950 and the f32 -> bf16
// conversion the compiler emits (diagnostic).  D = A[16x32] * B[32x16] against a host product of
// the bf16-rounded operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

__global__ void k(const float* A, const float* B, float* D, long long* cyc) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  bf8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[r * 32 + 8 * g + j];        // A: row r, k = 8 g + j
    b[j] = (__bf16)B[(8 * g + j) * 16 + r];      // B: k = 8 g + j, column r
  }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = acc[i];  // D: row 4 g + i, column r
  // issue rate: two independent chains
  f4 c0 = acc, c1 = acc;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 1000; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
  D[256 + lane] = c0[0] + c1[1];
}

static float bf16r(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&x, &u, 4);
  return x;
}

int main() {
  float hA[512], hB[512], hD[512], ref[256];
  for (int i = 0; i < 512; ++i) {
    hA[i] = sinf(0.37f * i) * 1.7f;
    hB[i] = cosf(0.11f * i + 0.3f);
  }
  float *A, *B, *D;
  long long* cyc;
  (void)hipMalloc(&A, 2048); (void)hipMalloc(&B, 2048); (void)hipMalloc(&D, 2048); (void)hipMalloc(&cyc, 8);
  (void)hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
  k<<<1, 64>>>(A, B, D, cyc);
  (void)hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost);
  long long c;
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double s = 0;
      for (int kk = 0; kk < 32; ++kk) s += (double)bf16r(hA[m * 32 + kk]) * (double)bf16r(hB[kk * 16 + n]);
      ref[m * 16 + n] = (float)s;
      worst = fmax(worst, fabs(hD[m * 16 + n] - s));
    }
  printf("max |D - ref| = %.3g (layout %s), %.1f cycles per matrix instruction\n", worst, worst < 1e-4 ? "as assumed" : "DIFFERENT",
         (double)c / 16000.0);
  return 0;
}
