// mfma_4x4_probe.hip -- v_mfma_f32_4x4x1_16B_f32 on gfx950 (diagnostic): issue rate, and whether a
// dense layer evaluated with it for FOUR rows gives the bits of the k-ordered fused-multiply-add
// chain (the numerics specification of the networks), as v_mfma_f32_16x16x4_f32 does for sixteen.
//
// The instruction: sixteen independent 4 x 4 blocks, D_b += A_b (4 x 1) * B_b (1 x 4).  Lane l
// supplies A_b[i] and B_b[j] with b = l / 4, i = j = l % 4, and holds D_b[0..3][j] in four
// registers.  With the same four rows in every block (A_b[i] = x[i][k]) and B_b[j] = W[k][4 b + j],
// one instruction covers 4 rows x 64 columns x 1 input.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_4x4_probe mfma_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
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
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
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
  hipMalloc(&out, 4096 * 4);
  hipMalloc(&cyc, 16 * 8);
  k_rate<CH><<<1, waves * 64>>>(n, out, cyc);
  k_rate<CH><<<1, waves * 64>>>(n, out, cyc);
  long long h[16];
  hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  long long slowest = 0;
  for (int w = 0; w < waves; ++w) slowest = h[w] > slowest ? h[w] : slowest;
  const double per = (double)slowest / ((double)n * 8 * CH);
  const int waves_per_simd = (waves + 3) / 4;
  printf("waves=%2d chains=%d: %.1f cycles per MFMA per wave (slowest wave), %.1f per MFMA per SIMD (256 multiply-adds each)\n",
         waves, CH, per, per / waves_per_simd);
  hipFree(out);
  hipFree(cyc);
}

// y[4][64] = x[4][K] * W[K][64], one wave, k in order
__global__ void k_layer(const float* __restrict__ x, const float* __restrict__ w, int K, float* __restrict__ y) {
  const int l = threadIdx.x, i = l & 3;
  f4 acc = {0, 0, 0, 0};
  for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x[i * K + k], w[k * 64 + l], acc, 0, 0, 0);
  // D_b[r][j]: row r, column 4 b + j = l
  y[0 * 64 + l] = acc.x;
  y[1 * 64 + l] = acc.y;
  y[2 * 64 + l] = acc.z;
  y[3 * 64 + l] = acc.w;
}

int main() {
  const int n = 2000;
  for (int waves : {1, 4, 8, 12, 16}) {
    run<1>(waves, n);
    run<2>(waves, n);
    run<4>(waves, n);
  }
  const int K = 256;
  std::vector<float> x(4 * K), w(K * 64), y(4 * 64), ref(4 * 64);
  srand(3);
  for (auto& v : x) v = (float)rand() / RAND_MAX * 4.0f - 2.0f;
  for (auto& v : w) v = (float)rand() / RAND_MAX - 0.5f;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 64; ++c) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k) acc = fmaf(x[r * K + k], w[k * 64 + c], acc);
      ref[r * 64 + c] = acc;
    }
  float *dx, *dw, *dy;
  hipMalloc(&dx, x.size() * 4);
  hipMalloc(&dw, w.size() * 4);
  hipMalloc(&dy, y.size() * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  k_layer<<<1, 64>>>(dx, dw, K, dy);
  hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (size_t e = 0; e < y.size(); ++e) bad += memcmp(&y[e], &ref[e], 4) != 0;
  printf("4 rows x 64 columns x %d inputs against the k-ordered fmaf chain: %d of %zu outputs differ\n", K, bad,
         y.size());
  return bad != 0;
}
