// head_probe.hip -- what one k-block of a vector-ALU output head costs a lone wave: 16 dependent
// fused multiply-adds fed from LDS (activations) and from LDS / the scalar cache (weights),
// software-pipelined one block ahead.  One wave of the workgroup works, the others wait at a
// barrier, as in the last phase of the networks.
// Build: hipcc --offload-arch=gfx950 -O3 -o head_probe.bin head_probe.hip ; run on one GPU.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f4;
typedef const __attribute__((address_space(4))) f4 cf4;
#define NKB 32
#define LDS_FLOATS (17 * (NKB * 16 + 4))

__device__ __forceinline__ long long now() { return __builtin_amdgcn_s_memtime(); }

#define STEP(A, W, c)                  \
  acc = __builtin_fmaf(A[0].c, W[0].c, acc); \
  acc = __builtin_fmaf(A[1].c, W[1].c, acc); \
  acc = __builtin_fmaf(A[2].c, W[2].c, acc); \
  acc = __builtin_fmaf(A[3].c, W[3].c, acc)
#define BLOCK(A, W) \
  STEP(A, W, x);    \
  STEP(A, W, y);    \
  STEP(A, W, z);    \
  STEP(A, W, w)

// kind 0: fma chain only (operands loaded once); 1: 4 ds_read_b128 per block, no fma; 2: 4 ds_read_b128
// (activations) + chain, weights fixed; 3: 8 ds_read_b128 (activations + weights) + chain; 4: 4 ds_read_b128
// + 4 scalar loads + chain; 5: like 2 with 8 ds_read_b64 instead; 6: like 2, all 64 lanes active
template <int kind>
__global__ __launch_bounds__(512) void k_probe(int lanes, const float* wg, long long* out, float* sink) {
  __shared__ float lds[LDS_FLOATS];
  const int tid = threadIdx.x;
  for (int i = tid; i < LDS_FLOATS; i += blockDim.x) lds[i] = 1.0f + (i & 7) * 0.125f;
  __syncthreads();
  float acc = 0.0f;
  long long t0 = 0, t1 = 0;
  if (tid < lanes) {
    const float* arow = lds + (tid & 15) * (NKB * 16 + 4);
    const float* wrow = lds + 16 * (NKB * 16 + 4);
    const int kpq = NKB * 4;
    cf4* ws = (cf4*)(uintptr_t)wg;
    f4 A[4], W[4], A2[4], W2[4];
    for (int m = 0; m < 4; ++m) {
      A[m] = *(const f4*)(arow + m * kpq);
      if (kind == 4) W[m] = ws[m * 16];
      else W[m] = *(const f4*)(wrow + m * kpq);
      A2[m] = A[m];
      W2[m] = W[m];
    }
    t0 = now();
#pragma unroll 1
    for (int kb = 0; kb < NKB; kb += 2) {
      asm volatile("" : "+v"(A[0].w), "+v"(A[1].w), "+v"(A[2].w), "+v"(A[3].w));
      if constexpr (kind == 4) asm volatile("" : "+s"(W[0].w), "+s"(W[1].w), "+s"(W[2].w), "+s"(W[3].w));
      if constexpr (kind == 3) asm volatile("" : "+v"(W[0].w), "+v"(W[1].w), "+v"(W[2].w), "+v"(W[3].w));
      const int k1 = kb + 1;
      if (kind >= 1 && kind != 5)
        for (int m = 0; m < 4; ++m) A2[m] = *(const f4*)(arow + m * kpq + 4 * k1);
      if (kind == 5)
        for (int m = 0; m < 4; ++m) {
          const float2 lo = *(const float2*)(arow + m * kpq + 4 * k1), hi = *(const float2*)(arow + m * kpq + 4 * k1 + 2);
          A2[m] = f4{lo.x, lo.y, hi.x, hi.y};
        }
      if (kind == 3)
        for (int m = 0; m < 4; ++m) W2[m] = *(const f4*)(wrow + m * kpq + 4 * k1);
      if (kind == 4)
        for (int m = 0; m < 4; ++m) W2[m] = ws[k1 * 64 + m * 16];
      __builtin_amdgcn_sched_barrier(0);
      if (kind != 1) {
        BLOCK(A, W);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(A2[0].w), "+v"(A2[1].w), "+v"(A2[2].w), "+v"(A2[3].w));
      if constexpr (kind == 4) asm volatile("" : "+s"(W2[0].w), "+s"(W2[1].w), "+s"(W2[2].w), "+s"(W2[3].w));
      if constexpr (kind == 3) asm volatile("" : "+v"(W2[0].w), "+v"(W2[1].w), "+v"(W2[2].w), "+v"(W2[3].w));
      const int k2 = kb + 2 < NKB ? kb + 2 : kb;
      if (kind >= 1 && kind != 5)
        for (int m = 0; m < 4; ++m) A[m] = *(const f4*)(arow + m * kpq + 4 * k2);
      if (kind == 5)
        for (int m = 0; m < 4; ++m) {
          const float2 lo = *(const float2*)(arow + m * kpq + 4 * k2), hi = *(const float2*)(arow + m * kpq + 4 * k2 + 2);
          A[m] = f4{lo.x, lo.y, hi.x, hi.y};
        }
      if (kind == 3)
        for (int m = 0; m < 4; ++m) W[m] = *(const f4*)(wrow + m * kpq + 4 * k2);
      if (kind == 4)
        for (int m = 0; m < 4; ++m) W[m] = ws[k2 * 64 + m * 16];
      __builtin_amdgcn_sched_barrier(0);
      if (kind != 1) {
        BLOCK(A2, W2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int m = 0; m < 4; ++m) acc += A[m].x + A2[m].y + W[m].z + W2[m].w;
    t1 = now();
  }
  __syncthreads();
  if (tid == 0) out[0] = t1 - t0;
  sink[tid] = acc;
}

int main() {
  long long* out;
  float *sink, *wg;
  hipMalloc(&out, 8 * sizeof(long long));
  hipMalloc(&sink, 512 * sizeof(float));
  hipMalloc(&wg, NKB * 64 * 4 * sizeof(float) + 4096);
  hipMemset(wg, 0, NKB * 64 * 4 * sizeof(float) + 4096);
  const char* names[] = {"fma chain only",
                         "4 ds_read_b128 only",
                         "4 ds_read_b128 + chain",
                         "8 ds_read_b128 + chain",
                         "4 ds_read_b128 + 4 s_load_dwordx4 + chain",
                         "8 ds_read_b64 + chain"};
  for (int kind = 0; kind < 6; ++kind)
    for (int lanes = 16; lanes <= 64; lanes += 48) {
      long long h = 0;
      for (int rep = 0; rep < 3; ++rep) {
        switch (kind) {
          case 0: k_probe<0><<<1, 512>>>(lanes, wg, out, sink); break;
          case 1: k_probe<1><<<1, 512>>>(lanes, wg, out, sink); break;
          case 2: k_probe<2><<<1, 512>>>(lanes, wg, out, sink); break;
          case 3: k_probe<3><<<1, 512>>>(lanes, wg, out, sink); break;
          case 4: k_probe<4><<<1, 512>>>(lanes, wg, out, sink); break;
          default: k_probe<5><<<1, 512>>>(lanes, wg, out, sink); break;
        }
        hipDeviceSynchronize();
        hipMemcpy(&h, out, sizeof h, hipMemcpyDeviceToHost);
      }
      printf("%-44s lanes=%2d : %7.1f ticks per k-block (16 fma)\n", names[kind], lanes, (double)h / NKB);
    }
  return 0;
}
