// lat_probe.hip -- latency / issue-rate probe for the building blocks of the tree search:
// dependent VALU chains, LDS pointer chase, readlane pointer chase, DPP hand-off, f32 division,
// each with 1 and 2 waves per SIMD (4 or 8 waves in the workgroup).
// Build: hipcc --offload-arch=gfx950 -O3 -o lat_probe.bin lat_probe.hip ; run on one GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 2048

__device__ __forceinline__ long long now() { return __builtin_amdgcn_s_memtime(); }

// kind 0: dependent fma chain; 1: dependent mul+add (no contraction); 2: LDS pointer chase;
// 3: readlane pointer chase; 4: chain step mul, add, cndmask, dpp; 5: dependent f32 division;
// 6: 4 independent fma chains; 7: dependent v_exp_f32 (transcendental); 8: LDS chase, 2 chains
// interleaved; 9: ds_bpermute chain
__global__ __launch_bounds__(512) void k_probe(int kind, float seed, long long* out, float* sink) {
  __shared__ int lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (i * 17 + 5) & 4095;
  __syncthreads();
  float x = seed + tid * 1e-3f, y = seed * 0.5f, z = 1.0f, w = 2.0f;
  int k = tid & 63;
  int reg = (tid * 7 + 3) & 63;  // readlane table: lane i holds the next index
  long long t0 = now();
  if (kind == 0) {
#pragma unroll 16
    for (int i = 0; i < N_IT; ++i) x = __builtin_fmaf(x, 0.999f, 0.001f);
  } else if (kind == 1) {
#pragma unroll 16
    for (int i = 0; i < N_IT; ++i) {
      float m = x * 0.999f;
      asm volatile("" : "+v"(m));
      x = m + 0.001f;
    }
  } else if (kind == 2) {
#pragma unroll 16
    for (int i = 0; i < N_IT; ++i) k = lds[k];
  } else if (kind == 3) {
    int s = 0;
#pragma unroll 16
    for (int i = 0; i < N_IT; ++i) s = __builtin_amdgcn_readlane(reg, s & 63);
    k = s;
  } else if (kind == 4) {
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
      float m = x * 0.999f;
      asm volatile("" : "+v"(m));
      float yy = m + 0.001f;
      z = (i == (tid & 31)) ? yy : z;
      x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, yy), 0x130, 0xf, 0xf, false));
    }
  } else if (kind == 5) {
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) x = (x + 1.5f) / (x + 2.5f);
  } else if (kind == 6) {
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
      x = __builtin_fmaf(x, 0.999f, 0.001f);
      y = __builtin_fmaf(y, 0.999f, 0.001f);
      z = __builtin_fmaf(z, 0.999f, 0.001f);
      w = __builtin_fmaf(w, 0.999f, 0.001f);
    }
  } else if (kind == 7) {
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) x = __builtin_amdgcn_exp2f(x * 0.01f);
  } else if (kind == 8) {
    int k2 = (tid * 3 + 1) & 63;
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
      k = lds[k];
      k2 = lds[k2];
    }
    k += k2;
  } else if (kind == 9) {
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) k = __builtin_amdgcn_ds_bpermute(((tid + 1) & 63) * 4, k);
  }
  long long t1 = now();
  if ((tid & 63) == 0) out[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = x + y + z + w + (float)k;
}

int main() {
  long long* out;
  float* sink;
  hipMalloc(&out, 8 * 8 * sizeof(long long));
  hipMalloc(&sink, 8 * 512 * sizeof(float));
  const char* names[] = {"dependent fma", "dependent mul+add", "LDS pointer chase", "readlane chase",
                         "chain step (mul,add,cndmask,dpp)", "dependent f32 div", "4 independent fma",
                         "dependent v_exp", "LDS chase x2 interleaved", "ds_bpermute chain"};
  const int per_it[] = {1, 2, 1, 1, 1, 1, 4, 1, 2, 1};
  for (int kind = 0; kind < 10; ++kind)
    for (int waves = 4; waves <= 8; waves += 4) {
      k_probe<<<1, waves * 64>>>(kind, 1.0f, out, sink);
      hipDeviceSynchronize();
      k_probe<<<1, waves * 64>>>(kind, 1.0f, out, sink);
      hipDeviceSynchronize();
      long long h[8];
      hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
      double mx = 0;
      for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
      printf("%-36s waves/SIMD=%d : %8.1f ticks per iteration (%d op(s) each), total %lld\n", names[kind],
             waves / 4, mx / N_IT, per_it[kind], (long long)mx);
    }
  // s_memtime tick vs wall clock: time a long kernel with events
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  for (int i = 0; i < 50; ++i) k_probe<<<1, 256>>>(0, 1.0f, out, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  long long h[8];
  hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  printf("50 launches of the fma chain: %.3f ms wall, %lld ticks per launch -> >= %.1f ticks/us\n", ms, h[0],
         h[0] * 50 / (ms * 1000.0));
  return 0;
}
