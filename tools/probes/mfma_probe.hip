// Probe: per-unit timing of the MFMA + LDS-A + global-B loop for one workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const f4* __restrict__ w, float* out, long long* stamps, int reps) {
  __shared__ __attribute__((aligned(16))) float act[16 * 260];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16 * 260; i += 512) act[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  const f4* w0 = w + (size_t)(wave * 2) * 16 * 64 + lane;
  const f4* w1 = w0 + 16 * 64;
  const float* ap = act + (lane & 15) * 260 + (lane >> 4) * 64;
  f4 W0[16], W1[16];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) { W0[kb] = w0[kb * 64]; W1[kb] = w1[kb * 64]; }
  f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      f4 a4;
      if (MODE & 1) a4 = *reinterpret_cast<const f4*>(ap + kb * 4); else a4 = f4{1.f, 2.f, 3.f, 4.f};
      f4 b0 = W0[kb], b1 = W1[kb];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, acc1, 0, 0, 0);
      if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
      if (MODE & 2) { W0[kb] = w0[(size_t)((r + 1) & 3) * 8 * 2 * 16 * 64 + kb * 64]; W1[kb] = w1[(size_t)((r + 1) & 3) * 8 * 2 * 16 * 64 + kb * 64]; }
      if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
  out[blockIdx.x * 512 + tid] = acc0.x + acc1.y + acc0.z + acc1.w;
}

int main() {
  const int reps = 64;
  f4* w; float* out; long long* st;
  size_t wbytes = (size_t)4 * 8 * 2 * 16 * 64 * sizeof(f4) + (1 << 20);
  hipMalloc(&w, wbytes); hipMemset(w, 0, wbytes);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 8 * 8);
  std::vector<long long> h(256 * 8);
  for (int blocks : {1, 256}) {
    for (int mode : {0, 3, 7, 11, 15}) {
      for (int it = 0; it < 3; ++it) {
        switch (mode) {
#define CASE(m) case m: probe<m><<<blocks, 512>>>(w, out, st, reps); break;
          CASE(0) CASE(3) CASE(7) CASE(11) CASE(15)
        }
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), st, blocks * 8 * 8, hipMemcpyDeviceToHost);
      long long mx = 0; for (int i = 0; i < blocks * 8; ++i) mx = h[i] > mx ? h[i] : mx;
      printf("blocks=%3d mode=%d (ldsA=%d gloadB=%d barrier=%d): %.0f memtime ticks per unit of 128 MFMA (max wave)\n", blocks, mode, mode & 1, (mode >> 1) & 1, (mode >> 2) & 1, (double)mx / reps); if (mode & 8) printf("   ^ with sched_barrier\n");
    }
  }
  return 0;
}
