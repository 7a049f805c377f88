// l2_bw.hip -- L2 -> CU streaming bandwidth of one workgroup (diagnostic): 8 waves stream
// an L2-resident buffer with dwordx4 loads, UNROLL loads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%d %s\n", __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

// MODE 0: plain; 1: nontemporal; 2: waves interleaved at 1 KiB granularity (wave w takes block i*8+w)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(512) void k_bw(const f4* __restrict__ buf, int n_f4, int reps, float* out, long long* cyc) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per_wave = n_f4 / 8;   // f4 elements per wave region
  f4 acc = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < per_wave / 64; i += UNROLL) {
      f4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int blk = MODE == 2 ? (i + u) * 8 + wave : wave * (per_wave / 64) + i + u;
        const f4* p = buf + (size_t)blk * 64 + lane;
        v[u] = MODE == 1 ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int UNROLL>
void run(const f4* buf, int n_f4, int wgs, float* out, long long* cyc) {
  const int reps = 20;
  k_bw<MODE, UNROLL><<<wgs, 512>>>(buf, n_f4, reps, out, cyc);
  k_bw<MODE, UNROLL><<<wgs, 512>>>(buf, n_f4, reps, out, cyc);
  long long h;
  CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("mode=%d unroll=%2d wgs=%3d: %.1f B/clk per CU\n", MODE, UNROLL, wgs, (double)n_f4 * 16 * reps / h);
}

int main() {
  const int n_f4 = 1664 * 1024 / 16;  // 1.66 MB
  f4* buf; float* out; long long* cyc;
  CK(hipMalloc(&buf, (size_t)n_f4 * 16));
  CK(hipMemset(buf, 0, (size_t)n_f4 * 16));
  CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMalloc(&cyc, 8));
  for (int wgs : {1, 256}) {
    run<0, 4>(buf, n_f4, wgs, out, cyc);
    run<0, 8>(buf, n_f4, wgs, out, cyc);
    run<0, 16>(buf, n_f4, wgs, out, cyc);
    run<0, 32>(buf, n_f4, wgs, out, cyc);
    run<1, 16>(buf, n_f4, wgs, out, cyc);
    run<2, 16>(buf, n_f4, wgs, out, cyc);
    run<2, 32>(buf, n_f4, wgs, out, cyc);
  }
  return 0;
}
