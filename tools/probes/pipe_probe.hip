// pipe_probe.hip -- upper bound of a tile-major network phase whose activation functions are
// dealt out between the matrix instructions of the SAME wave, phases ordered by LDS counters
// instead of barriers (diagnostic; arithmetic of the right shape, not of a real network).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I alphazero_amd/csrc -I include \
//         -o tools/probes/pipe_probe.bin tools/probes/pipe_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "az_mlp.h"

// LDS phase counters as in tools/probes/phase_counters.patch (the library orders its phases by barriers).
__device__ int az_sync_timeouts;
typedef __attribute__((address_space(3))) uint32_t az_lds_u32;
struct PhaseSync {
  az_lds_u32* ctr;
  int mine, done, seen[2];
};
AZ_DEV void az_phase_wait(PhaseSync& ps, int cls, int q) {
  if (ps.seen[cls] >= q) return;
  const uint32_t need = (uint32_t)(q * (AZ_MLP_WAVES / 2));
  for (int spin = 0;; ++spin) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(*(volatile az_lds_u32*)(ps.ctr + cls));
    if (v >= need) break;
    if (spin >= (1 << 22)) {
      if ((threadIdx.x & 63) == 0) atomicAdd(&az_sync_timeouts, 1);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
  ps.seen[cls] = q;
}

struct Sw {  // az_swish2 on two pairs (one tile's four values per thread), one step at a time
  az_f32x2 x[2], t[2], m[2], kf[2], r[2], rr[2], p[2], sc[2], d[2], y[2], e[2], q[2];
};
#define SW2(arr, expr) _Pragma("unroll") for (int i = 0; i < 2; ++i) s.arr[i] = (expr)
template <int STEP>
AZ_DEV void sw_step(Sw& s) {
  const az_f32x2 one = az_splat2(1.0f);
  if (STEP == 0) SW2(t, (az_f32x2{__builtin_amdgcn_fmed3f(-s.x[i].x, -64.0f, 64.0f), __builtin_amdgcn_fmed3f(-s.x[i].y, -64.0f, 64.0f)}));
  if (STEP == 1) SW2(m, s.t[i] * az_splat2(1.44269502162933349609375f));
  if (STEP == 2) SW2(m, s.m[i] + az_splat2(12582912.0f));
  if (STEP == 3) SW2(kf, s.m[i] - az_splat2(12582912.0f));
  if (STEP == 4) SW2(r, az_fma2(s.kf[i], az_splat2(-0.693359375f), s.t[i]));
  if (STEP == 5) SW2(r, az_fma2(s.kf[i], az_splat2(2.12194440e-4f), s.r[i]));
  if (STEP == 6) SW2(p, az_fma2(az_splat2(1.9875691500e-4f), s.r[i], az_splat2(1.3981999507e-3f)));
  if (STEP == 7) SW2(p, az_fma2(s.p[i], s.r[i], az_splat2(8.3334519073e-3f)));
  if (STEP == 8) SW2(p, az_fma2(s.p[i], s.r[i], az_splat2(4.1665795894e-2f)));
  if (STEP == 9) SW2(p, az_fma2(s.p[i], s.r[i], az_splat2(1.6666665459e-1f)));
  if (STEP == 10) SW2(p, az_fma2(s.p[i], s.r[i], az_splat2(5.0000001201e-1f)));
  if (STEP == 11) SW2(rr, s.r[i] * s.r[i]);
  if (STEP == 12) SW2(p, az_fma2(s.p[i], s.rr[i], s.r[i]));
  if (STEP == 13) SW2(p, s.p[i] + one);
  if (STEP == 14) SW2(sc, (az_f32x2{az_u2f((az_f2u(s.m[i].x) << 23) + 0x3f800000u), az_u2f((az_f2u(s.m[i].y) << 23) + 0x3f800000u)}));
  if (STEP == 15) SW2(p, s.p[i] * s.sc[i]);
  if (STEP == 16) SW2(d, one + s.p[i]);
  if (STEP == 17) SW2(y, (az_f32x2{__builtin_amdgcn_rcpf(s.d[i].x), __builtin_amdgcn_rcpf(s.d[i].y)}));
  if (STEP == 18) SW2(e, az_fma2(-s.d[i], s.y[i], one));
  if (STEP == 19) SW2(y, az_fma2(s.e[i], s.y[i], s.y[i]));
  if (STEP == 20) SW2(e, az_fma2(-s.d[i], s.y[i], one));
  if (STEP == 21) SW2(q, az_fma2(s.e[i], s.y[i], s.y[i]));
  if (STEP == 22) SW2(e, az_fma2(-s.d[i], s.q[i], one));
  if (STEP == 23) SW2(q, az_fma2(s.e[i], s.y[i], s.q[i]));
  if (STEP == 24) SW2(x, s.x[i] * s.q[i]);
}

// eight k-blocks [KB0, KB0 + 8) of one tile; with SWISH, step 4 * (kb - KB0) + j of `sw` follows the
// j-th matrix instruction of the k-block, and the finished values are stored after the last
template <int KB0, bool SWISH, bool REFILL>
AZ_DEV void half_tile(az_f32x4& acc, az_f32x4 (&W)[AZ_SLOTS], const float* ap, Sw& sw, float* outp,
                      const az_f32x4* wnext, int lane) {
  az_f32x4 a[2];
  a[0] = *reinterpret_cast<const az_f32x4*>(ap + KB0 * 4);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int kb = KB0 + k;
    if (k + 1 < 8) a[(k + 1) & 1] = *reinterpret_cast<const az_f32x4*>(ap + (kb + 1) * 4);
    const az_f32x4 a4 = a[k & 1], b = W[kb];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b.x, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#define STEP_AT(J)                                   \
  if (SWISH) {                                       \
    switch (4 * k + J) {                             \
      case 0: sw_step<0>(sw); break;                 \
      case 1: sw_step<1>(sw); break;                 \
      case 2: sw_step<2>(sw); break;                 \
      case 3: sw_step<3>(sw); break;                 \
      case 4: sw_step<4>(sw); break;                 \
      case 5: sw_step<5>(sw); break;                 \
      case 6: sw_step<6>(sw); break;                 \
      case 7: sw_step<7>(sw); break;                 \
      case 8: sw_step<8>(sw); break;                 \
      case 9: sw_step<9>(sw); break;                 \
      case 10: sw_step<10>(sw); break;               \
      case 11: sw_step<11>(sw); break;               \
      case 12: sw_step<12>(sw); break;               \
      case 13: sw_step<13>(sw); break;               \
      case 14: sw_step<14>(sw); break;               \
      case 15: sw_step<15>(sw); break;               \
      case 16: sw_step<16>(sw); break;               \
      case 17: sw_step<17>(sw); break;               \
      case 18: sw_step<18>(sw); break;               \
      case 19: sw_step<19>(sw); break;               \
      case 20: sw_step<20>(sw); break;               \
      case 21: sw_step<21>(sw); break;               \
      case 22: sw_step<22>(sw); break;               \
      case 23: sw_step<23>(sw); break;               \
      case 24: sw_step<24>(sw); break;               \
      default: break;                                \
    }                                                \
    __builtin_amdgcn_sched_barrier(0);               \
  }
    STEP_AT(0)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b.y, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    STEP_AT(1)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b.z, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    STEP_AT(2)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    STEP_AT(3)
    if (SWISH && k == 6) {
      outp[0] = sw.x[0].x;
      outp[260] = sw.x[0].y;
      outp[520] = sw.x[1].x;
      outp[780] = sw.x[1].y;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (REFILL) {
      W[kb] = az_ldg4_lane(wnext + kb * 64, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// MODE 0: counters between the half tiles as the real schedule would need them; 1: none (bound)
template <int MODE>
__global__ __launch_bounds__(512) void k_pipe(const float* w, int phases, float* out, long long* cyc, int* timeouts) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  for (int i = tid; i < 2 * 16 * 260; i += blockDim.x) lds[i] = 0.001f * (i % 97);
  uint32_t* sync = reinterpret_cast<uint32_t*>(lds + 2 * 16 * 260);
  if (tid < 4) sync[tid] = 0u;
  __syncthreads();
  const az_f32x4* wb = reinterpret_cast<const az_f32x4*>(w) + (size_t)(blockIdx.x % 4) * 16 * 16 * 64;
  const az_f32x4* wa = wb + (size_t)wave * 16 * 64;        // tile `wave`
  const az_f32x4* wbb = wb + (size_t)(wave + 8) * 16 * 64;  // tile `wave + 8`
  az_f32x4 WA[AZ_SLOTS], WB[AZ_SLOTS];
#pragma unroll
  for (int kb = 0; kb < AZ_SLOTS; ++kb) {
    WA[kb] = az_ldg4(wa + kb * 64 + lane);
    WB[kb] = az_ldg4(wbb + kb * 64 + lane);
  }
  PhaseSync ps;
  ps.ctr = (az_lds_u32*)sync;
  ps.mine = 0;
  ps.done = 0;
  ps.seen[0] = ps.seen[1] = 0;
  az_f32x4 accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
  Sw swA{}, swB{};
  swB.x[0] = swB.x[1] = az_f32x2{0.1f, 0.2f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int ph = 0; ph < phases; ++ph) {
    const float* in = lds + (ph & 1) * 16 * 260;
    float* outb = lds + ((ph + 1) & 1) * 16 * 260;
    const float* ap = in + (lane & 15) * 260 + (lane >> 4) * 64;
    float* outA = outb + (lane >> 4) * 4 * 260 + az_act_index(wave * 16 + (lane & 15), 256);
    float* outB = outb + (lane >> 4) * 4 * 260 + az_act_index((wave + 8) * 16 + (lane & 15), 256);
    float* outBprev = const_cast<float*>(in) + (lane >> 4) * 4 * 260 + az_act_index((wave + 8) * 16 + (lane & 15), 256);
    // S1: tile A, k-blocks 0..7, with the previous phase's tile B activation functions
    if (MODE == 0) az_phase_wait(ps, 0, 2 * ph);  // every wave's tile A of the previous phase is stored
    accA = az_f32x4{0, 0, 0, 0};
    half_tile<0, true, true>(accA, WA, ap, swB, outBprev, wa, lane);
    if (MODE == 0) {
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(ps.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      az_phase_wait(ps, 1, 2 * (ph + 1));  // every wave's tile B of the previous phase is stored
    }
    half_tile<8, false, true>(accA, WA, ap, swA, outA, wa, lane);
    swA.x[0] = az_f32x2{accA.x, accA.y};
    swA.x[1] = az_f32x2{accA.z, accA.w};
    accB = az_f32x4{0, 0, 0, 0};
    half_tile<0, true, true>(accB, WB, ap, swA, outA, wbb, lane);
    if (MODE == 0) {
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(ps.ctr + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    half_tile<8, false, true>(accB, WB, ap, swB, outB, wbb, lane);
    swB.x[0] = az_f32x2{accB.x, accB.y};
    swB.x[1] = az_f32x2{accB.z, accB.w};
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = accA.x + accB.y + swA.x[0].x + swB.x[1].y;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
  if (blockIdx.x == 0 && tid == 0) *timeouts = az_sync_timeouts;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;
  const int phases = 200;
  float *w, *out;
  long long* cyc;
  int* to;
  const size_t wfloats = (size_t)4 * 16 * 16 * 64 * 4 + 4096;
  (void)hipMalloc(&w, wfloats * 4);
  (void)hipMemset(w, 0, wfloats * 4);
  (void)hipMalloc(&out, (size_t)wgs * 512 * 4);
  (void)hipMalloc(&cyc, 64);
  (void)hipMalloc(&to, 4);
  const size_t lds = (2 * 16 * 260 + 16) * 4;
  auto run = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 2; ++rep) kern<<<wgs, 512, lds>>>(w, phases, out, cyc, to);
    long long h[8];
    int hto = 0;
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost);
    printf("%-10s: %.0f / %.0f cycles per phase (waves 0 / 4; 256 matrix instructions per SIMD: floor 8192), timeouts %d\n", name,
           (double)h[0] / phases, (double)h[4] / phases, hto);
  };
  run(k_pipe<0>, "counters");
  run(k_pipe<1>, "free");
  return 0;
}
