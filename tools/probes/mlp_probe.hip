// mlp_probe.hip -- standalone timing probe of the MLP phase (diagnostic, not part of the
// library): a persistent kernel that runs `iters` forward passes per workgroup, like the
// fused search without its tree phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DAZ_STAMP \
//         -I alphazero_amd/csrc -I include -o tools/probes/mlp_probe.bin tools/probes/mlp_probe.hip
//   mlp_probe.bin [workgroups] [iters] [wrap]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "az_mlp.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <bool WRAP>
__global__ __launch_bounds__(AZ_MLP_WAVES * 64) void k_probe(ModelDev md, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  MlpLds s(lds, md.pol_kp, md.val_kp);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  for (int i = tid; i < 16 * AZ_IN0_STRIDE; i += blockDim.x) s.in0[i] = 0.0f;
  az_mlp_table_build<AZ_MLP_WAVES>(md, s, wave, lane);
  __syncthreads();
  if (tid < 16)
    for (int i = 0; i < md.obs_dim; ++i)
      s.in0[tid * AZ_IN0_STRIDE + az_act_index(i, md.val.kp[0])] = 0.01f * (tid + i + blockIdx.x % 7);
  MlpRegs regs;
  if (WRAP) az_forward_prime<AZ_MLP_WAVES>(md, s, wave, lane, regs);
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    if (!WRAP) az_forward_prime<AZ_MLP_WAVES>(md, s, wave, lane, regs);
    __syncthreads();
    az_forward_tile<AZ_MLP_WAVES, WRAP>(md, s, wave, lane, regs);
    if (tid < 16) {
      acc += s.value[tid * AZ_MAX_ATOMS] + s.logits[tid * AZ_MAX_ACTIONS];
      s.in0[tid * AZ_IN0_STRIDE] = 0.001f * s.logits[tid * AZ_MAX_ACTIONS];
    }
  }
  if (tid < 16) out[blockIdx.x * 16 + tid] = acc;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;
  const int iters = argc > 2 ? atoi(argv[2]) : 64;
  const int wrap = argc > 3 ? atoi(argv[3]) : 1;
  const int32_t psz[] = {4, 32, 32, 32, 32, 2}, vsz[] = {4, 256, 256, 256, 256, 256, 1};
  ModelDev md{};
  md.env_id = 0;
  md.gamma = 0.99f;
  md.obs_dim = 4;
  md.A = 2;
  md.K = 1;
  int64_t off = az_mlp_layout(5, psz, 0, &md.pol);
  off = az_mlp_layout(6, vsz, off, &md.val);
  md.pol_kp = 64;
  md.val_kp = 256;
  std::vector<float> h(off);
  srand(1);
  for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  float *packed, *out;
  CK(hipMalloc(&packed, off * 4));
  CK(hipMalloc(&out, wgs * 16 * 4));
  CK(hipMemcpy(packed, h.data(), off * 4, hipMemcpyHostToDevice));
  md.packed = packed;
  md.mean = md.std = nullptr;
  const size_t ldsb = (size_t)az_mlp_lds_floats(md.pol_kp, md.val_kp) * 4;
  auto kern = wrap ? k_probe<true> : k_probe<false>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    kern<<<dim3(wgs), dim3(AZ_MLP_WAVES * 64), ldsb>>>(md, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<float> ho(wgs * 16);
  CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
  printf("wgs=%d iters=%d wrap=%d: %.1f us total, %.2f us per forward (checksum %.6g)\n", wgs, iters,
         wrap, best * 1e3, best * 1e3 / iters, ho[0] + ho[17]);
#ifdef AZ_STAMP
  long long st[8][64];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(az_stamps), sizeof st));
  for (int w = 0; w < 8; ++w) {
    printf("wave %d deltas:", w);
    for (int i = 1; i < 64 && st[w][i] > 0; ++i) printf(" %lld", st[w][i] - st[w][i - 1]);
    printf("\n");
  }
#endif
  return 0;
}
