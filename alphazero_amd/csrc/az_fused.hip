// az_fused.hip -- the in-tree model: weight packing, network forward, recurrent step and the fused persistent search.
// Part of libaz_mcts.so (gfx950 only); built with -ffp-contract=off, which is part of
// the numerics contract (every fused multiply-add is an explicit az_fmaf, az_math.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#ifdef AZ_STAMP
// Diagnostic build only: cycle stamps of thread 0 of workgroup 0, one row per simulation
// (columns 0-9: the simulation loop's own stamps; 9 + k: stamp k of the path update).
__device__ long long az_sim_stamps[64][16];
#define AZ_WAVE_STAMP(k)                                                            \
  do {                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0 && sim < 64)                            \
      az_sim_stamps[sim][9 + (k)] = __builtin_amdgcn_s_memtime();                   \
  } while (0)
#endif

#include "az_host.h"
#include "az_wave.h"
#include "az_envs.h"
#include "az_mlp.h"

// ---------------------------------------------------------------------------
// model: weight packing, forward, recurrent step
static int az_check_mlp(const az_mlp_params* p, const char* name) {
  if (!p) return az_fail(AZ_ERR_INVALID, "null %s network", name);
  if (p->n_layers < 1 || p->n_layers > AZ_MAX_LAYERS)
    return az_fail(AZ_ERR_UNSUPPORTED, "%s network has %d layers (1..%d supported)", name,
                   p->n_layers, AZ_MAX_LAYERS);
  for (int i = 0; i <= p->n_layers; ++i)
    if (p->sizes[i] < 1 || p->sizes[i] > AZ_MAX_WIDTH)
      return az_fail(AZ_ERR_UNSUPPORTED, "%s layer width %d outside [1,%d]", name, p->sizes[i],
                     AZ_MAX_WIDTH);
  return AZ_OK;
}

// Hidden layers in bfloat16 need input widths that fill whole k-blocks of 32.
static int az_check_bf16(const az_mlp_params* p, const char* name) {
  for (int i = 1; i <= p->n_layers - 2; ++i)
    if (p->sizes[i] % 32)
      return az_fail(AZ_ERR_UNSUPPORTED,
                     "bfloat16 hidden layers need widths that are multiples of 32: %s layer %d has %d inputs",
                     name, i, p->sizes[i]);
  return AZ_OK;
}

static int64_t az_packed_floats(const az_mlp_params* policy, const az_mlp_params* value, int hidden_bf16) {
  if (az_check_mlp(policy, "policy") || az_check_mlp(value, "value")) return -1;
  if (hidden_bf16 && (az_check_bf16(policy, "policy") || az_check_bf16(value, "value"))) return -1;
  MlpLayout lp, lv;
  int64_t off = az_mlp_layout(policy->n_layers, policy->sizes, 0, &lp, hidden_bf16);
  // (behind the weights: the networks' schedule table, az_mlp_table_load)
  return az_mlp_layout(value->n_layers, value->sizes, off, &lv, hidden_bf16) + AZ_TABLE_MAX_WORDS;
}
extern "C" int64_t az_model_packed_floats(const az_mlp_params* policy, const az_mlp_params* value) {
  return az_packed_floats(policy, value, 0);
}
extern "C" int64_t az_model_packed_floats_bf16(const az_mlp_params* policy, const az_mlp_params* value) {
  return az_packed_floats(policy, value, 1);
}

__global__ void k_pack_layer(const float* __restrict__ w, const float* __restrict__ bsrc, int kin,
                             int nout, int kp, int np, float* __restrict__ wp,
                             float* __restrict__ bp) {
  const int64_t total = (int64_t)kp * np;
  const int nkb = kp >> 4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const int64_t blk = e >> 8;  // t * nkb + kb
    const int kb = (int)(blk % nkb), t = (int)(blk / nkb);
    const int k = 16 * kb + 4 * q + (lane >> 4), n = 16 * t + (lane & 15);
    wp[e] = (k < kin && n < nout) ? w[(size_t)k * nout + n] : 0.0f;
  }
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < np; n += gridDim.x * blockDim.x)
    bp[n] = n < nout ? bsrc[n] : 0.0f;
}

// A bfloat16 layer: element ((t * nkb + kb) * 64 + lane) * 8 + j = W[32 kb + 8 (lane >> 4) + j][16 t + (lane & 15)],
// rounded to nearest even; zero padded.  The bias stays float32.
__global__ void k_pack_layer_bf16(const float* __restrict__ w, const float* __restrict__ bsrc, int kin,
                                  int nout, int kp, int np, __bf16* __restrict__ wp, float* __restrict__ bp) {
  const int64_t total = (int64_t)kp * np;
  const int nkb = kp >> 5;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const int64_t blk = e >> 9;  // t * nkb + kb
    const int kb = (int)(blk % nkb), t = (int)(blk / nkb);
    const int k = 32 * kb + 8 * (lane >> 4) + j, n = 16 * t + (lane & 15);
    wp[e] = (__bf16)((k < kin && n < nout) ? w[(size_t)k * nout + n] : 0.0f);
  }
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < np; n += gridDim.x * blockDim.x)
    bp[n] = n < nout ? bsrc[n] : 0.0f;
}

// The parts of a ModelDev that follow from the networks alone: layouts, widest layers, schedule
// sizes, where the schedule table sits behind the weights and what its header must say.  AZ_OK, or
// the reason why the kernels cannot run these networks.
static int az_networks_dev(const az_mlp_params* pp, const az_mlp_params* vp, int hidden_bf16, const float* packed,
                           ModelDev* md) {
  md->bf16 = hidden_bf16 ? 1 : 0;
  int64_t off = az_mlp_layout(pp->n_layers, pp->sizes, 0, &md->pol, md->bf16);
  const int64_t weights_end = az_mlp_layout(vp->n_layers, vp->sizes, off, &md->val, md->bf16);
  int pk = 16, vk = 16;
  for (int i = 0; i < md->pol.n_layers; ++i) pk = md->pol.np[i] > pk ? md->pol.np[i] : pk;
  for (int i = 0; i < md->val.n_layers; ++i) vk = md->val.np[i] > vk ? md->val.np[i] : vk;
  md->pol_kp = pk;
  md->val_kp = vk;
  md->packed = packed;
  md->table = reinterpret_cast<const uint32_t*>(packed + weights_end);
  md->head_floats = 0;
  md->head_floats = az_mlp_head_floats(*md);  // (before anything that depends on the unit phases)
  md->tab_units = az_mlp_units_per_wave(*md) > 0 ? az_mlp_units_per_wave(*md) : 1;
  if (az_mlp_units_per_wave(*md) > AZ_TABLE_UNITS)
    return az_fail(AZ_ERR_UNSUPPORTED,
                   "the networks need %d work units per wave, the schedule table holds %d "
                   "(fewer or narrower layers, or a recurrent_fn in Python through the callback path)",
                   az_mlp_units_per_wave(*md), AZ_TABLE_UNITS);
  if ((size_t)az_mlp_lds_floats(pk, vk, md->tab_units, md->head_floats) * 4 > 160 * 1024)
    return az_fail(AZ_ERR_UNSUPPORTED, "layer widths %d / %d need more than 160 KiB of LDS", pk, vk);
  md->table_hash = az_mlp_table_hash(*md);
  return AZ_OK;
}

// The schedule table of `md`, built by the kernels' own functions in LDS and copied out behind the
// weights (layout: az_mlp.h, above az_mlp_table_load).  One workgroup.
__global__ __launch_bounds__(AZ_MLP_WAVES * 64) void k_table_build(ModelDev md, uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  MlpLds s(lds, md.pol_kp, md.val_kp, md.tab_units);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int words = AZ_MLP_WAVES * md.tab_units * AZ_UNIT_WORDS;
  for (int i = tid; i < words; i += blockDim.x) s.tab[i] = 0u;  // (rows a wave does not use)
  __syncthreads();
  az_mlp_table_build<AZ_MLP_WAVES>(md, s, wave, lane);
  __syncthreads();
  for (int i = tid; i < words; i += blockDim.x) out[AZ_TABLE_HDR_WORDS + i] = s.tab[i];
  for (int i = tid; i < 16; i += blockDim.x) out[4 + i] = (uint32_t)s.tab_n[i];
  if (tid == 0) {
    const uint64_t base = (uint64_t)(uintptr_t)md.packed;
    out[0] = md.table_hash;
    out[1] = (uint32_t)md.tab_units;
    out[2] = (uint32_t)base;
    out[3] = (uint32_t)(base >> 32);
  }
}

static int az_set_lds(const void* fn, size_t bytes);

static int az_pack(const az_mlp_params* policy, const az_mlp_params* value, float* packed, int hidden_bf16,
                   void* stream) {
  if (int rc = az_check_mlp(policy, "policy")) return rc;
  if (int rc = az_check_mlp(value, "value")) return rc;
  if (hidden_bf16) {
    if (int rc = az_check_bf16(policy, "policy")) return rc;
    if (int rc = az_check_bf16(value, "value")) return rc;
  }
  if (!packed) return az_fail(AZ_ERR_INVALID, "null packed buffer");
  MlpLayout L[2];
  int64_t off = az_mlp_layout(policy->n_layers, policy->sizes, 0, &L[0], hidden_bf16);
  az_mlp_layout(value->n_layers, value->sizes, off, &L[1], hidden_bf16);
  const az_mlp_params* P[2] = {policy, value};
  for (int net = 0; net < 2; ++net)
    for (int i = 0; i < L[net].n_layers; ++i) {
      if (!P[net]->w[i] || !P[net]->b[i]) return az_fail(AZ_ERR_INVALID, "null layer weights");
      const int64_t total = (int64_t)L[net].kp[i] * L[net].np[i];
      const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
      if (L[net].wb[i])
        k_pack_layer_bf16<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(
            P[net]->w[i], P[net]->b[i], L[net].kin[i], L[net].nout[i], L[net].kp[i], L[net].np[i],
            reinterpret_cast<__bf16*>(packed + L[net].w_off[i]), packed + L[net].b_off[i]);
      else
        k_pack_layer<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(
            P[net]->w[i], P[net]->b[i], L[net].kin[i], L[net].nout[i], L[net].kp[i], L[net].np[i],
            packed + L[net].w_off[i], packed + L[net].b_off[i]);
      AZ_LAUNCH_CHECK();
    }
  // the schedule table behind the weights; networks the kernels cannot run (they say why when
  // asked to) get a header that matches nothing
  ModelDev md{};
  uint32_t* table = nullptr;
  if (az_networks_dev(policy, value, hidden_bf16, packed, &md) == AZ_OK) {
    table = const_cast<uint32_t*>(md.table);
    const size_t lds = (size_t)az_mlp_lds_floats(md.pol_kp, md.val_kp, md.tab_units, md.head_floats) * 4;
    if (int rc = az_set_lds((const void*)k_table_build, lds)) return rc;
    k_table_build<<<dim3(1), dim3(AZ_MLP_WAVES * 64), lds, (hipStream_t)stream>>>(md, table);
    AZ_LAUNCH_CHECK();
  } else {
    table = const_cast<uint32_t*>(md.table);
    AZ_HIP(hipMemsetAsync(table, 0, AZ_TABLE_HDR_WORDS * 4, (hipStream_t)stream));
  }
  return AZ_OK;
}
extern "C" int az_model_pack(const az_mlp_params* policy, const az_mlp_params* value, float* packed,
                             void* stream) {
  return az_pack(policy, value, packed, 0, stream);
}
extern "C" int az_model_pack_bf16(const az_mlp_params* policy, const az_mlp_params* value, float* packed,
                                  void* stream) {
  return az_pack(policy, value, packed, 1, stream);
}

static int az_model_dev(const az_model* m, ModelDev* md) {
  if (!m) return az_fail(AZ_ERR_INVALID, "null model");
  if (m->env_id < AZ_ENV_CARTPOLE || m->env_id > AZ_ENV_ACROBOT_NOTERM)
    return az_fail(AZ_ERR_UNSUPPORTED, "unknown env_id %d", m->env_id);
  if (!m->packed || !m->norm_mean || !m->norm_std)
    return az_fail(AZ_ERR_INVALID, "model has null buffers");
  az_mlp_params pp{}, vp{};
  pp.n_layers = m->policy_layers;
  vp.n_layers = m->value_layers;
  if (pp.n_layers < 1 || pp.n_layers > AZ_MAX_LAYERS || vp.n_layers < 1 || vp.n_layers > AZ_MAX_LAYERS)
    return az_fail(AZ_ERR_UNSUPPORTED, "layer count outside 1..%d", AZ_MAX_LAYERS);
  memcpy(pp.sizes, m->policy_sizes, sizeof pp.sizes);
  memcpy(vp.sizes, m->value_sizes, sizeof vp.sizes);
  if (int rc = az_check_mlp(&pp, "policy")) return rc;
  if (int rc = az_check_mlp(&vp, "value")) return rc;
  const int od = az_env_obs_dim(m->env_id), na = az_env_num_actions(m->env_id);
  if (m->obs_dim != od || pp.sizes[0] != od || vp.sizes[0] != od)
    return az_fail(AZ_ERR_INVALID, "network input width must equal the observation size %d", od);
  if (m->num_actions != na || pp.sizes[pp.n_layers] != na)
    return az_fail(AZ_ERR_INVALID, "policy output width must equal the %d actions", na);
  if (m->num_atoms != vp.sizes[vp.n_layers] || m->num_atoms > AZ_MAX_ATOMS)
    return az_fail(AZ_ERR_INVALID, "value output width must equal num_atoms (<= %d)", AZ_MAX_ATOMS);
  md->env_id = m->env_id;
  md->gamma = m->gamma;
  md->obs_dim = od;
  md->A = na;
  md->K = m->num_atoms;
  if (m->hidden_bf16) {
    if (int rc = az_check_bf16(&pp, "policy")) return rc;
    if (int rc = az_check_bf16(&vp, "value")) return rc;
  }
  md->mean = m->norm_mean;
  md->std = m->norm_std;
  return az_networks_dev(&pp, &vp, m->hidden_bf16, m->packed, md);
}

// running_statistics.py:183-199 on one observation row, written in the permuted
// activation layout of the first dense layer (both networks read the same rows).
// IN_REGISTERS: `obs` is an EnvOut's array of 6 (only ever indexed with constants, so that it
// stays in registers; entries past obs_dim are computed on and dropped); otherwise `obs` points
// at obs_dim floats in memory.
// OBS_DIM: the observation size when it is known at compile time, else 0 (md.obs_dim).
template <bool IN_REGISTERS, int OBS_DIM = 0>
AZ_DEV void az_store_obs(const ModelDev& md, const MlpLds& s, int row, const float* obs) {
  // all reads first, then the divisions side by side, then the writes: one LDS round trip and
  // one division's latency instead of one of each per element (obs_dim <= 6, EnvOut)
  const int kp = md.val.kp[0];
  const int n = OBS_DIM ? OBS_DIM : md.obs_dim;
  constexpr int N = OBS_DIM ? OBS_DIM : 6;
  float mean[N], dev[N], x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    mean[i] = s.norm_mean[i];  // (the normaliser's LDS rows are longer than 6)
    dev[i] = s.norm_std[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = ((IN_REGISTERS ? obs[i] : obs[i < n ? i : 0]) - mean[i]) / dev[i];
  // (pinned: the quotients are all computed here, not one by one under the conditions below)
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(x[i]));
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) s.in0[row * AZ_IN0_STRIDE + az_act_index(i, kp)] = x[i];
}

// Kernel prologue: zeroes the first layer's input rows and copies the normaliser into
// LDS (the caller places a workgroup barrier before the first az_store_obs).
AZ_DEV void az_stage_inputs(const ModelDev& md, const MlpLds& s, int tid, int nthreads) {
  for (int i = tid; i < 16 * AZ_IN0_STRIDE; i += nthreads) s.in0[i] = 0.0f;
  for (int i = tid; i < md.obs_dim; i += nthreads) {
    s.norm_mean[i] = md.mean[i];
    s.norm_std[i] = md.std[i];
  }
  az_mlp_stage_heads(md, s, tid, nthreads);
}

#ifdef AZ_STAMP
// Diagnostic build only: kernel-entry stamps of workgroup 0 in the last columns of the forward stamps.
#define AZ_FWD_EDGE(k)                                                                          \
  do {                                                                                          \
    if (blockIdx.x == 0 && lane == 0) az_stamps[wave][k] = __builtin_amdgcn_s_memtime();        \
  } while (0)
#else
#define AZ_FWD_EDGE(k) \
  do {                 \
  } while (0)
#endif
template <bool BF16>
__global__ __launch_bounds__(AZ_MLP_WAVES * 64) void k_forward(ModelDev md, int B,
                                                               const float* __restrict__ obs,
                                                               float* __restrict__ logits,
                                                               float* __restrict__ value) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  MlpLds s(lds, md.pol_kp, md.val_kp, md.tab_units);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  AZ_FWD_EDGE(60);
  az_stage_inputs(md, s, tid, blockDim.x);
  AZ_FWD_EDGE(61);
  az_mlp_table_setup<AZ_MLP_WAVES>(md, s, tid, blockDim.x, wave, lane);
  __syncthreads();
  AZ_FWD_EDGE(62);
  const int b = blockIdx.x * 16 + tid;
  if (tid < 16 && b < B) az_store_obs<false>(md, s, tid, obs + (size_t)b * md.obs_dim);
  __syncthreads();
  MlpRegs regs;
  MlpFirst first;
  az_forward_prime<AZ_MLP_WAVES, BF16 ? AZ_SLOTS_BF16 : AZ_SLOTS>(md, s, wave, lane, regs, first, true);
  AZ_FWD_EDGE(63);
  az_forward_tile<AZ_MLP_WAVES, false, BF16>(md, s, wave, lane, regs, first);
  if (tid < 16 && b < B) {
    for (int a = 0; a < md.A; ++a) logits[(size_t)b * md.A + a] = s.logits[tid * AZ_MAX_ACTIONS + a];
    for (int k = 0; k < md.K; ++k) value[(size_t)b * md.K + k] = s.value[tid * AZ_MAX_ATOMS + k];
  }
}

static int az_set_lds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024)
    AZ_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return AZ_OK;
}

extern "C" int az_forward(const az_model* m, int32_t B, const float* obs, float* logits,
                          float* value, void* stream) {
  ModelDev md;
  if (int rc = az_model_dev(m, &md)) return rc;
  if (B < 1 || !obs || !logits || !value) return az_fail(AZ_ERR_INVALID, "bad forward arguments");
  const size_t lds = (size_t)az_mlp_lds_floats(md.pol_kp, md.val_kp, md.tab_units, md.head_floats) * 4;
  auto kern = md.bf16 ? k_forward<true> : k_forward<false>;
  if (int rc = az_set_lds((const void*)kern, lds)) return rc;
  kern<<<dim3((B + 15) / 16), dim3(AZ_MLP_WAVES * 64), lds, (hipStream_t)stream>>>(md, B, obs, logits, value);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// alphazero.py:317-339 on one lane: environment step, reward and discount.
// A_EXACT: the model's action count when it is known at compile time, else 0.
template <int A_EXACT = 0, bool QUAD = false>
AZ_DEV void az_model_env_step(const ModelDev& md, EnvState& st, int action, EnvOut& o, float& reward,
                              float& discount) {
  az_env_step_for_actions<A_EXACT, QUAD>(md.env_id, st, action, o);
  reward = o.reward;
  discount = md.gamma * (o.truncated != 0.0f ? 1.0f : 1.0f - o.terminated);
}

template <bool BF16>
__global__ __launch_bounds__(AZ_MLP_WAVES * 64) void k_recurrent_step(
    ModelDev md, int B, float* __restrict__ state, int32_t* __restrict__ time,
    const int32_t* __restrict__ action, float* __restrict__ reward, float* __restrict__ discount,
    float* __restrict__ logits, float* __restrict__ value) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  MlpLds s(lds, md.pol_kp, md.val_kp, md.tab_units);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  az_stage_inputs(md, s, tid, blockDim.x);
  az_mlp_table_setup<AZ_MLP_WAVES>(md, s, tid, blockDim.x, wave, lane);
  __syncthreads();
  const int b = blockIdx.x * 16 + tid;
  if (tid < 16 && b < B) {
    EnvState st;
    EnvOut o{};
    for (int i = 0; i < 4; ++i) st.s[i] = state[(size_t)b * 4 + i];
    st.time = time[b];
    float r, dsc;
    az_model_env_step(md, st, action[b], o, r, dsc);
    for (int i = 0; i < 4; ++i) state[(size_t)b * 4 + i] = st.s[i];
    time[b] = st.time;
    reward[b] = r;
    discount[b] = dsc;
    az_store_obs<true>(md, s, tid, o.obs);
  }
  __syncthreads();
  MlpRegs regs;
  MlpFirst first;
  az_forward_prime<AZ_MLP_WAVES, BF16 ? AZ_SLOTS_BF16 : AZ_SLOTS>(md, s, wave, lane, regs, first, true);
  az_forward_tile<AZ_MLP_WAVES, false, BF16>(md, s, wave, lane, regs, first);
  if (tid < 16 && b < B) {
    for (int a = 0; a < md.A; ++a) logits[(size_t)b * md.A + a] = s.logits[tid * AZ_MAX_ACTIONS + a];
    for (int k = 0; k < md.K; ++k) value[(size_t)b * md.K + k] = s.value[tid * AZ_MAX_ATOMS + k];
  }
}

extern "C" int az_recurrent_step(const az_model* m, int32_t B, float* state, int32_t* time,
                                 const int32_t* action, float* reward, float* discount,
                                 float* logits, float* value, void* stream) {
  ModelDev md;
  if (int rc = az_model_dev(m, &md)) return rc;
  if (B < 1 || !state || !time || !action || !reward || !discount || !logits || !value)
    return az_fail(AZ_ERR_INVALID, "bad recurrent_step arguments");
  const size_t lds = (size_t)az_mlp_lds_floats(md.pol_kp, md.val_kp, md.tab_units, md.head_floats) * 4;
  auto kern = md.bf16 ? k_recurrent_step<true> : k_recurrent_step<false>;
  if (int rc = az_set_lds((const void*)kern, lds)) return rc;
  kern<<<dim3((B + 15) / 16), dim3(AZ_MLP_WAVES * 64), lds, (hipStream_t)stream>>>(
      md, B, state, time, action, reward, discount, logits, value);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// ---------------------------------------------------------------------------
// the fused persistent search: 16 lanes per workgroup; per simulation the first 16
// threads select + step the environment, all waves run the two MLPs on the matrix
// cores, then the 16 threads expand and back up.
template <int AMAX, int KMAX>
__global__ __launch_bounds__(AZ_MLP_WAVES * 64) void k_search_fused(SearchCtx c, az_tree t,
                                                                    ModelDev md,
                                                                    unsigned long long* sum_depth) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const az_search_desc& d = c.d;
  MlpLds s(lds, md.pol_kp, md.val_kp, md.tab_units);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int b = blockIdx.x * 16 + tid;
  const bool active = tid < 16 && b < d.B;
  Lane<AMAX, KMAX> l(t, active ? b : 0, d.S + 1, d.A, d.K);
  az_stage_inputs(md, s, tid, blockDim.x);
  az_mlp_table_setup<AZ_MLP_WAVES>(md, s, tid, blockDim.x, wave, lane);
  __syncthreads();
  MlpRegs regs;
  MlpFirst first;
  az_forward_prime<AZ_MLP_WAVES>(md, s, wave, lane, regs, first, false);
  const bool need_keys = d.variant != AZ_VARIANT_GUMBEL && d.noise_mode == AZ_NOISE_THREEFRY;
  uint32_t r0 = d.key[0], r1 = d.key[1];
  long long depth_sum = 0;
  for (int sim = 0; sim < d.S; ++sim) {
    int parent = 0, action = 0, node = 0;
    float reward = 0.0f, discount = 0.0f;
    EnvState st;
    if (active) {
      uint32_t k0 = 0, k1 = 0;
      if (need_keys) {
        uint32_t s0, s1;
        az_next_sim_key(r0, r1, s0, s1);
        az_key_split(s0, s1, (uint32_t)b, &k0, &k1);
      }
      int nxt, depth;
      az_simulate_lane<AMAX, KMAX>(c, l, b, sim, k0, k1, parent, action, nxt, depth);
      node = nxt == AZ_UNVISITED ? sim + 1 : nxt;
      depth_sum += depth;
#pragma unroll
      for (int i = 0; i < 4; ++i) st.s[i] = l.es[(size_t)parent * 4 + i];
      st.time = l.et[parent];
      EnvOut o{};
      az_model_env_step(md, st, action, o, reward, discount);
      az_store_obs<true>(md, s, tid, o.obs);
    }
    __syncthreads();
    az_forward_tile<AZ_MLP_WAVES, true>(md, s, wave, lane, regs, first);
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) l.es[(size_t)node * 4 + i] = st.s[i];
      l.et[node] = st.time;
      az_expand_write<AMAX, KMAX>(l, parent, action, node, s.logits + tid * AZ_MAX_ACTIONS,
                                  s.value + tid * AZ_MAX_ATOMS, reward, discount);
      az_backward_lane<AMAX, KMAX>(l, node);
    }
  }
  if (tid < 64) az_add_depth(sum_depth, depth_sum);
}


#ifdef AZ_STAMP
#define AZ_SIM_STAMP(k)                                                                        \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (blockIdx.x == 0 && tid == 0 && sim < 64) az_sim_stamps[sim][k] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#else
#define AZ_SIM_STAMP(k) \
  do {                  \
  } while (0)
#endif

// A copy of `v` the compiler cannot trace back: whatever is derived from it (here the
// per-lane base addresses of the tree arrays) is recomputed where it is used instead of
// being hoisted out of the simulation loop and kept in registers across the MLP phase,
// whose weight slots need the registers.
AZ_DEV int az_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// The fused search in the cooperative form (az_wave.h): all threads are tree threads (16 lanes
// x 32 path positions, two lanes per wave) and all AZ_MLP_WAVES waves run the networks.  Per
// simulation: every wave walks its two lanes' trees and steps their environments, one
// workgroup barrier hands the 16 observations to the networks, and after the networks' last
// phase every wave updates its own two lanes -- nothing else in the loop crosses a wave.
// EXACT: A == AMAX and K == KMAX (compile-time shapes); RULE: see az_rule_ctx.
template <int AMAX, int KMAX, bool EXACT, int RULE, bool NOISY, bool BF16 = false>
__global__ __launch_bounds__(AZ_MLP_WAVES * 64) void k_search_fused_wave(
    SearchCtx c_in, az_tree t, ModelDev md, int emb_in_lds, unsigned long long* sum_depth) {
  static_assert(AZ_MLP_WAVES * 64 == AZ_TREE_THREADS, "one thread layout for both phases");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const SearchCtx c = az_rule_ctx<RULE>(c_in);
  const az_search_desc& d = c.d;
  const int A = EXACT ? AMAX : d.A, K = EXACT ? KMAX : d.K, S = d.S;
  MlpLds s(lds, md.pol_kp, md.val_kp, md.tab_units);
  float* tree_lds = lds + az_mlp_lds_floats(md.pol_kp, md.val_kp, md.tab_units, md.head_floats);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int ln = tid / AZ_POS, j = tid & (AZ_POS - 1);
  const int b = blockIdx.x * AZ_WG_LANES + ln;
  const bool active = b < d.B;
  WaveLds ws(tree_lds, ln, S);
  // With room in LDS, the lane's environment states live there too ([N][5] words: the
  // state and the step counter); global memory keeps a write-through copy for the caller.
  uint32_t* sim_keys = reinterpret_cast<uint32_t*>(tree_lds + AZ_WG_LANES * WaveLds::lane_words(S));
  float* const emb_base = reinterpret_cast<float*>(sim_keys + az_key_words(S));
  float* emb = emb_in_lds ? emb_base + ln * (S + 1) * 5 : nullptr;
  const bool need_keys = NOISY && d.variant != AZ_VARIANT_GUMBEL && d.noise_mode == AZ_NOISE_THREEFRY;
  if (need_keys) az_wave_sim_keys(d, sim_keys, tid, blockDim.x, c.in.sim_keys);
  az_stage_inputs(md, s, tid, blockDim.x);
  if (j == 0) ws.selx[0] = az_i32x4{0, 0, 0, 0};
  {
    Lane<AMAX, KMAX> l(t, active ? b : 0, S + 1, A, K);
    az_wave_init<AMAX, KMAX, NOISY>(c, l, ws, b, j, active);
  }
  if (emb && active) {
    // all of the caller's states, not only the root's: slots that no simulation expands (a
    // depth limit can leave some) are written back unchanged after the last simulation
    Lane<AMAX, KMAX> l(t, b, S + 1, A, K);
    for (int node = j; node <= S; node += AZ_POS) {
      const az_f32x4 st4 = *reinterpret_cast<const az_f32x4*>(l.es + (size_t)node * 4);
      emb[node * 5] = st4.x;
      emb[node * 5 + 1] = st4.y;
      emb[node * 5 + 2] = st4.z;
      emb[node * 5 + 3] = st4.w;
      emb[node * 5 + 4] = az_u2f((uint32_t)l.et[node]);
    }
  }
  az_mlp_table_setup<AZ_MLP_WAVES>(md, s, tid, blockDim.x, wave, lane);
  if (j == 0) ws.hdr[6] = 0;  // the lane's sum of path depths (in LDS: one register fewer across the networks)
  __syncthreads();
  for (int sim = 0; sim < S; ++sim) {
    AZ_SIM_STAMP(0);
    MlpRegs regs;
#ifndef AZ_PREFETCH_MAX_WORDS
#define AZ_PREFETCH_MAX_WORDS 34
#endif
    constexpr bool kPrefetch = sizeof(WaveRow<AMAX, KMAX>) <= AZ_PREFETCH_MAX_WORDS * 4;
    WaveRow<AMAX, KMAX> P0;
    WavePath wp;
    {
      // Everything that depends on the thread only is derived again here, and once more
      // before the update, from a value the optimiser cannot see through: hoisted out of
      // the loop, those indices and addresses would sit in registers across the networks,
      // where there are none to spare (a spilled one costs a reload and, worse, a wait for
      // every vector-memory operation in flight).
      const int tid = az_opaque((int)threadIdx.x), lane = tid & 63;
      const int ln = tid / AZ_POS, h = ln & 1, j = tid & (AZ_POS - 1);
      const int b = blockIdx.x * AZ_WG_LANES + ln;
      const bool active = b < d.B, leader = active && j == 0;
      const WaveLds ws(tree_lds, ln, S);
      // The first unit's weights are requested first of all: the eight waves' requests
      // (288 KB) keep the CU's memory pipeline busy for about as long as the walk, the
      // environment step and the observation take, and none of those needs many registers
      // or waits for vector memory.  The weight registers are not kept across the update at
      // the end of the loop body, which needs them.
      // (in three portions, so that no wave sits in the queue behind 36 loads of each of
      // the other seven before it can start its walk)
      MlpFirst first;
      const MlpPre first_unit = az_forward_prime_begin<AZ_MLP_WAVES>(s, wave, lane, first, true);
      constexpr int kSlots = BF16 ? AZ_SLOTS_BF16 : AZ_SLOTS;
      az_forward_prime_slots<0, 6, kSlots>(first_unit, lane, regs);
      Lane<AMAX, KMAX> l(t, active ? b : 0, S + 1, A, K);
      if (NOISY)
        wp = az_wave_chase_flagged<AMAX, KMAX>(c, l, ws, b, h, j, sim, d.max_depth, active,
                                               need_keys ? sim_keys : nullptr, [] {});
      else
        wp = az_wave_chase_lean(ws, h, j, sim, d.max_depth, active);
      AZ_SIM_STAMP(1);
      az_forward_prime_slots<6, 11, kSlots>(first_unit, lane, regs);
      // Three actions (MountainCar, Acrobot): the lane's first FOUR threads step the environment
      // together -- Acrobot's step is 21 sines and cosines, four at a time with four threads
      // (az_trig4) -- and the first of them alone writes.
      constexpr bool kQuadEnv = EXACT && AMAX == 3;
      if (kQuadEnv ? (active && j < 4) : leader) {
        if (leader)
          __hip_atomic_fetch_add(&ws.hdr[6], wp.dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // (one LDS add, nothing returned)
        const WaveLeaf wl = az_wave_leaf(ws, wp.dep, sim);
        EnvState st;
        float* const emb = emb_in_lds ? emb_base + ln * (S + 1) * 5 : nullptr;
        if (emb) {
#pragma unroll
          for (int i = 0; i < 4; ++i) st.s[i] = emb[wl.parent * 5 + i];
          st.time = (int32_t)az_f2u(emb[wl.parent * 5 + 4]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) st.s[i] = l.es[(size_t)wl.parent * 4 + i];
          st.time = l.et[wl.parent];
          // waited for in here: left to the compiler, the wait lands behind the join, where it
          // also holds up the other branch until the weights above have arrived
          __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        }
        EnvOut o{};
        float reward, discount;
        AZ_SIM_STAMP(6);
        az_model_env_step<EXACT ? AMAX : 0, kQuadEnv>(md, st, wl.action, o, reward, discount);
        AZ_SIM_STAMP(7);
        if (leader) {
          if (emb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) emb[wl.node * 5 + i] = st.s[i];
            emb[wl.node * 5 + 4] = az_u2f((uint32_t)st.time);
          }
          ws.hdr[4] = (int32_t)az_f2u(reward);
          ws.hdr[5] = (int32_t)az_f2u(discount);
          az_store_obs<true, (EXACT && AMAX == 2) ? 4 : 0>(md, s, ln, o.obs);  // two actions: CartPole
          // the caller's copy of the new node's environment state (with the states in LDS it is
          // written once, after the last simulation)
          if (!emb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) l.es[(size_t)wl.node * 4 + i] = st.s[i];
            l.et[wl.node] = st.time;
          }
        }
        AZ_SIM_STAMP(8);
      }
      az_forward_prime_slots<11, AZ_SLOTS, kSlots>(first_unit, lane, regs);
      // The path is known now, and the rows the update will need depend on the path only: when a
      // row fits in a few registers its loads are issued before the networks run, so that
      // their latency hides behind the MLPs.
      if (kPrefetch) az_wave_load<AMAX, KMAX>(l, active, j, wp.dep, sim, wp.e_prev, wp.e_mine, P0);
      AZ_SIM_STAMP(9);
      AZ_SIM_STAMP(2);
      __syncthreads();
      AZ_SIM_STAMP(3);
      az_forward_tile<AZ_MLP_WAVES, false, BF16>(md, s, wave, lane, regs, first);
      // Nothing is in flight any more (the row prefetch returned long ago), and saying so
      // here costs nothing.  Left unsaid, the compiler has to assume on some paths that the
      // prefetch is still pending when its registers are reused in the next simulation, and
      // waits there -- behind that simulation's 36 weight loads.
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    }
    AZ_SIM_STAMP(4);
    {
      const int tid = az_opaque((int)threadIdx.x);
      const int ln = tid / AZ_POS, h = ln & 1, j = tid & (AZ_POS - 1);
      const int b = blockIdx.x * AZ_WG_LANES + ln;
      const bool active = b < d.B;
      const WaveLds ws(tree_lds, ln, S);
      Lane<AMAX, KMAX> l(t, active ? b : 0, S + 1, A, K);
      const float reward = az_u2f((uint32_t)ws.hdr[4]), discount = az_u2f((uint32_t)ws.hdr[5]);
      az_wave_update<AMAX, KMAX, NOISY, RULE == 1>(c, l, ws, active, h, j, sim, wp, s.logits + ln * AZ_MAX_ACTIONS,
                                        s.value + ln * AZ_MAX_ATOMS, reward, discount,
                                        kPrefetch ? &P0 : nullptr);
      AZ_SIM_STAMP(5);
    }
  }
  {
    // (thread indices derived afresh, as in the loop: nothing of the prologue stays live)
    const int tid = az_opaque((int)threadIdx.x);
    const int ln = tid / AZ_POS, j = tid & (AZ_POS - 1);
    const int b = blockIdx.x * AZ_WG_LANES + ln;
    const bool active = b < d.B;
    if (emb_in_lds && active) {
      // (a simulation that ends on the depth limit steps an existing node's state into the
      // slot it already has -- az_wave_leaf -- so the LDS copy is the final one either way)
      Lane<AMAX, KMAX> l(t, b, S + 1, A, K);
      const float* e = emb_base + ln * (S + 1) * 5;
      for (int node = 1 + j; node <= S; node += AZ_POS) {
        az_f32x4 st4 = {e[node * 5], e[node * 5 + 1], e[node * 5 + 2], e[node * 5 + 3]};
        *reinterpret_cast<az_f32x4*>(l.es + (size_t)node * 4) = st4;
        l.et[node] = (int32_t)az_f2u(e[node * 5 + 4]);
      }
    }
    const WaveLds ws(tree_lds, ln, S);
    az_add_depth(sum_depth, (long long)(active && j == 0 ? ws.hdr[6] : 0));
  }
}

extern "C" int az_search_fused(const az_search_desc* d, const az_tree* t,
                               const az_search_inputs* in, const az_model* m, int64_t* sum_depth,
                               void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (int rc = az_check_inputs(d, in)) return rc;
  ModelDev md;
  if (int rc = az_model_dev(m, &md)) return rc;
  if (!t->emb_state || !t->emb_time)
    return az_fail(AZ_ERR_INVALID, "fused search needs the tree's emb_state / emb_time arrays");
  if (d->variant == AZ_VARIANT_STOCHASTIC)
    return az_fail(AZ_ERR_UNSUPPORTED, "the stochastic variant runs on az_simulate / az_expand_backward only");
  if (d->A != md.A || d->K != md.K)
    return az_fail(AZ_ERR_INVALID, "search A=%d K=%d do not match the model's A=%d K=%d", d->A,
                   d->K, md.A, md.K);
  const size_t lds = (size_t)az_mlp_lds_floats(md.pol_kp, md.val_kp, md.tab_units, md.head_floats) * 4;
  SearchCtx c{*d, *in};
  if (az_coop_applicable(d, lds, false)) {
    size_t lds2 = lds + ((size_t)AZ_WG_LANES * WaveLds::lane_words(d->S) + (size_t)az_key_words(d->S)) * 4;
    const size_t emb_bytes = (size_t)AZ_WG_LANES * (d->S + 1) * 5 * 4;
    const int emb_in_lds = lds2 + emb_bytes <= 160 * 1024;
    if (emb_in_lds) lds2 += emb_bytes;
    const bool noisy = d->variant != AZ_VARIANT_GUMBEL && d->noise_mode != AZ_NOISE_ZERO;
    const int rule = az_rule_of(d);
    if (md.bf16) {
      // bfloat16 hidden layers: the instantiations of the built-in environments' shapes with a
      // scalar value head, under the rules the three policies use by default
      using Kern = void (*)(SearchCtx, az_tree, ModelDev, int, unsigned long long*);
      Kern kern = nullptr;
      if (d->K == 1 && d->A == 2) {
        if (noisy && rule == 2) kern = k_search_fused_wave<2, 1, true, 2, true, true>;
        if (!noisy && rule == 1) kern = k_search_fused_wave<2, 1, true, 1, false, true>;
      } else if (d->K == 1 && d->A == 3) {
        if (noisy && rule == 2) kern = k_search_fused_wave<3, 1, true, 2, true, true>;
        if (!noisy && rule == 1) kern = k_search_fused_wave<3, 1, true, 1, false, true>;
      }
      if (!kern)
        return az_fail(AZ_ERR_UNSUPPORTED,
                       "bfloat16 hidden layers: fused search built for 2 or 3 actions, one value atom and the "
                       "policies' default selection rules only");
      if (int rc = az_set_lds((const void*)kern, lds2)) return rc;
      kern<<<dim3((d->B + AZ_WG_LANES - 1) / AZ_WG_LANES), dim3(AZ_MLP_WAVES * 64), lds2,
             (hipStream_t)stream>>>(c, *t, md, emb_in_lds, reinterpret_cast<unsigned long long*>(sum_depth));
      AZ_LAUNCH_CHECK();
      return AZ_OK;
    }
    return az_dispatch_coop(d->A, d->K, [&]<int AMAX, int KMAX, bool EXACT>() -> int {
      using Kern = void (*)(SearchCtx, az_tree, ModelDev, int, unsigned long long*);
      Kern kern;
      if (noisy)
        kern = EXACT && rule == 2 ? k_search_fused_wave<AMAX, KMAX, EXACT, EXACT ? 2 : 0, true>
                                  : k_search_fused_wave<AMAX, KMAX, EXACT, 0, true>;
      else
        kern = EXACT && rule == 1 ? k_search_fused_wave<AMAX, KMAX, EXACT, EXACT ? 1 : 0, false>
                                  : k_search_fused_wave<AMAX, KMAX, EXACT, 0, false>;
      if (int rc = az_set_lds((const void*)kern, lds2)) return rc;
      kern<<<dim3((d->B + AZ_WG_LANES - 1) / AZ_WG_LANES), dim3(AZ_MLP_WAVES * 64), lds2,
             (hipStream_t)stream>>>(c, *t, md, emb_in_lds,
                                    reinterpret_cast<unsigned long long*>(sum_depth));
      AZ_LAUNCH_CHECK();
      return AZ_OK;
    });
  }
  if (md.bf16)
    return az_fail(AZ_ERR_UNSUPPORTED, "bfloat16 hidden layers: the one-thread-per-lane fused search is float32 only");
  return az_dispatch_ak(d->A, d->K, [&]<int AMAX, int KMAX>() -> int {
    if (int rc = az_set_lds((const void*)k_search_fused<AMAX, KMAX>, lds)) return rc;
    k_search_fused<AMAX, KMAX>
        <<<dim3((d->B + 15) / 16), dim3(AZ_MLP_WAVES * 64), lds, (hipStream_t)stream>>>(
            c, *t, md, reinterpret_cast<unsigned long long*>(sum_depth));
    AZ_LAUNCH_CHECK();
    return AZ_OK;
  });
}


#ifdef AZ_STAMP
// Diagnostic build only: copies the cycle stamps of the last forward to the host.
extern "C" int az_debug_read_stamps(long long* host_out) {
  AZ_HIP(hipDeviceSynchronize());
  AZ_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(az_stamps), sizeof(long long) * 8 * 64));
  return AZ_OK;
}
extern "C" int az_debug_read_flag_counts(unsigned long long* host_out, int reset) {
  AZ_HIP(hipDeviceSynchronize());
  AZ_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(az_dbg_flag_counts), sizeof(unsigned long long) * 4));
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    AZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(az_dbg_flag_counts), z, sizeof(z)));
  }
  return AZ_OK;
}
extern "C" int az_debug_read_sim_stamps(long long* host_out) {
  AZ_HIP(hipDeviceSynchronize());
  AZ_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(az_sim_stamps), sizeof(long long) * 64 * 16));
  return AZ_OK;
}
#endif
