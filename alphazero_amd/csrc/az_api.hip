// az_api.hip -- error channel, sequential-halving table, policy prologues/epilogues, environments, numerics probes.
// Part of libaz_mcts.so (gfx950 only); built with -ffp-contract=off, which is part of
// the numerics contract (every fused multiply-add is an explicit az_fmaf, az_math.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "az_host.h"
#include "az_envs.h"
#include "az_swish.h"

// ---------------------------------------------------------------------------
// error channel
static thread_local std::string g_err;

int az_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int az_check_desc(const az_search_desc* d, const az_tree* t) {
  if (!d || !t) return az_fail(AZ_ERR_INVALID, "null descriptor or tree");
  if (d->B < 1 || d->S < 1) return az_fail(AZ_ERR_INVALID, "B=%d S=%d must be >= 1", d->B, d->S);
  if (d->A < 1 || d->A > AZ_MAX_ACTIONS)
    return az_fail(AZ_ERR_UNSUPPORTED, "num_actions=%d outside [1,%d]", d->A, AZ_MAX_ACTIONS);
  if (d->K < 1 || d->K > AZ_MAX_ATOMS)
    return az_fail(AZ_ERR_UNSUPPORTED, "num_atoms=%d outside [1,%d]", d->K, AZ_MAX_ATOMS);
  if (d->variant < 0 || d->variant > 3) return az_fail(AZ_ERR_INVALID, "bad variant %d", d->variant);
  if (d->variant == AZ_VARIANT_STOCHASTIC && (d->num_decision_actions < 1 || d->num_decision_actions >= d->A))
    return az_fail(AZ_ERR_INVALID, "num_decision_actions=%d must lie in [1, A=%d)", d->num_decision_actions, d->A);
  if (d->qtransform < 0 || d->qtransform > 2)
    return az_fail(AZ_ERR_INVALID, "bad qtransform %d", d->qtransform);
  if (d->max_depth < 1) return az_fail(AZ_ERR_INVALID, "max_depth=%d must be >= 1", d->max_depth);
  // (the kernels address a lane's rows by 32-bit byte offsets: no array may reach 4 GiB; the
  // widest ones are children_values [B,N,A,K] and the [B,N,4] environment states)
  const int64_t widest = (int64_t)d->A * d->K > 4 ? (int64_t)d->A * d->K : 4;
  if ((int64_t)d->B * (d->S + 1) * widest * 4 >= (int64_t)1 << 32)
    return az_fail(AZ_ERR_UNSUPPORTED, "tree too large: an array of B*(S+1)*%lld words reaches 4 GiB", (long long)widest);
  if (!t->node_visits || !t->raw_values || !t->node_values || !t->parents ||
      !t->action_from_parent || !t->children_index || !t->children_prior_logits ||
      !t->children_visits || !t->children_rewards || !t->children_discounts || !t->children_values)
    return az_fail(AZ_ERR_INVALID, "tree has a null array");
  return AZ_OK;
}

int az_check_inputs(const az_search_desc* d, const az_search_inputs* in) {
  if (!in) return az_fail(AZ_ERR_INVALID, "null search inputs");
  if (d->variant == AZ_VARIANT_STOCHASTIC && !in->is_decision)
    return az_fail(AZ_ERR_INVALID, "stochastic search needs the is_decision array");
  if (d->variant == AZ_VARIANT_GUMBEL) {
    if (!in->root_gumbel || !in->table)
      return az_fail(AZ_ERR_INVALID, "gumbel search needs root_gumbel and the considered-visits table");
    if (d->max_num_considered < 0) return az_fail(AZ_ERR_INVALID, "max_num_considered < 0");
  } else if (d->noise_mode == AZ_NOISE_INJECTED) {
    if (!in->noise || d->noise_depth < 1)
      return az_fail(AZ_ERR_INVALID, "injected tie-break noise needs a tensor and noise_depth >= 1");
  } else if (d->noise_mode != AZ_NOISE_THREEFRY && d->noise_mode != AZ_NOISE_ZERO) {
    return az_fail(AZ_ERR_INVALID, "bad noise_mode %d", d->noise_mode);
  }
  return AZ_OK;
}

extern "C" const char* az_last_error(void) { return g_err.c_str(); }
extern "C" const char* az_build_arch(void) { return "gfx950"; }
extern "C" int az_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------
// sequential-halving table (host)
static void az_considered_sequence(int m, int S, int32_t* seq) {
  if (m <= 1) {
    for (int i = 0; i < S; ++i) seq[i] = i;
    return;
  }
  int log2max = 0;
  while ((1 << log2max) < m) ++log2max;
  std::vector<int32_t> visits(m, 0);
  int n = 0, considered = m;
  while (n < S) {
    int extra = S / (log2max * considered);
    if (extra < 1) extra = 1;
    for (int e = 0; e < extra; ++e) {
      for (int i = 0; i < considered && n < S; ++i) seq[n++] = visits[i];
      for (int i = 0; i < considered; ++i) visits[i] += 1;
    }
    considered = considered / 2 < 2 ? 2 : considered / 2;
  }
}

extern "C" int az_considered_table(int32_t max_m, int32_t S, int32_t* table_host) {
  if (max_m < 0 || S < 1 || !table_host) return az_fail(AZ_ERR_INVALID, "bad table arguments");
  for (int m = 0; m <= max_m; ++m) az_considered_sequence(m, S, table_host + (size_t)m * S);
  return AZ_OK;
}

extern "C" int az_sim_keys(const uint32_t* key, int32_t S, uint32_t* keys_host) {
  if (!key || S < 1 || !keys_host) return az_fail(AZ_ERR_INVALID, "bad sim_keys arguments");
  // the same two splits per simulation as az_wave_sim_keys (az_wave.h), on the host
  uint32_t r0 = key[0], r1 = key[1];
  for (int i = 0; i < S; ++i) {
    az_key_split(r0, r1, 1u, &keys_host[2 * i], &keys_host[2 * i + 1]);
    uint32_t n0, n1;
    az_key_split(r0, r1, 0u, &n0, &n1);
    r0 = n0;
    r1 = n1;
  }
  return AZ_OK;
}

// ---------------------------------------------------------------------------
// policy prologues / epilogues: one thread per lane, run-time A
AZ_DEV void az_softmax_rt(const float* x, int A, float* out) {
  float m = x[0];
  for (int a = 1; a < A; ++a) m = az_maxf(m, x[a]);
  float s = 0.0f;
  for (int a = 0; a < A; ++a) {
    out[a] = az_expf(x[a] - m);
    s = s + out[a];
  }
  for (int a = 0; a < A; ++a) out[a] = out[a] / s;
}

AZ_DEV int az_argmax_rt(const float* x, int A) {
  int best = 0;
  float bv = x[0];
  for (int a = 1; a < A; ++a) {
    float v = x[a];
    if ((v > bv) || (v != v && bv == bv)) {
      best = a;
      bv = v;
    }
  }
  return best;
}

// mctx_dist/_src/policies.py:486-496
AZ_DEV void az_mask_invalid_rt(float* logits, const float* invalid, int A) {
  if (!invalid) return;
  float mx = logits[0];
  for (int a = 1; a < A; ++a) mx = az_maxf(mx, logits[a]);
  for (int a = 0; a < A; ++a) {
    logits[a] = logits[a] - mx;
    if (invalid[a] != 0.0f) logits[a] = -AZ_FLT_MAX;
  }
}

// mctx_dist/_src/policies.py:519-524
AZ_DEV void az_apply_temperature_rt(float* logits, float temperature, int A) {
  float mx = logits[0];
  for (int a = 1; a < A; ++a) mx = az_maxf(mx, logits[a]);
  float den = az_maxf(AZ_FLT_TINY, temperature);
  for (int a = 0; a < A; ++a) logits[a] = (logits[a] - mx) / den;
}

// mctx_dist/_src/policies.py:80-90, 499-516
AZ_DEV void az_noisy_root_logits(const float* prior_logits, const float* dirichlet, float fraction,
                                 const float* invalid, int A, float* out) {
  float p[AZ_MAX_ACTIONS];
  az_softmax_rt(prior_logits, A, p);
  for (int a = 0; a < A; ++a) {
    float noisy = (1.0f - fraction) * p[a] + fraction * dirichlet[a];
    out[a] = az_logf(az_maxf(noisy, AZ_FLT_TINY));
  }
  az_mask_invalid_rt(out, invalid, A);
}

__global__ void k_root_muzero(int B, int A, const float* __restrict__ prior_logits,
                              const float* __restrict__ dirichlet, float fraction,
                              const float* __restrict__ invalid, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float o[AZ_MAX_ACTIONS];
  az_noisy_root_logits(prior_logits + (size_t)b * A, dirichlet + (size_t)b * A, fraction,
                       invalid ? invalid + (size_t)b * A : nullptr, A, o);
  for (int a = 0; a < A; ++a) out[(size_t)b * A + a] = o[a];
}

extern "C" int az_root_muzero(int32_t B, int32_t A, const float* prior_logits,
                              const float* dirichlet, float dirichlet_fraction,
                              const float* invalid, float* out_logits, void* stream) {
  if (B < 1 || A < 1 || A > AZ_MAX_ACTIONS || !prior_logits || !dirichlet || !out_logits)
    return az_fail(AZ_ERR_INVALID, "bad root_muzero arguments");
  k_root_muzero<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      B, A, prior_logits, dirichlet, dirichlet_fraction, invalid, out_logits);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// AZ_SAMPLE_T threads per lane share the lane's draws (sample smp belongs to thread smp %
// AZ_SAMPLE_T; every draw has its own counter, and the counts are small integers, so the order
// in which they are added up changes nothing).
#define AZ_SAMPLE_T 16
__global__ void k_root_sampled(int B, int A, const float* __restrict__ prior_logits,
                               const float* __restrict__ dirichlet, float fraction,
                               const float* __restrict__ invalid, int num_samples, float temperature,
                               float exponent, const float* __restrict__ counts_in, uint32_t key0,
                               uint32_t key1, float* __restrict__ out) {
  const int g = blockIdx.x * (blockDim.x / AZ_SAMPLE_T) + threadIdx.x / AZ_SAMPLE_T;
  const int t = threadIdx.x % AZ_SAMPLE_T;
  const int b = g < B ? g : B - 1;  // (the threads past the last lane repeat it and write nothing)
  float L[AZ_MAX_ACTIONS], sd[AZ_MAX_ACTIONS], counts[AZ_MAX_ACTIONS], prior[AZ_MAX_ACTIONS];
  az_noisy_root_logits(prior_logits + (size_t)b * A, dirichlet + (size_t)b * A, fraction,
                       invalid ? invalid + (size_t)b * A : nullptr, A, L);
  for (int a = 0; a < A; ++a) sd[a] = L[a];
  az_apply_temperature_rt(sd, temperature, A);
  if (counts_in) {
    for (int a = 0; a < A; ++a) counts[a] = counts_in[(size_t)b * A + a];
  } else {
    for (int a = 0; a < A; ++a) counts[a] = 0.0f;
    for (int smp = t; smp < num_samples; smp += AZ_SAMPLE_T) {
      int best = 0;
      float bv = 0.0f;
      for (int a = 0; a < A; ++a) {
        uint32_t idx = (uint32_t)(((size_t)b * num_samples + smp) * A + a);
        float z = az_gumbel_from_bits(az_random_bits32(key0, key1, idx)) + sd[a];
        if (a == 0 || (z > bv) || (z != z && bv == bv)) {
          best = a;
          bv = z;
        }
      }
      for (int a = 0; a < A; ++a)
        if (a == best) counts[a] += 1.0f;
    }
    for (int a = 0; a < A; ++a)
      for (int o = AZ_SAMPLE_T / 2; o > 0; o >>= 1) counts[a] += __shfl_xor(counts[a], o, AZ_SAMPLE_T);
  }
  if (g >= B || t != 0) return;
  az_softmax_rt(L, A, prior);
  for (int a = 0; a < A; ++a)
    out[(size_t)b * A + a] = (counts[a] / (float)num_samples) * az_powf(prior[a], exponent);
}

extern "C" int az_root_sampled(int32_t B, int32_t A, const float* prior_logits,
                               const float* dirichlet, float dirichlet_fraction,
                               const float* invalid, int32_t num_samples, float temperature,
                               float exponent, const float* counts_in, const uint32_t* key_host,
                               float* out_probs, void* stream) {
  if (B < 1 || A < 1 || A > AZ_MAX_ACTIONS || !prior_logits || !dirichlet || !out_probs ||
      num_samples < 1 || (!counts_in && !key_host))
    return az_fail(AZ_ERR_INVALID, "bad root_sampled arguments");
  constexpr int kLanesPerBlock = 256 / AZ_SAMPLE_T;
  k_root_sampled<<<dim3((B + kLanesPerBlock - 1) / kLanesPerBlock), dim3(256), 0, (hipStream_t)stream>>>(
      B, A, prior_logits, dirichlet, dirichlet_fraction, invalid, num_samples, temperature,
      exponent, counts_in, key_host ? key_host[0] : 0u, key_host ? key_host[1] : 0u, out_probs);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

__global__ void k_root_gumbel(int B, int A, const float* __restrict__ prior_logits,
                              const float* __restrict__ invalid,
                              const float* __restrict__ gumbel_in, uint32_t key0, uint32_t key1,
                              float scale, float* __restrict__ out_logits,
                              float* __restrict__ out_gumbel) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float o[AZ_MAX_ACTIONS];
  for (int a = 0; a < A; ++a) o[a] = prior_logits[(size_t)b * A + a];
  az_mask_invalid_rt(o, invalid ? invalid + (size_t)b * A : nullptr, A);
  for (int a = 0; a < A; ++a) {
    const size_t i = (size_t)b * A + a;
    out_logits[i] = o[a];
    float g = gumbel_in ? gumbel_in[i] : az_gumbel_from_bits(az_random_bits32(key0, key1, (uint32_t)i));
    out_gumbel[i] = scale * g;
  }
}

extern "C" int az_root_gumbel(int32_t B, int32_t A, const float* prior_logits, const float* invalid,
                              const float* gumbel_in, const uint32_t* key_host, float gumbel_scale,
                              float* out_logits, float* out_gumbel, void* stream) {
  if (B < 1 || A < 1 || A > AZ_MAX_ACTIONS || !prior_logits || !out_logits || !out_gumbel ||
      (!gumbel_in && !key_host))
    return az_fail(AZ_ERR_INVALID, "bad root_gumbel arguments");
  k_root_gumbel<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      B, A, prior_logits, invalid, gumbel_in, key_host ? key_host[0] : 0u,
      key_host ? key_host[1] : 0u, gumbel_scale, out_logits, out_gumbel);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// Dirichlet(alpha) rows: Marsaglia-Tsang gamma draws (alpha < 1 boosted by
// U^(1/alpha)) on a per-element threefry stream, normalised per row.
__global__ void k_dirichlet(int B, int A, float alpha, uint32_t key0, uint32_t key1,
                            float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float g[AZ_MAX_ACTIONS];
  float total = 0.0f;
  const float a1 = alpha < 1.0f ? alpha + 1.0f : alpha;
  const float dd = a1 - 0.3333333432674408f;
  const float cc = 1.0f / az_sqrtf(9.0f * dd);
  for (int a = 0; a < A; ++a) {
    uint32_t e0, e1;
    az_key_split(key0, key1, (uint32_t)(b * A + a), &e0, &e1);
    uint32_t ctr = 0;
    float v = 1.0f;
    for (int it = 0; it < 64; ++it) {
      float u1 = az_uniform_from_bits(az_random_bits32(e0, e1, ctr++), AZ_FLT_TINY, 1.0f);
      float u2 = az_bits_to_unit(az_random_bits32(e0, e1, ctr++));
      float u3 = az_uniform_from_bits(az_random_bits32(e0, e1, ctr++), AZ_FLT_TINY, 1.0f);
      float x = az_sqrtf(-2.0f * az_logf(u1)) * az_cosf(6.2831854820251465f * u2);
      float w = 1.0f + cc * x;
      if (w <= 0.0f) continue;
      v = (w * w) * w;
      if (az_logf(u3) < (0.5f * x) * x + dd - dd * v + dd * az_logf(v)) break;
    }
    float gam = dd * v;
    if (alpha < 1.0f) {
      float u4 = az_uniform_from_bits(az_random_bits32(e0, e1, ctr++), AZ_FLT_TINY, 1.0f);
      gam = gam * az_expf(az_logf(u4) / alpha);
    }
    g[a] = gam;
    total = total + gam;
  }
  for (int a = 0; a < A; ++a) out[(size_t)b * A + a] = total > 0.0f ? g[a] / total : 1.0f / (float)A;
}

extern "C" int az_dirichlet(int32_t B, int32_t A, float alpha, const uint32_t* key_host, float* out,
                            void* stream) {
  if (B < 1 || A < 1 || A > AZ_MAX_ACTIONS || !(alpha > 0.0f) || !key_host || !out)
    return az_fail(AZ_ERR_INVALID, "bad dirichlet arguments");
  k_dirichlet<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(B, A, alpha, key_host[0],
                                                                         key_host[1], out);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// mctx_dist/_src/policies.py:113-122 / 225-234 with tree.py:102-120
__global__ void k_epilogue_visits(az_search_desc d, az_tree t, float temperature,
                                  const float* __restrict__ gumbel_in, uint32_t key0, uint32_t key1,
                                  int32_t* __restrict__ action, float* __restrict__ weights) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int A = d.A;
  const int32_t* vis = t.children_visits + (size_t)b * (d.S + 1) * A;
  float counts[AZ_MAX_ACTIONS], lg[AZ_MAX_ACTIONS], z[AZ_MAX_ACTIONS];
  float total = 0.0f;
  for (int a = 0; a < A; ++a) {
    counts[a] = (float)vis[a];
    total = total + counts[a];
  }
  for (int a = 0; a < A; ++a) {
    float p = counts[a] / az_maxf(total, 1.0f);
    if (!(total > 0.0f)) p = (float)(1.0 / (double)A);
    weights[(size_t)b * A + a] = p;
    lg[a] = az_logf(az_maxf(p, AZ_FLT_TINY));
  }
  az_apply_temperature_rt(lg, temperature, A);
  for (int a = 0; a < A; ++a) {
    const size_t i = (size_t)b * A + a;
    float g = gumbel_in ? gumbel_in[i] : az_gumbel_from_bits(az_random_bits32(key0, key1, (uint32_t)i));
    z[a] = g + lg[a];
  }
  action[b] = az_argmax_rt(z, A);
}

extern "C" int az_epilogue_visits(const az_search_desc* d, const az_tree* t, float temperature,
                                  const float* gumbel_in, const uint32_t* key_host, int32_t* action,
                                  float* action_weights, void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (!action || !action_weights || (!gumbel_in && !key_host))
    return az_fail(AZ_ERR_INVALID, "bad epilogue arguments");
  k_epilogue_visits<<<dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      *d, *t, temperature, gumbel_in, key_host ? key_host[0] : 0u, key_host ? key_host[1] : 0u,
      action, action_weights);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// mctx_dist/_src/tree.py:102-120
__global__ void k_summary(az_search_desc d, az_tree t, float* __restrict__ visit_counts,
                          float* __restrict__ visit_probs, float* __restrict__ value,
                          float* __restrict__ qvalues) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int A = d.A, K = d.K;
  const size_t base = (size_t)b * (d.S + 1);
  const int32_t* vis = t.children_visits + base * A;
  float total = 0.0f;
  for (int a = 0; a < A; ++a) total = total + (float)vis[a];
  for (int a = 0; a < A; ++a) {
    float cnt = (float)vis[a];
    float p = cnt / az_maxf(total, 1.0f);
    if (!(total > 0.0f)) p = (float)(1.0 / (double)A);
    visit_counts[(size_t)b * A + a] = cnt;
    visit_probs[(size_t)b * A + a] = p;
    float rw = t.children_rewards[base * A + a], dc = t.children_discounts[base * A + a];
    for (int k = 0; k < K; ++k)
      qvalues[((size_t)b * A + a) * K + k] = rw + dc * t.children_values[(base * A + a) * K + k];
  }
  for (int k = 0; k < K; ++k) value[(size_t)b * K + k] = t.node_values[base * K + k];
}

extern "C" int az_summary(const az_search_desc* d, const az_tree* t, float* visit_counts,
                          float* visit_probs, float* value, float* qvalues, void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (!visit_counts || !visit_probs || !value || !qvalues)
    return az_fail(AZ_ERR_INVALID, "null summary outputs");
  k_summary<<<dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      *d, *t, visit_counts, visit_probs, value, qvalues);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// alphazero.py:367-379
__global__ void k_actor_post(az_search_desc d, az_tree t, const float* __restrict__ action_weights,
                             const int32_t* __restrict__ sampled_action, int deterministic,
                             const float* __restrict__ prior_logits, int32_t* __restrict__ action,
                             float* __restrict__ search_value, float* __restrict__ prior_log_prob) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int A = d.A, K = d.K;
  const size_t base = (size_t)b * (d.S + 1);
  float masked[AZ_MAX_ACTIONS];
  for (int a = 0; a < A; ++a) {
    float rw = t.children_rewards[base * A + a], dc = t.children_discounts[base * A + a];
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s = s + (rw + dc * t.children_values[(base * A + a) * K + k]);
    float q = s / (float)K;
    masked[a] = action_weights[(size_t)b * A + a] != 0.0f ? q : -AZ_INF;
  }
  const int best = az_argmax_rt(masked, A);
  const int act = deterministic ? best : sampled_action[b];
  action[b] = act;
  for (int k = 0; k < K; ++k) search_value[(size_t)b * K + k] = t.node_values[base * K + k];
  const float* pl = prior_logits + (size_t)b * A;
  float mx = pl[0];
  for (int a = 1; a < A; ++a) mx = az_maxf(mx, pl[a]);
  float se = 0.0f;
  for (int a = 0; a < A; ++a) se = se + az_expf(pl[a] - mx);
  prior_log_prob[b] = (pl[act] - mx) - az_logf(se);
}

extern "C" int az_actor_post(const az_search_desc* d, const az_tree* t, const float* action_weights,
                             const int32_t* sampled_action, int32_t deterministic,
                             const float* prior_logits, int32_t* action, float* search_value,
                             float* prior_log_prob, void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (!action_weights || !sampled_action || !prior_logits || !action || !search_value ||
      !prior_log_prob)
    return az_fail(AZ_ERR_INVALID, "null actor_post arrays");
  k_actor_post<<<dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      *d, *t, action_weights, sampled_action, deterministic, prior_logits, action, search_value,
      prior_log_prob);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// ---------------------------------------------------------------------------
// environment reset and the real (auto-resetting) environment step
__global__ void k_split_keys(uint32_t k0, uint32_t k1, int n, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t a, b;
  az_key_split(k0, k1, (uint32_t)i, &a, &b);
  out[2 * i] = a;
  out[2 * i + 1] = b;
}

extern "C" int az_split_keys(const uint32_t* key_host, int32_t n, uint32_t* out, void* stream) {
  if (!key_host || n < 1 || !out) return az_fail(AZ_ERR_INVALID, "bad split_keys arguments");
  k_split_keys<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(key_host[0],
                                                                             key_host[1], n, out);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

__global__ void k_random_sort_keys(uint32_t k0, uint32_t k1, int n, int64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int64_t)az_random_bits32(k0, k1, (uint32_t)i);
}

extern "C" int az_random_sort_keys(const uint32_t* key_host, int32_t n, int64_t* out, void* stream) {
  if (!key_host || n < 1 || !out) return az_fail(AZ_ERR_INVALID, "bad random_sort_keys arguments");
  k_random_sort_keys<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(key_host[0], key_host[1], n, out);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

__global__ void k_env_reset(int env_id, int B, const uint32_t* __restrict__ keys,
                            float* __restrict__ state, int32_t* __restrict__ time,
                            float* __restrict__ obs, uint32_t* __restrict__ rng_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  EnvState st;
  az_env_reset_one(env_id, keys[2 * b], keys[2 * b + 1], st);
  if (rng_out) {
    uint32_t r0, r1;  // gymnax/wrappers/brax.py:53: info['_rng'] = split(rng)[0]
    az_key_split(keys[2 * b], keys[2 * b + 1], 0u, &r0, &r1);
    rng_out[2 * b] = r0;
    rng_out[2 * b + 1] = r1;
  }
  for (int i = 0; i < 4; ++i) state[(size_t)b * 4 + i] = st.s[i];
  time[b] = st.time;
  float o[6];
  az_env_obs(env_id, st, o);
  const int od = az_env_obs_dim(env_id);
  for (int i = 0; i < od; ++i) obs[(size_t)b * od + i] = o[i];
}

extern "C" int az_env_reset(int32_t env_id, int32_t B, const uint32_t* keys, float* state,
                            int32_t* time, float* obs, uint32_t* rng_out, void* stream) {
  if (env_id < AZ_ENV_CARTPOLE || env_id > AZ_ENV_ACROBOT_NOTERM)
    return az_fail(AZ_ERR_UNSUPPORTED, "unknown env_id %d", env_id);
  if (B < 1 || !keys || !state || !time || !obs) return az_fail(AZ_ERR_INVALID, "bad reset arguments");
  k_env_reset<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(env_id, B, keys, state,
                                                                         time, obs, rng_out);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// envs/brax_wrappers.py:81-97, 109-134; gymnax/gymnax/wrappers/brax.py:57-70
__global__ void k_real_env_step(int env_id, int B, int episode_length, float* __restrict__ state,
                                int32_t* __restrict__ time, float* __restrict__ obs,
                                float* __restrict__ real_obs, float* __restrict__ reward,
                                float* __restrict__ done, float* __restrict__ truncation,
                                float* __restrict__ steps, uint32_t* __restrict__ rng,
                                const int32_t* __restrict__ action) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int od = az_env_obs_dim(env_id);
  float st_steps = done[b] != 0.0f ? 0.0f : steps[b];
  uint32_t r0, r1;
  az_key_split(rng[2 * b], rng[2 * b + 1], 0u, &r0, &r1);
  rng[2 * b] = r0;
  rng[2 * b + 1] = r1;
  EnvState st;
  EnvOut o;
  for (int i = 0; i < 4; ++i) st.s[i] = state[(size_t)b * 4 + i];
  st.time = time[b];
  az_env_step_rt(env_id, st, action[b], o);
  st_steps = st_steps + 1.0f;
  const float env_done = o.terminated;
  const float dn = st_steps >= (float)episode_length ? 1.0f : env_done;
  truncation[b] = st_steps >= (float)episode_length ? 1.0f - env_done : 0.0f;
  steps[b] = st_steps;
  reward[b] = o.reward;
  done[b] = dn;
  for (int i = 0; i < od; ++i) real_obs[(size_t)b * od + i] = o.obs[i];
  if (dn != 0.0f) {
    az_env_reset_one(env_id, r0, r1, st);
    az_env_obs(env_id, st, o.obs);
  }
  for (int i = 0; i < 4; ++i) state[(size_t)b * 4 + i] = st.s[i];
  time[b] = st.time;
  for (int i = 0; i < od; ++i) obs[(size_t)b * od + i] = o.obs[i];
}

extern "C" int az_real_env_step(int32_t env_id, int32_t B, int32_t episode_length, float* state,
                                int32_t* time, float* obs, float* real_obs, float* reward,
                                float* done, float* truncation, float* steps, uint32_t* rng,
                                const int32_t* action, void* stream) {
  if (env_id < AZ_ENV_CARTPOLE || env_id > AZ_ENV_ACROBOT_NOTERM)
    return az_fail(AZ_ERR_UNSUPPORTED, "unknown env_id %d", env_id);
  if (B < 1 || !state || !time || !obs || !real_obs || !reward || !done || !truncation || !steps ||
      !rng || !action)
    return az_fail(AZ_ERR_INVALID, "bad env step arguments");
  k_real_env_step<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      env_id, B, episode_length, state, time, obs, real_obs, reward, done, truncation, steps, rng,
      action);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// ---------------------------------------------------------------------------
// numerics specification on the device, for host/device bit comparison
__global__ void k_math_apply(int fn, int n, const float* __restrict__ x, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  float r;
  switch (fn) {
    case 0: r = az_expf(v); break;
    case 1: r = az_logf(v); break;
    case 2: r = az_sinf(v); break;
    case 3: r = az_cosf(v); break;
    case 4: r = az_swish(v); break;
    case 5: r = az_sqrtf(v); break;
    case 6: r = 1.0f / v; break;
    case 7: r = az_gumbel_from_bits(az_f2u(v)); break;
    case 9: r = az_trig_(v, 0); break;   // the select-only sine ...
    case 10: r = az_trig_(v, 1); break;  // ... and cosine (the Acrobot step's form)
    default: r = v;
  }
  y[i] = r;
}

// fn = 8: swish as the dense layers' epilogue evaluates it (four values per thread)
__global__ void k_swish_packed(int n, const float* __restrict__ x, float* __restrict__ y) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float v[4];
  for (int j = 0; j < 4; ++j) v[j] = i + j < n ? x[i + j] : 0.0f;
  const az_f32x2 a = az_swish2(az_f32x2{v[0], v[1]}), b = az_swish2(az_f32x2{v[2], v[3]});
  const float r[4] = {a.x, a.y, b.x, b.y};
  for (int j = 0; j < 4; ++j)
    if (i + j < n) y[i + j] = r[j];
}

extern "C" int az_math_apply(int32_t fn, int32_t n, const float* x, float* y, void* stream) {
  if (n < 1 || !x || !y) return az_fail(AZ_ERR_INVALID, "bad math_apply arguments");
  if (fn == 8) {
    k_swish_packed<<<dim3((n + 1023) / 1024), dim3(256), 0, (hipStream_t)stream>>>(n, x, y);
    AZ_LAUNCH_CHECK();
    return AZ_OK;
  }
  k_math_apply<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(fn, n, x, y);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

__global__ void k_threefry(int n, const uint32_t* __restrict__ key, const uint32_t* __restrict__ ctr,
                           uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t a, b;
  az_threefry2x32(key[2 * i], key[2 * i + 1], ctr[2 * i], ctr[2 * i + 1], &a, &b);
  out[2 * i] = a;
  out[2 * i + 1] = b;
}

extern "C" int az_threefry(int32_t n, const uint32_t* key, const uint32_t* ctr, uint32_t* out,
                           void* stream) {
  if (n < 1 || !key || !ctr || !out) return az_fail(AZ_ERR_INVALID, "bad threefry arguments");
  k_threefry<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(n, key, ctr, out);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

