// az_search.hip -- tree initialisation, the persistent tree search with injected evaluations, the per-simulation kernels and the Gumbel epilogue.
// Part of libaz_mcts.so (gfx950 only); built with -ffp-contract=off, which is part of
// the numerics contract (every fused multiply-add is an explicit az_fmaf, az_math.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#ifdef AZ_STAMP
// Diagnostic build only: cycle stamps of thread 0 of workgroup 0, one row per simulation.
__device__ long long az_tree_stamps[64][8];
#define AZ_WAVE_STAMP(k)                                                        \
  do {                                                                          \
    if (blockIdx.x == 0 && threadIdx.x == 0 && sim < 64)                        \
      az_tree_stamps[sim][k] = __builtin_amdgcn_s_memtime();                    \
  } while (0)
#endif

#include "az_host.h"
#include "az_wave.h"

// ---------------------------------------------------------------------------
// tree init: one thread per (lane, node)
__global__ void k_tree_init(az_search_desc d, az_tree t, const float* __restrict__ root_logits,
                            const float* __restrict__ root_value,
                            const float* __restrict__ root_state,
                            const int32_t* __restrict__ root_time) {
  const int N = d.S + 1, A = d.A, K = d.K;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)d.B * N) return;
  const int b = (int)(i / N), n = (int)(i % N);
  const bool root = n == 0;
  t.node_visits[i] = root ? 1 : 0;
  t.parents[i] = -1;
  t.action_from_parent[i] = -1;
  for (int k = 0; k < K; ++k) {
    float v = root ? root_value[(size_t)b * K + k] : 0.0f;
    t.raw_values[i * K + k] = v;
    t.node_values[i * K + k] = v;
  }
  for (int a = 0; a < A; ++a) {
    t.children_index[i * A + a] = AZ_UNVISITED;
    t.children_prior_logits[i * A + a] = root ? root_logits[(size_t)b * A + a] : 0.0f;
    t.children_visits[i * A + a] = 0;
    t.children_rewards[i * A + a] = 0.0f;
    t.children_discounts[i * A + a] = 0.0f;
    for (int k = 0; k < K; ++k) t.children_values[(i * A + a) * K + k] = 0.0f;
  }
  if (t.emb_state)
    for (int j = 0; j < 4; ++j)
      t.emb_state[i * 4 + j] = (root && root_state) ? root_state[(size_t)b * 4 + j] : 0.0f;
  if (t.emb_time) t.emb_time[i] = (root && root_time) ? root_time[b] : 0;
}

extern "C" int az_tree_init(const az_search_desc* d, const az_tree* t, const float* root_logits,
                            const float* root_value, const float* root_state,
                            const int32_t* root_time, void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (!root_logits || !root_value) return az_fail(AZ_ERR_INVALID, "null root arrays");
  const size_t total = (size_t)d->B * (d->S + 1);
  k_tree_init<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      *d, *t, root_logits, root_value, root_state, root_time);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// persistent tree search with injected evaluations: one thread per lane
template <int AMAX, int KMAX>
__global__ __launch_bounds__(64) void k_search_injected(SearchCtx c, az_tree t, az_injected inj,
                                                        unsigned long long* sum_depth) {
  const az_search_desc& d = c.d;
  const int A = d.A, K = d.K;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  long long depth_sum = 0;
  if (b < d.B) {
    Lane<AMAX, KMAX> l(t, b, d.S + 1, d.A, d.K);
    const bool need_keys = d.variant != AZ_VARIANT_GUMBEL && d.noise_mode == AZ_NOISE_THREEFRY;
    uint32_t r0 = d.key[0], r1 = d.key[1];
    for (int sim = 0; sim < d.S; ++sim) {
      uint32_t k0 = 0, k1 = 0;
      if (need_keys) {
        uint32_t s0, s1;
        az_next_sim_key(r0, r1, s0, s1);
        az_key_split(s0, s1, (uint32_t)b, &k0, &k1);
      }
      int parent, action, nxt, depth;
      az_simulate_lane<AMAX, KMAX>(c, l, b, sim, k0, k1, parent, action, nxt, depth);
      const int node = nxt == AZ_UNVISITED ? sim + 1 : nxt;
      depth_sum += depth;
      const size_t e = (size_t)b * d.S + sim;
      float logits[AMAX], value[KMAX];
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a < d.A) logits[a] = inj.prior_logits[e * d.A + a];
AZ_UNROLL_K
      for (int k = 0; k < AZ_TRIP_K; ++k)
        if (k < d.K) value[k] = inj.value[e * d.K + k];
      az_expand_write<AMAX, KMAX>(l, parent, action, node, logits, value, inj.reward[e],
                                  inj.discount[e]);
      az_backward_lane<AMAX, KMAX>(l, node);
    }
  }
  az_add_depth(sum_depth, depth_sum);
}

#ifdef AZ_STAMP
#define AZ_TREE_STAMP(k) AZ_WAVE_STAMP(k)
// kernel entry, simulation loop entered, loop left, kernel end (thread 0 of workgroup 0)
__device__ long long az_tree_edges[4];
#define AZ_TREE_EDGE(k)                                                                     \
  do {                                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0) az_tree_edges[k] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int az_debug_read_tree_stamps(long long* host_out) {
  AZ_HIP(hipDeviceSynchronize());
  AZ_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(az_tree_stamps), sizeof(long long) * 64 * 8));
  return AZ_OK;
}
extern "C" int az_debug_read_tree_edges(long long* host_out) {
  AZ_HIP(hipDeviceSynchronize());
  AZ_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(az_tree_edges), sizeof(long long) * 4));
  return AZ_OK;
}
#else
#define AZ_TREE_EDGE(k) \
  do {                  \
  } while (0)
#define AZ_TREE_STAMP(k) \
  do {                   \
  } while (0)
#endif

// The same search in the cooperative form (az_wave.h): 16 lanes per workgroup, two lanes per
// wave, 32 threads per lane; cached selections and the path in LDS; no workgroup barrier inside
// the simulation loop.
// TREE_LDS: the statistics arrays are staged in LDS for the whole search (a template parameter
// so that the compiler sees LDS addresses and emits LDS instructions, not generic ones).
// EXACT: A == AMAX and K == KMAX (compile-time shapes); RULE: see az_rule_ctx.
template <int AMAX, int KMAX, bool EXACT, int RULE, bool NOISY, bool TREE_LDS>
__global__ __launch_bounds__(AZ_TREE_THREADS) void k_search_wave_injected(
    SearchCtx c_in, az_tree t, az_injected inj, unsigned long long* sum_depth) {
  constexpr int tree_in_lds = TREE_LDS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  AZ_TREE_EDGE(0);
  const SearchCtx c = az_rule_ctx<RULE>(c_in);
  const az_search_desc& d = c.d;
  const int A = EXACT ? AMAX : d.A, K = EXACT ? KMAX : d.K, S = d.S;
  const int tid = threadIdx.x, ln = tid / AZ_POS, h = ln & 1, j = tid & (AZ_POS - 1);
  const int b0 = blockIdx.x * AZ_WG_LANES, b = b0 + ln;
  const bool active = b < d.B;
  WaveLds s(lds, ln, S);
  uint32_t* sim_keys = reinterpret_cast<uint32_t*>(lds + AZ_WG_LANES * WaveLds::lane_words(S));
  const int lanes_here = d.B - b0 < AZ_WG_LANES ? d.B - b0 : AZ_WG_LANES;
  TreeLds tl(reinterpret_cast<float*>(sim_keys + az_key_words(S)), S + 1, A, K);
  // the first simulation's recurrent_fn outputs are requested before anything waits for memory
  float ev_logits[AMAX], ev_value[KMAX], reward, discount;
  {
    const size_t e = (size_t)(active ? b : 0) * S;
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) ev_logits[a] = inj.prior_logits[e * A + a];
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k)
      if (k < K) ev_value[k] = inj.value[e * K + k];
    reward = inj.reward[e];
    discount = inj.discount[e];
  }
  // Gumbel rules with the tree in LDS: max(tiny, softmax(prior logits)) of every node the search
  // can create is worked out here, one node per thread, instead of by all the path's threads in
  // every simulation -- node n > 0 takes the prior logits of evaluation n - 1.
  constexpr bool cache_probs = TREE_LDS && RULE == 1;
  float* const node_probs =
      cache_probs ? reinterpret_cast<float*>(sim_keys + az_key_words(S)) + TreeLds::words(S + 1, A, K) +
                        (size_t)ln * (S + 1) * A
                  : nullptr;
  if (cache_probs) {
    const size_t bb = active ? b : 0;
    for (int n = j; n <= S; n += AZ_POS) {
      const float* src = n == 0 ? t.children_prior_logits + bb * (S + 1) * A
                                : inj.prior_logits + (bb * S + (n - 1)) * A;
      float lg[AMAX], pr[AMAX];
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a < A) lg[a] = src[a];
      az_prior_probs<AMAX>(lg, A, pr);
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a < A) node_probs[n * A + a] = pr[a];
    }
  }
  if (tree_in_lds) tl.template copy<true>(t, b0, lanes_here, tid, blockDim.x);
  const az_tree tv = tree_in_lds ? tl.view() : t;
  Lane<AMAX, KMAX> l(tv, tree_in_lds ? ln : (active ? b : 0), S + 1, A, K);
  const bool need_keys = NOISY && d.variant != AZ_VARIANT_GUMBEL && d.noise_mode == AZ_NOISE_THREEFRY;
  if (need_keys) az_wave_sim_keys(d, sim_keys, tid, blockDim.x, c.in.sim_keys);
  if (j == 0) s.selx[0] = az_i32x4{0, 0, 0, 0};
  __syncthreads();
  az_wave_init<AMAX, KMAX, NOISY>(c, l, s, b, j, active);
  __syncthreads();
  long long depth_sum = 0;
  float nx_logits[AMAX], nx_value[KMAX], nx_reward, nx_discount;
  AZ_TREE_EDGE(1);
  for (int sim = 0; sim < S; ++sim) {
    AZ_TREE_STAMP(0);
    // the recurrent_fn outputs of the NEXT simulation of this lane are requested a whole
    // simulation before the update that consumes them, so that their latency is never waited for
    auto prefetch = [&] {
      const int sn = sim + 1 < S ? sim + 1 : sim;
      const size_t e = (size_t)(active ? b : 0) * S + sn;
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a < A) nx_logits[a] = inj.prior_logits[e * A + a];
AZ_UNROLL_K
      for (int k = 0; k < AZ_TRIP_K; ++k)
        if (k < K) nx_value[k] = inj.value[e * K + k];
      nx_reward = inj.reward[e];
      nx_discount = inj.discount[e];
    };
    WavePath wp;
    if (NOISY) {
      wp = az_wave_chase_flagged<AMAX, KMAX>(c, l, s, b, h, j, sim, d.max_depth, active,
                                             need_keys ? sim_keys : nullptr, prefetch);
    } else {
      wp = az_wave_chase_lean(s, h, j, sim, d.max_depth, active, prefetch);
    }
    if (j == 0) depth_sum += wp.dep;
    AZ_TREE_STAMP(2);
    az_wave_update<AMAX, KMAX, NOISY, RULE == 1>(c, l, s, active, h, j, sim, wp, ev_logits, ev_value, reward,
                                      discount, nullptr, node_probs);
    AZ_TREE_STAMP(7);
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a) ev_logits[a] = nx_logits[a];
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k) ev_value[k] = nx_value[k];
    reward = nx_reward;
    discount = nx_discount;
  }
  AZ_TREE_EDGE(2);
  if (tree_in_lds) {
    __syncthreads();
    tl.template copy<false>(t, b0, lanes_here, tid, blockDim.x);
  }
  az_add_depth(sum_depth, depth_sum);
  AZ_TREE_EDGE(3);
}

// The cooperative tree search for two actions and one value atom under the Gumbel rules with the
// whole tree of a workgroup's lanes in LDS as array-of-structures rows (az_aos21_load): the
// simulation loop touches no global memory at all.  The tree handed in must be FRESH (as
// az_tree_init leaves it): only its root is read; every other row starts as the fresh pattern with
// the fields an expansion would write -- prior logits, their prior probabilities, raw = node
// value -- already filled in from the evaluation that creates it (node n > 0 is created by
// simulation n - 1, search.py:97-99; a simulation that ends on the depth limit re-expands an
// existing node from row sim + 1, which then stays unused).  The evaluation's reward and
// discount wait in LDS next to the rows.  After the last simulation every thread unpacks its
// rows into the reference's arrays; rows that were never expanded go back as the fresh pattern.
static size_t az_aos_lds_bytes(int S) {
  const int N = S + 1;
  return ((size_t)AZ_WG_LANES * WaveLds::lane_words(S) + (size_t)AZ_WG_LANES * N * AZ_AOS_WORDS +
          (size_t)2 * AZ_WG_LANES * N + (size_t)2 * AZ_WG_LANES * S) * 4;
}

__global__ __launch_bounds__(AZ_TREE_THREADS) void k_search_wave_aos21(SearchCtx c_in, az_tree t, az_injected inj,
                                                                      unsigned long long* sum_depth) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  AZ_TREE_EDGE(0);
  const SearchCtx c = az_rule_ctx<1>(c_in);
  const az_search_desc& d = c.d;
  constexpr int A = 2;
  const int S = d.S, N = S + 1;
  const int tid = threadIdx.x, ln = tid / AZ_POS, h = ln & 1, j = tid & (AZ_POS - 1);
  const int b = blockIdx.x * AZ_WG_LANES + ln;
  const bool active = b < d.B;
  const size_t bb = active ? b : 0;
  WaveLds s(lds, ln, S);
  float* const rows = lds + AZ_WG_LANES * WaveLds::lane_words(S) + (size_t)ln * N * AZ_AOS_WORDS;
  int32_t* const par_base =
      reinterpret_cast<int32_t*>(lds + AZ_WG_LANES * WaveLds::lane_words(S) + (size_t)AZ_WG_LANES * N * AZ_AOS_WORDS);
  int32_t* const par = par_base + ln * N;
  int32_t* const afp = par_base + AZ_WG_LANES * N + ln * N;
  float* const evrd = reinterpret_cast<float*>(par_base + 2 * AZ_WG_LANES * N) + ln * S * 2;
  // the rows, one node per thread and turn
  for (int n = j; n <= S; n += AZ_POS) {
    FullRow<2, 1> f{};
    if (n == 0) {
      const size_t i = bb * N;
      f.nv = t.node_visits[i];
      f.raw[0] = t.raw_values[i];
      f.nval[0] = t.node_values[i];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        f.cidx[a] = t.children_index[i * A + a];
        f.vis[a] = t.children_visits[i * A + a];
        f.logits[a] = t.children_prior_logits[i * A + a];
        f.rew[a] = t.children_rewards[i * A + a];
        f.disc[a] = t.children_discounts[i * A + a];
        f.cval[a][0] = t.children_values[i * A + a];
      }
      par[0] = t.parents[i];
      afp[0] = t.action_from_parent[i];
    } else {
      const size_t e = bb * S + (n - 1);
      f.cidx[0] = f.cidx[1] = AZ_UNVISITED;
      f.logits[0] = inj.prior_logits[e * A];
      f.logits[1] = inj.prior_logits[e * A + 1];
      f.raw[0] = f.nval[0] = inj.value[e];
      evrd[(n - 1) * 2] = inj.reward[e];
      evrd[(n - 1) * 2 + 1] = inj.discount[e];
      par[n] = -1;
      afp[n] = -1;
    }
    float pr[2];
    az_prior_probs<2>(f.logits, A, pr);
    az_aos21_store(rows + n * AZ_AOS_WORDS, f, pr[0], pr[1]);
  }
  az_tree tv{};
  tv.parents = par_base;
  tv.action_from_parent = par_base + AZ_WG_LANES * N;
  Lane<2, 1> l(tv, ln, N, A, 1);
  if (j == 0) s.selx[0] = az_i32x4{0, 0, 0, 0};
  __syncthreads();
  az_wave_init<2, 1, false>(c, l, s, b, j, active, rows);
  __syncthreads();
  long long depth_sum = 0;
  AZ_TREE_EDGE(1);
  for (int sim = 0; sim < S; ++sim) {
    AZ_TREE_STAMP(0);
    const WavePath wp = az_wave_chase_lean(s, h, j, sim, d.max_depth, active);
    if (j == 0) depth_sum += wp.dep;
    AZ_TREE_STAMP(2);
    const float* ev = rows + (sim + 1) * AZ_AOS_WORDS;  // (the row this simulation's evaluation was filled into)
    az_wave_update<2, 1, false, true, true>(c, l, s, active, h, j, sim, wp, ev + AZ_AOS_LOGITS, ev + AZ_AOS_RAW,
                                            evrd[sim * 2], evrd[sim * 2 + 1], nullptr, nullptr, rows);
    AZ_TREE_STAMP(7);
  }
  AZ_TREE_EDGE(2);
  // (a wave writes its own two lanes back: no workgroup barrier, so that the waves that finish
  // early are done while the slowest one still searches)
  az_wave_sync();
  if (active) {
    for (int n = j; n <= S; n += AZ_POS) {
      FullRow<2, 1> f;
      az_aos21_load(rows + n * AZ_AOS_WORDS, f);
      if (n > 0 && f.nv == 0) {  // never expanded: the fresh pattern (its evaluation was not used)
        f.logits[0] = f.logits[1] = 0.0f;
        f.raw[0] = f.nval[0] = 0.0f;
      }
      const size_t i = (size_t)b * N + n;
      t.node_visits[i] = f.nv;
      t.raw_values[i] = f.raw[0];
      t.node_values[i] = f.nval[0];
      t.parents[i] = par[n];
      t.action_from_parent[i] = afp[n];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        t.children_index[i * A + a] = f.cidx[a];
        t.children_visits[i * A + a] = f.vis[a];
        t.children_prior_logits[i * A + a] = f.logits[a];
        t.children_rewards[i * A + a] = f.rew[a];
        t.children_discounts[i * A + a] = f.disc[a];
        t.children_values[i * A + a] = f.cval[a][0];
      }
    }
  }
  az_add_depth(sum_depth, depth_sum);
  AZ_TREE_EDGE(3);
}

// LDS bytes of the cooperative form's own state (cached selections, path, root inputs, keys).
static size_t az_wave_lds_bytes(const az_search_desc* d) {
  return ((size_t)AZ_WG_LANES * WaveLds::lane_words(d->S) + (size_t)az_key_words(d->S)) * 4;
}

// Whether the cooperative form applies (see az_wave.h) and fits in LDS next to
// `extra_lds_bytes` of other use.
bool az_coop_applicable(const az_search_desc* d, size_t extra_lds_bytes, bool borrowed) {
  if (d->kernel_form == 1) return false;  // the caller asked for the literal kernels
  const bool k16 = d->K == 16 && (d->A == 2 || d->A == 3);  // the compile-time 16-atom shapes
  if (d->A > 8 || (d->K > 8 && !k16) || d->S + 1 > 65535) return false;
  (void)borrowed;
  return az_wave_lds_bytes(d) + extra_lds_bytes <= 160 * 1024;
}

extern "C" int az_search_injected(const az_search_desc* d, const az_tree* t,
                                  const az_search_inputs* in, const az_injected* inj,
                                  int64_t* sum_depth, void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (int rc = az_check_inputs(d, in)) return rc;
  if (!inj || !inj->prior_logits || !inj->value || !inj->reward || !inj->discount)
    return az_fail(AZ_ERR_INVALID, "null injected evaluation arrays");
  if (d->variant == AZ_VARIANT_STOCHASTIC)
    return az_fail(AZ_ERR_UNSUPPORTED, "the stochastic variant runs on az_simulate / az_expand_backward only");
  SearchCtx c{*d, *in};
  if (d->kernel_form == 0 && d->A == 2 && d->K == 1 && d->variant == AZ_VARIANT_GUMBEL && az_rule_of(d) == 1 &&
      d->S + 1 <= 65535 && az_aos_lds_bytes(d->S) <= 160 * 1024) {
    const size_t lds = az_aos_lds_bytes(d->S);
    if (lds > 48 * 1024)
      AZ_HIP(hipFuncSetAttribute((const void*)k_search_wave_aos21, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_search_wave_aos21<<<dim3((d->B + AZ_WG_LANES - 1) / AZ_WG_LANES), dim3(AZ_TREE_THREADS), lds,
                          (hipStream_t)stream>>>(c, *t, *inj, reinterpret_cast<unsigned long long*>(sum_depth));
    AZ_LAUNCH_CHECK();
    return AZ_OK;
  }
  if (az_coop_applicable(d, 0, false)) {
    size_t lds = az_wave_lds_bytes(d);
    const int rule = az_rule_of(d);
    // (the tree's copy, and behind it the Gumbel rules' per-node prior probabilities)
    const size_t tree_bytes =
        (TreeLds::words(d->S + 1, d->A, d->K) + (rule == 1 ? (size_t)AZ_WG_LANES * (d->S + 1) * d->A : 0)) * 4;
    const int tree_in_lds = lds + tree_bytes <= 160 * 1024;
    if (tree_in_lds) lds += tree_bytes;
    const bool noisy = d->variant != AZ_VARIANT_GUMBEL && d->noise_mode != AZ_NOISE_ZERO;
    return az_dispatch_coop(d->A, d->K, [&]<int AMAX, int KMAX, bool EXACT>() -> int {
      using Kern = void (*)(SearchCtx, az_tree, az_injected, unsigned long long*);
      Kern kern;
      // the specialised rules exist for the compile-time shapes only
      if (noisy) {
        if (EXACT && rule == 2)
          kern = tree_in_lds ? k_search_wave_injected<AMAX, KMAX, EXACT, EXACT ? 2 : 0, true, true>
                             : k_search_wave_injected<AMAX, KMAX, EXACT, EXACT ? 2 : 0, true, false>;
        else
          kern = tree_in_lds ? k_search_wave_injected<AMAX, KMAX, EXACT, 0, true, true>
                             : k_search_wave_injected<AMAX, KMAX, EXACT, 0, true, false>;
      } else {
        if (EXACT && rule == 1)
          kern = tree_in_lds ? k_search_wave_injected<AMAX, KMAX, EXACT, EXACT ? 1 : 0, false, true>
                             : k_search_wave_injected<AMAX, KMAX, EXACT, EXACT ? 1 : 0, false, false>;
        else
          kern = tree_in_lds ? k_search_wave_injected<AMAX, KMAX, EXACT, 0, false, true>
                             : k_search_wave_injected<AMAX, KMAX, EXACT, 0, false, false>;
      }
      if (lds > 48 * 1024)
        AZ_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
      kern<<<dim3((d->B + AZ_WG_LANES - 1) / AZ_WG_LANES), dim3(AZ_TREE_THREADS), lds,
             (hipStream_t)stream>>>(c, *t, *inj, reinterpret_cast<unsigned long long*>(sum_depth));
      AZ_LAUNCH_CHECK();
      return AZ_OK;
    });
  }
  return az_dispatch_ak(d->A, d->K, [&]<int AMAX, int KMAX>() -> int {
    k_search_injected<AMAX, KMAX><<<dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
        c, *t, *inj, reinterpret_cast<unsigned long long*>(sum_depth));
    AZ_LAUNCH_CHECK();
    return AZ_OK;
  });
}

// ---------------------------------------------------------------------------
// per-simulation kernels for an arbitrary caller-side recurrent_fn
template <int AMAX, int KMAX>
__global__ __launch_bounds__(64) void k_simulate(SearchCtx c, az_tree t, int sim, uint32_t s0,
                                                 uint32_t s1, int32_t* __restrict__ parent_out,
                                                 int32_t* __restrict__ action_out,
                                                 int32_t* __restrict__ next_out,
                                                 unsigned long long* sum_depth) {
  const az_search_desc& d = c.d;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  long long depth_sum = 0;
  if (b < d.B) {
    Lane<AMAX, KMAX> l(t, b, d.S + 1, d.A, d.K);
    uint32_t k0 = 0, k1 = 0;
    if (d.variant != AZ_VARIANT_GUMBEL && d.noise_mode == AZ_NOISE_THREEFRY)
      az_key_split(s0, s1, (uint32_t)b, &k0, &k1);
    int parent, action, nxt, depth;
    az_simulate_lane<AMAX, KMAX>(c, l, b, sim, k0, k1, parent, action, nxt, depth);
    parent_out[b] = parent;
    action_out[b] = action;
    next_out[b] = nxt == AZ_UNVISITED ? sim + 1 : nxt;
    depth_sum = depth;
  }
  az_add_depth(sum_depth, depth_sum);
}

extern "C" int az_simulate(const az_search_desc* d, const az_tree* t, const az_search_inputs* in,
                           int32_t sim, int32_t* parent, int32_t* action, int32_t* next,
                           int64_t* sum_depth, void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (int rc = az_check_inputs(d, in)) return rc;
  if (sim < 0 || sim >= d->S) return az_fail(AZ_ERR_INVALID, "sim=%d outside [0,%d)", sim, d->S);
  if (!parent || !action || !next) return az_fail(AZ_ERR_INVALID, "null output arrays");
  // simulate_key of simulation `sim`: walk the split(rng, 3) chain on the host
  uint32_t r0 = d->key[0], r1 = d->key[1], s0 = 0, s1 = 0;
  for (int i = 0; i <= sim; ++i) {
    uint32_t n0, n1;
    az_key_split(r0, r1, 1u, &s0, &s1);
    az_key_split(r0, r1, 0u, &n0, &n1);
    r0 = n0;
    r1 = n1;
  }
  SearchCtx c{*d, *in};
  return az_dispatch_ak(d->A, d->K, [&]<int AMAX, int KMAX>() -> int {
    k_simulate<AMAX, KMAX><<<dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
        c, *t, sim, s0, s1, parent, action, next, reinterpret_cast<unsigned long long*>(sum_depth));
    AZ_LAUNCH_CHECK();
    return AZ_OK;
  });
}

template <int AMAX, int KMAX>
__global__ __launch_bounds__(64) void k_expand_backward(
    az_search_desc d, az_tree t, const int32_t* __restrict__ parent,
    const int32_t* __restrict__ action, const int32_t* __restrict__ next,
    const float* __restrict__ prior_logits, const float* __restrict__ value,
    const float* __restrict__ reward, const float* __restrict__ discount) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int A = d.A, K = d.K;
  Lane<AMAX, KMAX> l(t, b, d.S + 1, d.A, d.K);
  float logits[AMAX], val[KMAX];
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < d.A) logits[a] = prior_logits[(size_t)b * d.A + a];
AZ_UNROLL_K
  for (int k = 0; k < AZ_TRIP_K; ++k)
    if (k < d.K) val[k] = value[(size_t)b * d.K + k];
  const int node = next[b];
  az_expand_write<AMAX, KMAX>(l, parent[b], action[b], node, logits, val, reward[b], discount[b]);
  az_backward_lane<AMAX, KMAX>(l, node);
}

extern "C" int az_expand_backward(const az_search_desc* d, const az_tree* t, const int32_t* parent,
                                  const int32_t* action, const int32_t* next,
                                  const float* prior_logits, const float* value,
                                  const float* reward, const float* discount, void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (!parent || !action || !next || !prior_logits || !value || !reward || !discount)
    return az_fail(AZ_ERR_INVALID, "null expand arrays");
  return az_dispatch_ak(d->A, d->K, [&]<int AMAX, int KMAX>() -> int {
    k_expand_backward<AMAX, KMAX><<<dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
        *d, *t, parent, action, next, prior_logits, value, reward, discount);
    AZ_LAUNCH_CHECK();
    return AZ_OK;
  });
}

// mctx_dist/_src/policies.py:321-343
template <int AMAX, int KMAX>
__global__ void k_epilogue_gumbel(az_search_desc d, az_tree t, const float* __restrict__ root_logits,
                                  const float* __restrict__ gumbel, const float* __restrict__ invalid,
                                  int32_t* __restrict__ action, float* __restrict__ weights) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int A = d.A;
  Lane<AMAX, KMAX> l(t, b, d.S + 1, A, d.K);
  NodeRow<AMAX> r;
  az_load_row<AMAX, KMAX>(l, AZ_ROOT, r);
  int32_t considered_visit = r.vis[0];
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a)
    if (a < A && r.vis[a] > considered_visit) considered_visit = r.vis[a];
  float cq[AMAX], score[AMAX], rl[AMAX], x[AMAX], w[AMAX];
  az_qtransform<AMAX>(d, A, r, cq);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) rl[a] = root_logits[(size_t)b * A + a];
  az_score_considered<AMAX>(considered_visit, gumbel + (size_t)b * A, rl, cq, r.vis, A, score);
  const float* inv = invalid ? invalid + (size_t)b * A : nullptr;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A && inv && inv[a] != 0.0f) score[a] = -AZ_INF;
  action[b] = az_argmax<AMAX>(score, A);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) x[a] = rl[a] + cq[a];
  if (inv) {
    float mx = x[0];
AZ_UNROLL_A
    for (int a = 1; a < AZ_TRIP_A; ++a)
      if (a < A) mx = az_maxf(mx, x[a]);
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) {
        x[a] = x[a] - mx;
        if (inv[a] != 0.0f) x[a] = -AZ_FLT_MAX;
      }
  }
  az_softmax<AMAX>(x, A, w);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) weights[(size_t)b * A + a] = w[a];
}

extern "C" int az_epilogue_gumbel(const az_search_desc* d, const az_tree* t,
                                  const float* root_logits, const float* root_gumbel,
                                  const float* invalid, int32_t* action, float* action_weights,
                                  void* stream) {
  if (int rc = az_check_desc(d, t)) return rc;
  if (!root_logits || !root_gumbel || !action || !action_weights)
    return az_fail(AZ_ERR_INVALID, "bad gumbel epilogue arguments");
  return az_dispatch_ak(d->A, d->K, [&]<int AMAX, int KMAX>() -> int {
    k_epilogue_gumbel<AMAX, KMAX><<<dim3((d->B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
        *d, *t, root_logits, root_gumbel, invalid, action, action_weights);
    AZ_LAUNCH_CHECK();
    return AZ_OK;
  });
}

