// az_device.h -- per-lane device functions of the batched tree search.
//
// One batch lane = one independent tree.  The tree arrays live in HBM in the
// reference's layout (mctx_dist/_src/tree.py:60-73); a Lane<> is the view of one
// lane's slices.  All arithmetic is binary32 in the order the reference writes it
// and uses the elementary functions of az_math.h, so the integer outputs (actions,
// visit counts, node indices) equal those of a host evaluation bit for bit.
//
// AMAX / KMAX are compile-time bounds on the number of actions / value atoms so
// that the per-action vectors stay in registers; the run-time A <= AMAX, K <= KMAX.
#ifndef AZ_DEVICE_H_
#define AZ_DEVICE_H_

#include <hip/hip_runtime.h>

#include "../../include/az_mcts.h"
#include "az_math.h"

#define AZ_ROOT 0
#define AZ_UNVISITED (-1)

#define AZ_DEV __device__ __forceinline__

// Loops over actions / atoms: fully unrolled for compile-time bounds up to 16 (the
// vectors then live in registers), plain run-time loops for the generic bound.
// The enclosing function must have `A` (resp. `K`) in scope.
constexpr int az_uf(int n) { return n <= 16 ? n : 1; }
#define AZ_UNROLL_A _Pragma("unroll az_uf(AMAX)")
#define AZ_TRIP_A (AMAX <= 16 ? AMAX : A)
#define AZ_UNROLL_K _Pragma("unroll az_uf(KMAX)")
#define AZ_TRIP_K (KMAX <= 16 ? KMAX : K)

// One lane's part of one of the tree's arrays: the array itself -- the same address in every
// thread, which stays in scalar registers -- and the lane's BYTE offset into it, 32 bits that all
// arrays of one shape share, instead of a 64-bit address per array in every thread (thirteen
// arrays: 26 vector registers).  The host refuses trees whose arrays reach 4 GiB
// (az_check_desc).
template <typename T>
struct LaneArr {
  T* base;
  uint32_t off;
  AZ_DEV T* at(uint32_t i) const {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (off + i * (uint32_t)sizeof(T)));
  }
  AZ_DEV T& operator[](size_t i) const { return *at((uint32_t)i); }
  AZ_DEV T* operator+(size_t i) const { return at((uint32_t)i); }
  AZ_DEV explicit operator bool() const { return base != nullptr; }
};

template <int AMAX, int KMAX>
struct Lane {
  int A, K;
  LaneArr<int32_t> nv, par, afp, cidx, cvis;
  LaneArr<float> raw, nval, clog, crew, cdisc, cval;
  LaneArr<float> es;
  LaneArr<int32_t> et;

  AZ_DEV Lane(const az_tree& t, int b, int N, int A_, int K_) : A(A_), K(K_) {
    const uint32_t bn = (uint32_t)b * (uint32_t)N * 4u;  // bytes of the [B, N] arrays ahead of the lane
    nv = {t.node_visits, bn};
    par = {t.parents, bn};
    afp = {t.action_from_parent, bn};
    cidx = {t.children_index, bn * A};
    cvis = {t.children_visits, bn * A};
    raw = {t.raw_values, bn * K};
    nval = {t.node_values, bn * K};
    clog = {t.children_prior_logits, bn * A};
    crew = {t.children_rewards, bn * A};
    cdisc = {t.children_discounts, bn * A};
    cval = {t.children_values, bn * A * K};
    es = {t.emb_state, bn * 4};
    et = {t.emb_time, bn};
  }
};

// jnp.argmax: first maximal element, NaN counts as maximal.
template <int AMAX>
AZ_DEV int az_argmax(const float (&x)[AMAX], int A) {
  int best = 0;
  float bv = x[0];
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a) {
    if (a < A) {
      float v = x[a];
      if ((v > bv) || (v != v && bv == bv)) {
        best = a;
        bv = v;
      }
    }
  }
  return best;
}

// jax.nn.softmax: exp(x - max) / sum, left-to-right sum.
template <int AMAX>
AZ_DEV void az_softmax(const float (&x)[AMAX], int A, float (&out)[AMAX]) {
  float m = x[0];
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a)
    if (a < A) m = az_maxf(m, x[a]);
  float s = 0.0f;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      out[a] = az_expf(x[a] - m);
      s = s + out[a];
    }
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) out[a] = out[a] / s;
}

// Row `node` of the per-action statistics, loaded once per selection step.
template <int AMAX>
struct NodeRow {
  int32_t vis[AMAX];
  float logits[AMAX];
  float q[AMAX];  // mean over atoms of reward + discount * child value
  float vmean;    // mean over atoms of node_values[node]
  float rawmean;  // mean over atoms of raw_values[node]
  int32_t node_visit;
};

template <int AMAX, int KMAX>
AZ_DEV float az_mean_atoms(const float* x, int K) {
  float s = 0.0f;
  if (KMAX == 1) {
    s = s + x[0];
  } else {
    for (int k = 0; k < K; ++k) s = s + x[k];
  }
  return s / (float)K;
}

// mctx_dist/_src/tree.py:142-147 + the mean over atoms every qtransform applies first.
template <int AMAX, int KMAX>
AZ_DEV void az_load_row(const Lane<AMAX, KMAX>& l, int node, NodeRow<AMAX>& r) {
  const int A = l.A, K = l.K;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a) {
    if (a < A) {
      r.vis[a] = l.cvis[node * A + a];
      r.logits[a] = l.clog[node * A + a];
      float rw = l.crew[node * A + a], dc = l.cdisc[node * A + a];
      const float* cv = l.cval + ((size_t)node * A + a) * K;
      float s = 0.0f;
      if (KMAX == 1) {
        s = s + (rw + dc * cv[0]);
      } else {
        for (int k = 0; k < K; ++k) s = s + (rw + dc * cv[k]);
      }
      r.q[a] = s / (float)K;
    }
  }
  r.vmean = az_mean_atoms<AMAX, KMAX>(l.nval + (size_t)node * K, K);
  r.rawmean = az_mean_atoms<AMAX, KMAX>(l.raw + (size_t)node * K, K);
  r.node_visit = l.nv[node];
}

// mctx_dist/_src/qtransforms.py:54-87
template <int AMAX>
AZ_DEV void az_qt_parent_siblings(const az_search_desc& d, int A, const NodeRow<AMAX>& r,
                                  float (&out)[AMAX]) {
  float v = r.vmean;
  float lo = r.vis[0] > 0 ? r.q[0] : v;
  float hi = lo;
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a)
    if (a < A) {
      float safe = r.vis[a] > 0 ? r.q[a] : v;
      lo = az_minf(lo, safe);
      hi = az_maxf(hi, safe);
    }
  lo = az_minf(v, lo);
  hi = az_maxf(v, hi);
  float den = az_maxf(hi - lo, d.q_epsilon);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      float cm = r.vis[a] > 0 ? r.q[a] : lo;
      out[a] = (cm - lo) / den;
    }
}

// mctx_dist/_src/qtransforms.py:24-51
template <int AMAX>
AZ_DEV void az_qt_min_max(const az_search_desc& d, int A, const NodeRow<AMAX>& r,
                          float (&out)[AMAX]) {
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      float s = r.vis[a] > 0 ? r.q[a] : d.min_value;
      out[a] = (s - d.min_value) / (d.max_value - d.min_value);
    }
}

// mctx_dist/_src/qtransforms.py:90-149, 152-198
template <int AMAX>
AZ_DEV void az_qt_completed_mix(const az_search_desc& d, int A, const NodeRow<AMAX>& r,
                                float (&out)[AMAX]) {
  float raw = r.rawmean;
  float value;
  if (d.use_mixed_value) {
    float probs[AMAX];
    az_softmax<AMAX>(r.logits, A, probs);
    int32_t sum_visits = 0;
    float sum_probs = 0.0f;
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) {
        sum_visits += r.vis[a];
        probs[a] = az_maxf(AZ_FLT_TINY, probs[a]);
        sum_probs = sum_probs + (r.vis[a] > 0 ? probs[a] : 0.0f);
      }
    float weighted_q = 0.0f;
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) {
        float term = r.vis[a] > 0 ? (probs[a] * r.q[a]) / sum_probs : 0.0f;
        weighted_q = weighted_q + term;
      }
    value = (raw + (float)sum_visits * weighted_q) / (float)(sum_visits + 1);
  } else {
    value = raw;
  }
  float cq[AMAX];
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) cq[a] = r.vis[a] > 0 ? r.q[a] : value;
  if (d.rescale_values) {
    float mn = cq[0], mx = cq[0];
AZ_UNROLL_A
    for (int a = 1; a < AZ_TRIP_A; ++a)
      if (a < A) {
        mn = az_minf(mn, cq[a]);
        mx = az_maxf(mx, cq[a]);
      }
    float den = az_maxf(mx - mn, d.q_epsilon);
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) cq[a] = (cq[a] - mn) / den;
  }
  int32_t maxvisit = r.vis[0];
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a)
    if (a < A && r.vis[a] > maxvisit) maxvisit = r.vis[a];
  float visit_scale = d.maxvisit_init + (float)maxvisit;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) out[a] = (visit_scale * d.value_scale) * cq[a];
}

template <int AMAX>
AZ_DEV void az_qtransform(const az_search_desc& d, int A, const NodeRow<AMAX>& r,
                          float (&out)[AMAX]) {
  if (d.qtransform == AZ_QT_PARENT_SIBLINGS)
    az_qt_parent_siblings<AMAX>(d, A, r, out);
  else if (d.qtransform == AZ_QT_COMPLETED_MIX)
    az_qt_completed_mix<AMAX>(d, A, r, out);
  else
    az_qt_min_max<AMAX>(d, A, r, out);
}

// mctx_dist/_src/seq_halving.py:23-33
template <int AMAX>
AZ_DEV void az_score_considered(int32_t considered_visit, const float* gumbel,
                                const float (&logits)[AMAX], const float (&nq)[AMAX],
                                const int32_t (&vis)[AMAX], int A, float (&out)[AMAX]) {
  float mx = logits[0];
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a)
    if (a < A) mx = az_maxf(mx, logits[a]);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      float lg = logits[a] - mx;
      float penalty = vis[a] == considered_visit ? 0.0f : -AZ_INF;
      out[a] = az_maxf(-1e9f, (gumbel[a] + lg) + nq[a]) + penalty;
    }
}

// Everything a selection step needs besides the tree.
struct SearchCtx {
  az_search_desc d;
  az_search_inputs in;
};

// What only the ROOT selection reads besides the tree, for one lane: either the global
// arrays themselves (az_root_inputs) or per-lane copies in LDS (az_coop.h).
struct RootCache {
  const float* gumbel;        // [A], Gumbel variant
  const float* invalid;       // [A], meaningful when has_invalid
  const int32_t* table_row;   // row `table_nc` of the considered-visits table
  int32_t table_nc;           // -1: no row cached, index the global table
  bool has_invalid;
};

// mctx_dist/_src/action_selection.py:168-174: how many actions the root considers.
AZ_DEV int32_t az_num_considered(const az_search_desc& d, float nvalid) {
  const int32_t num_valid = (int32_t)nvalid;
  int32_t num_considered = d.max_num_considered < num_valid ? d.max_num_considered : num_valid;
  return num_considered < 0 ? 0 : num_considered;
}

// The root inputs of lane b straight from the global arrays.
AZ_DEV RootCache az_root_inputs(const SearchCtx& c, int A, int b) {
  RootCache rc;
  rc.has_invalid = c.in.invalid_actions != nullptr;
  rc.gumbel = c.in.root_gumbel + (size_t)b * A;  // only read by the Gumbel variant
  rc.invalid = c.in.invalid_actions + (size_t)b * A;
  rc.table_row = c.in.table;
  rc.table_nc = -1;
  return rc;
}

// mctx_dist/_src/action_selection.py:48-90 (muzero) and :93-135 (sampled): the scores
// value + policy before the tie-break noise is added.
template <int AMAX>
AZ_DEV void az_puct_scores(const SearchCtx& c, int A, const NodeRow<AMAX>& r, float (&score)[AMAX]) {
  const az_search_desc& d = c.d;
  int32_t node_visit = r.node_visit;
  float pb_c = d.pb_c_init + az_logf((((float)node_visit + d.pb_c_base) + 1.0f) / d.pb_c_base);
  float probs[AMAX], value_score[AMAX];
  if (d.variant == AZ_VARIANT_MUZERO || d.variant == AZ_VARIANT_STOCHASTIC) {
    az_softmax<AMAX>(r.logits, A, probs);
  } else {
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a) probs[a] = r.logits[a];
  }
  az_qtransform<AMAX>(d, A, r, value_score);
  float sq = az_sqrtf((float)node_visit);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      float policy_score = ((sq * pb_c) * probs[a]) / (float)(r.vis[a] + 1);
      score[a] = value_score[a] + policy_score;
    }
}

// argmax(score + noise) with the root-only invalid mask (action_selection.py:85-90).
template <int AMAX>
AZ_DEV int az_puct_argmax(int A, int depth, const float (&score)[AMAX], const float (&noise)[AMAX],
                          const RootCache& rc) {
  const bool use_mask = depth == 0 && rc.has_invalid;
  float to_argmax[AMAX];
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      float v = score[a] + noise[a];
      // action_selection.py:90: mask = root_invalid_actions * (depth == 0)
      bool masked = use_mask && rc.invalid[a] != 0.0f;
      to_argmax[a] = masked ? -AZ_INF : v;
    }
  return az_argmax<AMAX>(to_argmax, A);
}

template <int AMAX>
AZ_DEV int az_select_puct(const SearchCtx& c, int A, int depth, const NodeRow<AMAX>& r,
                          const float (&noise)[AMAX], const RootCache& rc) {
  float score[AMAX];
  az_puct_scores<AMAX>(c, A, r, score);
  return az_puct_argmax<AMAX>(A, depth, score, noise, rc);
}

// The tie-break noise of one selection step (action_selection.py:85, :130): 1e-7 * u.
// THREEFRY: (k0, k1) is the lane's key at this depth; it is advanced as simulate() does
// (search.py:155: rng_key, action_selection_key = split(rng_key)).
template <int AMAX>
AZ_DEV void az_puct_noise(const SearchCtx& c, int A, int b, int sim, int depth, uint32_t& k0,
                          uint32_t& k1, float (&noise)[AMAX]) {
  const az_search_desc& d = c.d;
  if (d.noise_mode == AZ_NOISE_THREEFRY) {
    uint32_t n0, n1, s0, s1;
    az_key_split(k0, k1, 0u, &n0, &n1);
    az_key_split(k0, k1, 1u, &s0, &s1);
    k0 = n0;
    k1 = n1;
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A)
        noise[a] = 1e-7f * az_uniform_from_bits(az_random_bits32(s0, s1, (uint32_t)a), 0.0f, 1.0f);
  } else if (d.noise_mode == AZ_NOISE_INJECTED) {
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) {
        float u = depth < d.noise_depth
                      ? c.in.noise[(((size_t)b * d.S + sim) * d.noise_depth + depth) * A + a]
                      : 0.0f;
        noise[a] = 1e-7f * u;
      }
  } else {
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a) noise[a] = 1e-7f * 0.0f;
  }
}

// mctx_dist/_src/action_selection.py:148-200 after the qtransform: `cq` are the completed
// Q-values of the root row.
template <int AMAX>
AZ_DEV int az_gumbel_root_tail(const SearchCtx& c, int A, const NodeRow<AMAX>& r, const float (&cq)[AMAX],
                               const RootCache& rc) {
  const az_search_desc& d = c.d;
  float score[AMAX];
  float nvalid = 0.0f;
  int32_t sim_index = 0;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      nvalid = nvalid + (1.0f - (rc.has_invalid ? rc.invalid[a] : 0.0f));
      sim_index += r.vis[a];
    }
  const int32_t num_considered = az_num_considered(d, nvalid);
  if (sim_index > d.S - 1) sim_index = d.S - 1;
  int32_t considered_visit;
  if (rc.table_nc == num_considered)
    considered_visit = rc.table_row[sim_index];
  else
    considered_visit = c.in.table[(size_t)num_considered * d.S + sim_index];
  az_score_considered<AMAX>(considered_visit, rc.gumbel, r.logits, cq, r.vis, A, score);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A && rc.has_invalid && rc.invalid[a] != 0.0f) score[a] = -AZ_INF;
  return az_argmax<AMAX>(score, A);
}

template <int AMAX>
AZ_DEV int az_select_gumbel_root(const SearchCtx& c, int A, const NodeRow<AMAX>& r,
                                 const RootCache& rc) {
  float cq[AMAX];
  az_qtransform<AMAX>(c.d, A, r, cq);
  return az_gumbel_root_tail<AMAX>(c, A, r, cq, rc);
}

// mctx_dist/_src/action_selection.py:203-240, 256-277 after the qtransform.
template <int AMAX>
AZ_DEV int az_gumbel_interior_tail(int A, const NodeRow<AMAX>& r, const float (&cq)[AMAX]) {
  float x[AMAX], probs[AMAX], to_argmax[AMAX];
  int32_t sum_visits = 0;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      x[a] = r.logits[a] + cq[a];
      sum_visits += r.vis[a];
    }
  az_softmax<AMAX>(x, A, probs);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) to_argmax[a] = probs[a] - (float)r.vis[a] / (float)(1 + sum_visits);
  return az_argmax<AMAX>(to_argmax, A);
}

template <int AMAX>
AZ_DEV int az_select_gumbel_interior(const SearchCtx& c, int A, const NodeRow<AMAX>& r) {
  float cq[AMAX];
  az_qtransform<AMAX>(c.d, A, r, cq);
  return az_gumbel_interior_tail<AMAX>(A, r, cq);
}

// mctx_dist/_src/search.py:127-187 with the root/interior switch of
// action_selection.py:28-45.  (k0,k1) = split(simulate_key, B)[b].
template <int AMAX, int KMAX>
AZ_DEV void az_simulate_lane(const SearchCtx& c, const Lane<AMAX, KMAX>& l, int b, int sim,
                             uint32_t k0, uint32_t k1, int& parent, int& action, int& next,
                             int& depth_out) {
  const az_search_desc& d = c.d;
  const int A = l.A;
  int node = AZ_ROOT, depth = 0, act = -1, nxt = AZ_ROOT;
  for (;;) {
    node = nxt;
    NodeRow<AMAX> r;
    az_load_row<AMAX, KMAX>(l, node, r);
    if (d.variant == AZ_VARIANT_STOCHASTIC) {
      // policies.py:676-708: the node's kind decides.  The key is split at every level
      // (search.py:155) whether or not the selection draws from it.
      const int AD = d.num_decision_actions;
      float noise[AMAX];
      az_puct_noise<AMAX>(c, AD, b, sim, depth, k0, k1, noise);
      if (c.in.is_decision[(size_t)b * (d.S + 1) + node] != 0) {
        // muzero_action_selection on the tree masked to its first AD columns (policies.py:658-673):
        // a prefix of the row, so the rule runs on the row as loaded with AD in place of A
        // (the caller's invalid_actions are [B, AD]: search() gets them unpadded, policies.py:472)
        act = az_select_puct<AMAX>(c, AD, depth, r, noise, az_root_inputs(c, AD, b));
      } else {
        // policies.py:684-694: argmax(softmax(chance logits) / (visits + 1)) over the other columns
        float x[AMAX], p[AMAX];
AZ_UNROLL_A
        for (int a = 0; a < AZ_TRIP_A; ++a) x[a] = a < A - AD ? l.clog[node * A + AD + a] : 0.0f;
        az_softmax<AMAX>(x, A - AD, p);
AZ_UNROLL_A
        for (int a = 0; a < AZ_TRIP_A; ++a)
          if (a < A - AD) p[a] = p[a] / (float)(l.cvis[node * A + AD + a] + 1);
        act = az_argmax<AMAX>(p, A - AD) + AD;
      }
    } else if (d.variant != AZ_VARIANT_GUMBEL) {
      float noise[AMAX];
      az_puct_noise<AMAX>(c, A, b, sim, depth, k0, k1, noise);
      act = az_select_puct<AMAX>(c, A, depth, r, noise, az_root_inputs(c, A, b));
    } else if (depth == 0) {
      act = az_select_gumbel_root<AMAX>(c, A, r, az_root_inputs(c, A, b));
    } else {
      act = az_select_gumbel_interior<AMAX>(c, A, r);
    }
    nxt = l.cidx[node * A + act];
    depth += 1;
    if (!(nxt != AZ_UNVISITED && depth < d.max_depth)) break;
  }
  parent = node;
  action = act;
  next = nxt;
  depth_out = depth;
}

// mctx_dist/_src/search.py:231-244, 304-342: the tree writes of expand.
template <int AMAX, int KMAX>
AZ_DEV void az_expand_write(const Lane<AMAX, KMAX>& l, int parent, int action, int node,
                            const float* logits, const float* value, float reward, float discount) {
  const int A = l.A, K = l.K;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) l.clog[node * A + a] = logits[a];
  if (KMAX == 1) {
    l.raw[node] = value[0];
    l.nval[node] = value[0];
  } else {
    for (int k = 0; k < K; ++k) {
      l.raw[(size_t)node * K + k] = value[k];
      l.nval[(size_t)node * K + k] = value[k];
    }
  }
  l.nv[node] = l.nv[node] + 1;
  l.cidx[parent * A + action] = node;
  l.crew[parent * A + action] = reward;
  l.cdisc[parent * A + action] = discount;
  l.par[node] = parent;
  l.afp[node] = action;
}

// mctx_dist/_src/search.py:247-292
template <int AMAX, int KMAX>
AZ_DEV void az_backward_lane(const Lane<AMAX, KMAX>& l, int leaf) {
  const int A = l.A, K = l.K;
  float lv[KMAX];
AZ_UNROLL_K
  for (int k = 0; k < AZ_TRIP_K; ++k) lv[k] = k < K ? l.nval[(size_t)leaf * K + k] : 0.0f;
  int index = leaf;
  while (index != AZ_ROOT) {
    int parent = l.par[index];
    int32_t count = l.nv[parent];
    int action = l.afp[index];
    float reward = l.crew[parent * A + action];
    float disc = l.cdisc[parent * A + action];
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k)
      if (k < K) {
        lv[k] = reward + disc * lv[k];
        float pv = (l.nval[(size_t)parent * K + k] * (float)count + lv[k]) / ((float)count + 1.0f);
        l.nval[(size_t)parent * K + k] = pv;
        l.cval[((size_t)parent * A + action) * K + k] = l.nval[(size_t)index * K + k];
      }
    l.cvis[parent * A + action] = l.cvis[parent * A + action] + 1;
    l.nv[parent] = count + 1;
    index = parent;
  }
}

// per-simulation keys: rng, simulate_key, expand_key = split(rng, 3)  (search.py:90)
AZ_DEV void az_next_sim_key(uint32_t& r0, uint32_t& r1, uint32_t& s0, uint32_t& s1) {
  uint32_t n0, n1;
  az_key_split(r0, r1, 1u, &s0, &s1);
  az_key_split(r0, r1, 0u, &n0, &n1);
  r0 = n0;
  r1 = n1;
}

#endif  // AZ_DEVICE_H_
