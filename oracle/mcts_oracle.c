/* mcts_oracle.c -- CPU checker for the batched-MCTS hot path (TEST
 * INFRASTRUCTURE ONLY; see mcts_oracle.h for who may load it).
 *
 * A literal, one-lane-at-a-time restatement in plain C of the reference's
 * search.  Every function cites the reference lines it follows (paths are
 * relative to the reference checkout).  All arithmetic is binary32 in the
 * order the reference writes it; the elementary functions come from the
 * numerics specification az_math.h so that a GPU run can be compared with
 * this checker bit for bit.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and cannot be
 * run here; the pins are the hand-derived known-answer cases in
 * tests/test_oracle_kats.py and the independent NumPy restatement
 * oracle/mctx_numpy.py.
 */
#include "mcts_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../alphazero_amd/csrc/az_math.h"

#define ROOT_INDEX 0
#define NO_PARENT (-1)
#define UNVISITED (-1)

/* ------------------------------------------------------------------ */
/* small helpers                                                        */

/* jnp.argmax: first maximal element, NaN counts as maximal. */
static int oz_argmax(const float* x, int n) {
  int best = 0;
  float bv = x[0];
  for (int i = 1; i < n; ++i) {
    if ((x[i] > bv) || (x[i] != x[i] && bv == bv)) {
      best = i;
      bv = x[i];
    }
  }
  return best;
}

/* jax.nn.softmax: exp(x - max) / sum(exp(x - max)), sum left to right. */
static void oz_softmax(const float* x, int n, float* out) {
  float m = x[0];
  for (int i = 1; i < n; ++i) m = az_maxf(m, x[i]);
  float s = 0.0f;
  for (int i = 0; i < n; ++i) {
    out[i] = az_expf(x[i] - m);
    s = s + out[i];
  }
  for (int i = 0; i < n; ++i) out[i] = out[i] / s;
}

/* mctx_dist/_src/action_selection.py:243-253 */
static int oz_masked_argmax(const float* x, const float* invalid, int n) {
  float t[OZ_MAX_A];
  for (int i = 0; i < n; ++i)
    t[i] = (invalid != NULL && invalid[i] != 0.0f) ? -AZ_INF : x[i];
  return oz_argmax(t, n);
}

/* mctx_dist/_src/seq_halving.py:36-70 */
static void oz_considered_sequence(int m, int S, int32_t* seq) {
  if (m <= 1) {
    for (int i = 0; i < S; ++i) seq[i] = i;
    return;
  }
  int log2max = 0;
  while ((1 << log2max) < m) ++log2max;
  int32_t* visits = (int32_t*)calloc((size_t)m, sizeof(int32_t));
  int n = 0, num_considered = m;
  while (n < S) {
    int extra = S / (log2max * num_considered);
    if (extra < 1) extra = 1;
    for (int e = 0; e < extra; ++e) {
      for (int i = 0; i < num_considered; ++i)
        if (n < S) seq[n++] = visits[i];
      for (int i = 0; i < num_considered; ++i) visits[i] += 1;
    }
    num_considered = num_considered / 2;
    if (num_considered < 2) num_considered = 2;
  }
  free(visits);
}

/* mctx_dist/_src/seq_halving.py:73-88: table[(m+1) x S]. */
void oz_considered_table(int max_m, int S, int32_t* table) {
  for (int m = 0; m <= max_m; ++m) oz_considered_sequence(m, S, table + (size_t)m * S);
}

/* ------------------------------------------------------------------ */
/* one lane's view of the tree                                          */

typedef struct {
  int N, A, K;
  int32_t *node_visits, *parents, *afp, *cindex, *cvisits;
  float *raw, *nval, *clogits, *crew, *cdisc, *cval;
  float* emb_state;
  int32_t* emb_time;
} lane_t;

static lane_t oz_lane(const oz_tree* t, const oz_cfg* c, int b) {
  lane_t l;
  int N = c->S + 1, A = c->A, K = c->K;
  l.N = N;
  l.A = A;
  l.K = K;
  l.node_visits = t->node_visits + (size_t)b * N;
  l.parents = t->parents + (size_t)b * N;
  l.afp = t->action_from_parent + (size_t)b * N;
  l.cindex = t->children_index + (size_t)b * N * A;
  l.cvisits = t->children_visits + (size_t)b * N * A;
  l.raw = t->raw_values + (size_t)b * N * K;
  l.nval = t->node_values + (size_t)b * N * K;
  l.clogits = t->children_prior_logits + (size_t)b * N * A;
  l.crew = t->children_rewards + (size_t)b * N * A;
  l.cdisc = t->children_discounts + (size_t)b * N * A;
  l.cval = t->children_values + (size_t)b * N * A * K;
  l.emb_state = t->emb_state ? t->emb_state + (size_t)b * N * 4 : NULL;
  l.emb_time = t->emb_time ? t->emb_time + (size_t)b * N : NULL;
  return l;
}

static float oz_mean(const float* x, int K) {
  float s = 0.0f;
  for (int k = 0; k < K; ++k) s = s + x[k];
  return s / (float)K;
}

/* mctx_dist/_src/tree.py:142-147 followed by the mean over atoms that every
 * qtransform applies first (qtransforms.py:46-47, 72-73, 124-125). */
static void oz_mean_qvalues(const lane_t* l, int node, float* q) {
  float tmp[OZ_MAX_K];
  for (int a = 0; a < l->A; ++a) {
    float r = l->crew[node * l->A + a], d = l->cdisc[node * l->A + a];
    const float* cv = l->cval + ((size_t)node * l->A + a) * l->K;
    for (int k = 0; k < l->K; ++k) tmp[k] = r + d * cv[k];
    q[a] = oz_mean(tmp, l->K);
  }
}

/* mctx_dist/_src/qtransforms.py:54-87 */
static void oz_qt_parent_siblings(const lane_t* l, const oz_cfg* c, int node, float* out) {
  float q[OZ_MAX_A];
  int A = l->A;
  oz_mean_qvalues(l, node, q);
  const int32_t* vis = l->cvisits + node * A;
  float v = oz_mean(l->nval + (size_t)node * l->K, l->K);
  float lo = 0, hi = 0;
  for (int a = 0; a < A; ++a) {
    float safe = vis[a] > 0 ? q[a] : v;
    if (a == 0) {
      lo = safe;
      hi = safe;
    } else {
      lo = az_minf(lo, safe);
      hi = az_maxf(hi, safe);
    }
  }
  lo = az_minf(v, lo);
  hi = az_maxf(v, hi);
  float den = az_maxf(hi - lo, c->q_epsilon);
  for (int a = 0; a < A; ++a) {
    float cm = vis[a] > 0 ? q[a] : lo;
    out[a] = (cm - lo) / den;
  }
}

/* mctx_dist/_src/qtransforms.py:24-51 */
static void oz_qt_min_max(const lane_t* l, const oz_cfg* c, int node, float* out) {
  float q[OZ_MAX_A];
  int A = l->A;
  oz_mean_qvalues(l, node, q);
  const int32_t* vis = l->cvisits + node * A;
  for (int a = 0; a < A; ++a) {
    float s = vis[a] > 0 ? q[a] : c->min_value;
    out[a] = (s - c->min_value) / (c->max_value - c->min_value);
  }
}

/* mctx_dist/_src/qtransforms.py:90-149, 152-198 */
static void oz_qt_completed_mix(const lane_t* l, const oz_cfg* c, int node, float* out) {
  float q[OZ_MAX_A], probs[OZ_MAX_A];
  int A = l->A;
  oz_mean_qvalues(l, node, q);
  const int32_t* vis = l->cvisits + node * A;
  float raw = oz_mean(l->raw + (size_t)node * l->K, l->K);
  oz_softmax(l->clogits + node * A, A, probs);
  float value;
  if (c->use_mixed_value) {
    int32_t sum_visits = 0;
    for (int a = 0; a < A; ++a) sum_visits += vis[a];
    float sum_probs = 0.0f;
    for (int a = 0; a < A; ++a) {
      probs[a] = az_maxf(AZ_FLT_TINY, probs[a]);
      sum_probs = sum_probs + (vis[a] > 0 ? probs[a] : 0.0f);
    }
    float weighted_q = 0.0f;
    for (int a = 0; a < A; ++a) {
      float term = vis[a] > 0 ? (probs[a] * q[a]) / (vis[a] > 0 ? sum_probs : 1.0f) : 0.0f;
      weighted_q = weighted_q + term;
    }
    value = (raw + (float)sum_visits * weighted_q) / (float)(sum_visits + 1);
  } else {
    value = raw;
  }
  float cq[OZ_MAX_A];
  for (int a = 0; a < A; ++a) cq[a] = vis[a] > 0 ? q[a] : value;
  if (c->rescale_values) {
    float mn = cq[0], mx = cq[0];
    for (int a = 1; a < A; ++a) {
      mn = az_minf(mn, cq[a]);
      mx = az_maxf(mx, cq[a]);
    }
    float den = az_maxf(mx - mn, c->q_epsilon);
    for (int a = 0; a < A; ++a) cq[a] = (cq[a] - mn) / den;
  }
  int32_t maxvisit = vis[0];
  for (int a = 1; a < A; ++a)
    if (vis[a] > maxvisit) maxvisit = vis[a];
  float visit_scale = c->maxvisit_init + (float)maxvisit;
  for (int a = 0; a < A; ++a) out[a] = (visit_scale * c->value_scale) * cq[a];
}

static void oz_qtransform(const lane_t* l, const oz_cfg* c, int node, float* out) {
  if (c->qtransform == OZ_QT_PARENT_SIBLINGS)
    oz_qt_parent_siblings(l, c, node, out);
  else if (c->qtransform == OZ_QT_COMPLETED_MIX)
    oz_qt_completed_mix(l, c, node, out);
  else
    oz_qt_min_max(l, c, node, out);
}

/* mctx_dist/_src/seq_halving.py:23-33 */
static void oz_score_considered(int32_t considered_visit, const float* gumbel,
                                const float* logits, const float* nq,
                                const int32_t* vis, int A, float* out) {
  float mx = logits[0];
  for (int a = 1; a < A; ++a) mx = az_maxf(mx, logits[a]);
  for (int a = 0; a < A; ++a) {
    float lg = logits[a] - mx;
    float penalty = vis[a] == considered_visit ? 0.0f : -AZ_INF;
    out[a] = az_maxf(-1e9f, (gumbel[a] + lg) + nq[a]) + penalty;
  }
}

/* per-search constants shared by the lanes */
typedef struct {
  const oz_cfg* c;
  const float* invalid;      /* [B,A] root_invalid_actions (zeros when none given) */
  const float* root_gumbel;  /* [B,A] (gumbel variant) */
  const int32_t* table;      /* [(m+1),S] */
  const float* noise;        /* injected tie-break noise [B,S,Dn,A] or NULL */
  const uint8_t* is_decision; /* [B,N] node kinds (stochastic variant) */
} search_t;

/* mctx_dist/_src/action_selection.py:48-90 (muzero) and :93-135 (sampled). */
static int oz_select_puct(const search_t* s, const lane_t* l, int b, int node, int depth,
                          const float* noise) {
  const oz_cfg* c = s->c;
  int A = l->A;
  const int32_t* vis = l->cvisits + node * A;
  int32_t node_visit = l->node_visits[node];
  float pb_c = c->pb_c_init + az_logf((((float)node_visit + c->pb_c_base) + 1.0f) / c->pb_c_base);
  float probs[OZ_MAX_A], value_score[OZ_MAX_A], to_argmax[OZ_MAX_A];
  if (c->variant == OZ_VARIANT_MUZERO || c->variant == OZ_VARIANT_STOCHASTIC)
    oz_softmax(l->clogits + node * A, A, probs);
  else
    for (int a = 0; a < A; ++a) probs[a] = l->clogits[node * A + a];
  oz_qtransform(l, c, node, value_score);
  float sq = az_sqrtf((float)node_visit);
  for (int a = 0; a < A; ++a) {
    float policy_score = ((sq * pb_c) * probs[a]) / (float)(vis[a] + 1);
    to_argmax[a] = (value_score[a] + policy_score) + noise[a];
  }
  float mask[OZ_MAX_A];
  for (int a = 0; a < A; ++a) mask[a] = s->invalid[(size_t)b * A + a] * (depth == 0 ? 1.0f : 0.0f);
  return oz_masked_argmax(to_argmax, mask, A);
}

/* mctx_dist/_src/action_selection.py:148-200 */
static int oz_select_gumbel_root(const search_t* s, const lane_t* l, int b, int node) {
  const oz_cfg* c = s->c;
  int A = l->A;
  const int32_t* vis = l->cvisits + node * A;
  float cq[OZ_MAX_A], score[OZ_MAX_A];
  oz_qtransform(l, c, node, cq);
  const float* inv = s->invalid + (size_t)b * A;
  float nv = 0.0f;
  for (int a = 0; a < A; ++a) nv = nv + (1.0f - inv[a]);
  int32_t num_valid = (int32_t)nv;
  int32_t num_considered = c->max_num_considered < num_valid ? c->max_num_considered : num_valid;
  int32_t sim_index = 0;
  for (int a = 0; a < A; ++a) sim_index += vis[a];
  /* out-of-range gathers clamp */
  if (num_considered < 0) num_considered = 0;
  if (sim_index > c->S - 1) sim_index = c->S - 1;
  int32_t considered_visit = s->table[(size_t)num_considered * c->S + sim_index];
  oz_score_considered(considered_visit, s->root_gumbel + (size_t)b * A,
                      l->clogits + node * A, cq, vis, A, score);
  return oz_masked_argmax(score, inv, A);
}

/* mctx_dist/_src/action_selection.py:203-240, 256-277 */
static int oz_select_gumbel_interior(const search_t* s, const lane_t* l, int node) {
  int A = l->A;
  const int32_t* vis = l->cvisits + node * A;
  float cq[OZ_MAX_A], x[OZ_MAX_A], probs[OZ_MAX_A], to_argmax[OZ_MAX_A];
  oz_qtransform(l, s->c, node, cq);
  for (int a = 0; a < A; ++a) x[a] = l->clogits[node * A + a] + cq[a];
  oz_softmax(x, A, probs);
  int32_t sum_visits = 0;
  for (int a = 0; a < A; ++a) sum_visits += vis[a];
  for (int a = 0; a < A; ++a)
    to_argmax[a] = probs[a] - (float)vis[a] / (float)(1 + sum_visits);
  return oz_argmax(to_argmax, A);
}

/* mctx_dist/_src/policies.py:676-708: the node's kind picks the rule.  Decision nodes run
 * muzero_action_selection on the tree masked to its first AD columns (policies.py:658-673),
 * restated here by copying the node's AD columns into a one-node lane of width AD; chance
 * nodes take argmax(softmax(chance logits) / (visits + 1)) over the other columns. */
static int oz_select_stochastic(const search_t* s, const lane_t* l, int b, int node, int depth,
                                const float* noise) {
  const oz_cfg* c = s->c;
  int A = l->A, K = l->K, AD = c->num_decision_actions, CN = A - AD;
  if (s->is_decision[(size_t)b * l->N + node]) {
    int32_t vis[OZ_MAX_A], idx[OZ_MAX_A];
    float lg[OZ_MAX_A], rw[OZ_MAX_A], dc[OZ_MAX_A], cv[OZ_MAX_A * OZ_MAX_K], inv[OZ_MAX_A];
    for (int a = 0; a < AD; ++a) {
      vis[a] = l->cvisits[node * A + a];
      idx[a] = l->cindex[node * A + a];
      lg[a] = l->clogits[node * A + a];
      rw[a] = l->crew[node * A + a];
      dc[a] = l->cdisc[node * A + a];
      for (int k = 0; k < K; ++k) cv[a * K + k] = l->cval[((size_t)node * A + a) * K + k];
      inv[a] = s->invalid[(size_t)b * AD + a]; /* search() gets [B,AD] (policies.py:472) */
    }
    lane_t m = *l;
    m.N = 1;
    m.A = AD;
    m.node_visits = l->node_visits + node;
    m.raw = l->raw + (size_t)node * K;
    m.nval = l->nval + (size_t)node * K;
    m.cvisits = vis;
    m.cindex = idx;
    m.clogits = lg;
    m.crew = rw;
    m.cdisc = dc;
    m.cval = cv;
    search_t s2 = *s;
    s2.invalid = inv;
    return oz_select_puct(&s2, &m, 0, 0, depth, noise);
  }
  float p[OZ_MAX_A];
  oz_softmax(l->clogits + node * A + AD, CN, p);
  for (int a = 0; a < CN; ++a) p[a] = p[a] / (float)(l->cvisits[node * A + AD + a] + 1);
  return oz_argmax(p, CN) + AD;
}

/* mctx_dist/_src/search.py:127-187 with the root/interior switch of
 * action_selection.py:28-45.  `lane_key` is split(simulate_key, B)[b]. */
static void oz_simulate(const search_t* s, const lane_t* l, int b, int sim,
                        uint32_t lane_k0, uint32_t lane_k1, int* parent_out,
                        int* action_out, int* next_out, int* depth_out) {
  const oz_cfg* c = s->c;
  int A = l->A;
  int node = ROOT_INDEX, depth = 0, action = NO_PARENT, nxt = ROOT_INDEX;
  uint32_t k0 = lane_k0, k1 = lane_k1;
  for (;;) {
    node = nxt;
    float noise[OZ_MAX_A];
    if (c->variant != OZ_VARIANT_GUMBEL) {
      /* rng_key, action_selection_key = split(rng_key)  (search.py:155) */
      uint32_t n0, n1, s0, s1;
      az_key_split(k0, k1, 0u, &n0, &n1);
      az_key_split(k0, k1, 1u, &s0, &s1);
      k0 = n0;
      k1 = n1;
      /* (stochastic: uniform(key, [AD]) -- the masked tree's width, policies.py:703-704) */
      const int An = c->variant == OZ_VARIANT_STOCHASTIC ? c->num_decision_actions : A;
      for (int a = 0; a < An; ++a) {
        float u;
        if (c->noise_mode == OZ_NOISE_THREEFRY)
          u = az_uniform_from_bits(az_random_bits32(s0, s1, (uint32_t)a), 0.0f, 1.0f);
        else if (c->noise_mode == OZ_NOISE_INJECTED)
          u = depth < c->noise_depth
                  ? s->noise[(((size_t)b * c->S + sim) * c->noise_depth + depth) * An + a]
                  : 0.0f;
        else
          u = 0.0f;
        noise[a] = 1e-7f * u; /* action_selection.py:85-86 */
      }
      action = c->variant == OZ_VARIANT_STOCHASTIC ? oz_select_stochastic(s, l, b, node, depth, noise)
                                                    : oz_select_puct(s, l, b, node, depth, noise);
    } else if (depth == 0) {
      action = oz_select_gumbel_root(s, l, b, node);
    } else {
      action = oz_select_gumbel_interior(s, l, node);
    }
    nxt = l->cindex[node * A + action];
    depth += 1;
    if (!(nxt != UNVISITED && depth < c->max_depth)) break;
  }
  *parent_out = node;
  *action_out = action;
  *next_out = nxt;
  *depth_out = depth;
}

/* mctx_dist/_src/search.py:190-244, 304-342: write the evaluated node and
 * the edge that leads to it. */
static void oz_expand_write(const lane_t* l, int parent, int action, int node,
                            const float* logits, const float* value, float reward,
                            float discount) {
  int A = l->A, K = l->K;
  for (int a = 0; a < A; ++a) l->clogits[node * A + a] = logits[a];
  for (int k = 0; k < K; ++k) {
    l->raw[(size_t)node * K + k] = value[k];
    l->nval[(size_t)node * K + k] = value[k];
  }
  l->node_visits[node] = l->node_visits[node] + 1;
  l->cindex[parent * A + action] = node;
  l->crew[parent * A + action] = reward;
  l->cdisc[parent * A + action] = discount;
  l->parents[node] = parent;
  l->afp[node] = action;
}

/* mctx_dist/_src/search.py:247-292 */
static void oz_backward(const lane_t* l, int leaf) {
  int A = l->A, K = l->K;
  float lv[OZ_MAX_K];
  for (int k = 0; k < K; ++k) lv[k] = l->nval[(size_t)leaf * K + k];
  int index = leaf;
  while (index != ROOT_INDEX) {
    int parent = l->parents[index];
    int32_t count = l->node_visits[parent];
    int action = l->afp[index];
    float reward = l->crew[parent * A + action];
    float disc = l->cdisc[parent * A + action];
    for (int k = 0; k < K; ++k) {
      lv[k] = reward + disc * lv[k];
      float pv = (l->nval[(size_t)parent * K + k] * (float)count + lv[k]) / ((float)count + 1.0f);
      l->nval[(size_t)parent * K + k] = pv;
      l->cval[((size_t)parent * A + action) * K + k] = l->nval[(size_t)index * K + k];
    }
    l->cvisits[parent * A + action] = l->cvisits[parent * A + action] + 1;
    l->node_visits[parent] = count + 1;
    index = parent;
  }
}

/* mctx_dist/_src/search.py:345-387 */
void oz_tree_init(const oz_cfg* c, const oz_tree* t, const float* root_logits,
                  const float* root_value, const float* root_state,
                  const int32_t* root_time) {
  int B = c->B, N = c->S + 1, A = c->A, K = c->K;
  for (size_t i = 0; i < (size_t)B * N; ++i) {
    t->node_visits[i] = 0;
    t->parents[i] = NO_PARENT;
    t->action_from_parent[i] = NO_PARENT;
  }
  for (size_t i = 0; i < (size_t)B * N * K; ++i) {
    t->raw_values[i] = 0.0f;
    t->node_values[i] = 0.0f;
  }
  for (size_t i = 0; i < (size_t)B * N * A; ++i) {
    t->children_index[i] = UNVISITED;
    t->children_prior_logits[i] = 0.0f;
    t->children_visits[i] = 0;
    t->children_rewards[i] = 0.0f;
    t->children_discounts[i] = 0.0f;
  }
  for (size_t i = 0; i < (size_t)B * N * A * K; ++i) t->children_values[i] = 0.0f;
  if (t->emb_state) memset(t->emb_state, 0, sizeof(float) * (size_t)B * N * 4);
  if (t->emb_time) memset(t->emb_time, 0, sizeof(int32_t) * (size_t)B * N);
  for (int b = 0; b < B; ++b) {
    lane_t l = oz_lane(t, c, b);
    for (int a = 0; a < A; ++a) l.clogits[a] = root_logits[(size_t)b * A + a];
    for (int k = 0; k < K; ++k) {
      l.raw[k] = root_value[(size_t)b * K + k];
      l.nval[k] = root_value[(size_t)b * K + k];
    }
    l.node_visits[0] = l.node_visits[0] + 1;
    if (l.emb_state && root_state)
      for (int i = 0; i < 4; ++i) l.emb_state[i] = root_state[(size_t)b * 4 + i];
    if (l.emb_time && root_time) l.emb_time[0] = root_time[b];
  }
}

/* ------------------------------------------------------------------ */
/* environments (gymnax classic_control, as modified by the reference)  */

typedef struct {
  float s[4];
  int32_t time;
} env_state_t;

typedef struct {
  float obs[6];
  float reward, terminated, truncated;
} env_out_t;

static int oz_obs_dim(int env_id) {
  return env_id == OZ_ENV_CARTPOLE ? 4 : env_id == OZ_ENV_MOUNTAINCAR ? 2 : 6;
}
int oz_env_obs_dim(int env_id) { return oz_obs_dim(env_id); }
int oz_env_num_actions(int env_id) { return env_id == OZ_ENV_CARTPOLE ? 2 : 3; }

/* gymnax/gymnax/environments/classic_control/cartpole.py:114-135 */
static void cartpole_is_terminal(const env_state_t* st, float* terminated, float* truncated) {
  const float x_threshold = 2.4f, theta_threshold = 0.20943951023931953f;
  int done1 = (st->s[0] < -x_threshold) || (st->s[0] > x_threshold);
  int done2 = (st->s[2] < -theta_threshold) || (st->s[2] > theta_threshold);
  *terminated = (done1 || done2) ? 1.0f : 0.0f;
  *truncated = st->time >= 50000 ? 1.0f - *terminated : 0.0f;
}

/* cartpole.py:49-90 */
static void cartpole_step(env_state_t* st, int action, env_out_t* o) {
  const float gravity = 9.8f, masspole = 0.1f, total_mass = 1.1f, length = 0.5f,
              polemass_length = 0.05f, force_mag = 10.0f, tau = 0.02f;
  float prev_terminal, prev_truncated;
  cartpole_is_terminal(st, &prev_terminal, &prev_truncated);
  float x = st->s[0], x_dot = st->s[1], theta = st->s[2], theta_dot = st->s[3];
  float force = force_mag * (float)action - force_mag * (float)(1 - action);
  float costheta = az_cosf(theta), sintheta = az_sinf(theta);
  float temp = (force + (polemass_length * (theta_dot * theta_dot)) * sintheta) / total_mass;
  float thetaacc = (gravity * sintheta - costheta * temp) /
                   (length * (1.3333333333333333f - (masspole * (costheta * costheta)) / total_mass));
  float xacc = temp - ((polemass_length * thetaacc) * costheta) / total_mass;
  st->s[0] = x + tau * x_dot;
  st->s[1] = x_dot + tau * xacc;
  st->s[2] = theta + tau * theta_dot;
  st->s[3] = theta_dot + tau * thetaacc;
  st->time = st->time + 1;
  o->reward = 1.0f - az_maxf(prev_terminal, prev_truncated);
  cartpole_is_terminal(st, &o->terminated, &o->truncated);
  for (int i = 0; i < 4; ++i) o->obs[i] = st->s[i];
}

/* mountain_car.py:43-72, 86-94 */
static void mountaincar_step(env_state_t* st, int action, env_out_t* o) {
  const float min_position = -1.2f, max_position = 0.6f, max_speed = 0.07f,
              goal_position = 0.5f, goal_velocity = 0.0f, force = 0.001f, gravity = 0.0025f;
  float position = st->s[0], velocity = st->s[1];
  velocity = (velocity + (float)(action - 1) * force) - az_cosf(3.0f * position) * gravity;
  velocity = az_minf(az_maxf(velocity, -max_speed), max_speed);
  position = position + velocity;
  position = az_minf(az_maxf(position, min_position), max_position);
  int stop = (position == min_position) && (velocity < 0.0f);
  velocity = velocity * (float)(1 - stop);
  st->s[0] = position;
  st->s[1] = velocity;
  st->s[2] = 0.0f;
  st->s[3] = 0.0f;
  st->time = st->time + 1;
  o->reward = -1.0f;
  o->terminated = ((position >= goal_position) && (velocity >= goal_velocity)) ? 1.0f : 0.0f;
  o->truncated = st->time >= 5000 ? 1.0f : 0.0f;
  o->obs[0] = position;
  o->obs[1] = velocity;
}

/* acrobot.py:202-233 */
static void acrobot_dsdt(const float* s, float* out) {
  const float m1 = 1.0f, m2 = 1.0f, l1 = 1.0f, lc1 = 0.5f, lc2 = 0.5f, I1 = 1.0f, I2 = 1.0f,
              g = 9.8f, half_pi = 1.5707963267948966f;
  float a = s[4];
  float theta1 = s[0], theta2 = s[1], dtheta1 = s[2], dtheta2 = s[3];
  float c2 = az_cosf(theta2), s2 = az_sinf(theta2);
  float d1 = ((m1 * (lc1 * lc1) + m2 * (((l1 * l1) + (lc2 * lc2)) + (((2.0f * l1) * lc2) * c2))) + I1) + I2;
  float d2 = m2 * ((lc2 * lc2) + (l1 * lc2) * c2) + I2;
  float phi2 = ((m2 * lc2) * g) * az_cosf((theta1 + theta2) - half_pi);
  float phi1 = ((((((-m2) * l1) * lc2) * (dtheta2 * dtheta2)) * s2 -
                 (((((2.0f * m2) * l1) * lc2) * dtheta2) * dtheta1) * s2) +
                ((m1 * lc1 + m2 * l1) * g) * az_cosf(theta1 - half_pi)) +
               phi2;
  float ddtheta2 = (((a + (d2 / d1) * phi1) - (((m2 * l1) * lc2) * (dtheta1 * dtheta1)) * s2) - phi2) /
                   ((m2 * (lc2 * lc2) + I2) - (d2 * d2) / d1);
  float ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
  out[0] = dtheta1;
  out[1] = dtheta2;
  out[2] = ddtheta1;
  out[3] = ddtheta2;
  out[4] = 0.0f;
}

/* acrobot.py:236-247 */
static float acrobot_wrap(float x, float m, float M) {
  float diff = M - m;
  float go_up = x < m ? 1.0f : 0.0f;
  float go_down = x >= M ? 1.0f : 0.0f;
  float how_often = go_up * __builtin_ceilf((m - x) / diff) +
                    go_down * __builtin_floorf((x - M) / diff + 1.0f);
  return (x - (how_often * diff) * go_down) + (how_often * diff) * go_up;
}

/* acrobot.py:52-106, 137-147, 250-258 */
static void acrobot_step(env_state_t* st, int action, env_out_t* o) {
  const float dt = 0.2f, pi = 3.141592653589793f, max_vel_1 = 12.566370614359172f,
              max_vel_2 = 28.274333882308138f;
  float torque = (float)(action - 1); /* available_torque = [-1, 0, +1] */
  torque = torque + 0.0f;             /* uniform(-0, +0): torque_noise_max = 0 */
  float y0[5] = {st->s[0], st->s[1], st->s[2], st->s[3], torque};
  float k1[5], k2[5], k3[5], k4[5], tmp[5];
  float dt2 = dt / 2.0f;
  acrobot_dsdt(y0, k1);
  for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt2 * k1[i];
  acrobot_dsdt(tmp, k2);
  for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt2 * k2[i];
  acrobot_dsdt(tmp, k3);
  for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt * k3[i];
  acrobot_dsdt(tmp, k4);
  float ns[4];
  for (int i = 0; i < 4; ++i)
    ns[i] = y0[i] + (dt / 6.0f) * (((k1[i] + 2.0f * k2[i]) + 2.0f * k3[i]) + k4[i]);
  float ja1 = acrobot_wrap(ns[0], -pi, pi);
  float ja2 = acrobot_wrap(ns[1], -pi, pi);
  float v1 = az_minf(az_maxf(ns[2], -max_vel_1), max_vel_1);
  float v2 = az_minf(az_maxf(ns[3], -max_vel_2), max_vel_2);
  int done_angle = ((-az_cosf(ja1)) - az_cosf(ja2 + ja1)) > 1.0f;
  o->reward = -1.0f * (float)(1 - done_angle);
  st->s[0] = ja1;
  st->s[1] = ja2;
  st->s[2] = v1;
  st->s[3] = v2;
  st->time = st->time + 1;
  o->terminated = ((-az_cosf(ja1)) - az_cosf(ja2 + ja1)) > 1.0f ? 1.0f : 0.0f;
  o->truncated = st->time >= 5000 ? 1.0f : 0.0f;
  o->obs[0] = az_cosf(ja1);
  o->obs[1] = az_sinf(ja1);
  o->obs[2] = az_cosf(ja2);
  o->obs[3] = az_sinf(ja2);
  o->obs[4] = v1;
  o->obs[5] = v2;
}

static void oz_env_step_one(int env_id, env_state_t* st, int action, env_out_t* o) {
  if (env_id == OZ_ENV_CARTPOLE)
    cartpole_step(st, action, o);
  else if (env_id == OZ_ENV_MOUNTAINCAR)
    mountaincar_step(st, action, o);
  else {
    acrobot_step(st, action, o);
    if (env_id == OZ_ENV_ACROBOT_NOTERM) o->terminated = 0.0f; /* acrobot_no_term.py:151-157 */
  }
}

static void oz_env_obs(int env_id, const env_state_t* st, float* obs) {
  if (env_id == OZ_ENV_CARTPOLE) {
    for (int i = 0; i < 4; ++i) obs[i] = st->s[i];
  } else if (env_id == OZ_ENV_MOUNTAINCAR) {
    obs[0] = st->s[0];
    obs[1] = st->s[1];
  } else {
    obs[0] = az_cosf(st->s[0]);
    obs[1] = az_sinf(st->s[0]);
    obs[2] = az_cosf(st->s[1]);
    obs[3] = az_sinf(st->s[1]);
    obs[4] = st->s[2];
    obs[5] = st->s[3];
  }
}

/* reset_env of the three environments (cartpole.py:93-108,
 * mountain_car.py:74-80, acrobot.py:108-122) with uniform() drawn from the
 * threefry key in the partitionable layout. */
static void oz_env_reset_one(int env_id, uint32_t k0, uint32_t k1, env_state_t* st) {
  if (env_id == OZ_ENV_MOUNTAINCAR) {
    st->s[0] = az_uniform_from_bits(az_random_bits32(k0, k1, 0u), -0.6f, -0.4f);
    st->s[1] = 0.0f;
    st->s[2] = 0.0f;
    st->s[3] = 0.0f;
  } else {
    float lim = env_id == OZ_ENV_CARTPOLE ? 0.05f : 0.1f;
    for (int i = 0; i < 4; ++i)
      st->s[i] = az_uniform_from_bits(az_random_bits32(k0, k1, (uint32_t)i), -lim, lim);
  }
  st->time = 0;
}

/* batched single env step, for the known-answer tests */
void oz_env_step(int env_id, int B, float* state, int32_t* time, const int32_t* action,
                 float* obs, float* reward, float* terminated, float* truncated) {
  int od = oz_obs_dim(env_id);
  for (int b = 0; b < B; ++b) {
    env_state_t st;
    env_out_t o;
    for (int i = 0; i < 4; ++i) st.s[i] = state[(size_t)b * 4 + i];
    st.time = time[b];
    oz_env_step_one(env_id, &st, action[b], &o);
    for (int i = 0; i < 4; ++i) state[(size_t)b * 4 + i] = st.s[i];
    time[b] = st.time;
    for (int i = 0; i < od; ++i) obs[(size_t)b * od + i] = o.obs[i];
    reward[b] = o.reward;
    terminated[b] = o.terminated;
    truncated[b] = o.truncated;
  }
}

void oz_env_reset(int env_id, int B, const uint32_t* keys, float* state, int32_t* time,
                  float* obs) {
  int od = oz_obs_dim(env_id);
  for (int b = 0; b < B; ++b) {
    env_state_t st;
    oz_env_reset_one(env_id, keys[2 * b], keys[2 * b + 1], &st);
    for (int i = 0; i < 4; ++i) state[(size_t)b * 4 + i] = st.s[i];
    time[b] = st.time;
    oz_env_obs(env_id, &st, obs + (size_t)b * od);
  }
}

/* ------------------------------------------------------------------ */
/* networks (networks/networks.py:25-48, 153-193; alphazero.py:208-216;  */
/* running_statistics.py:183-199)                                        */

/* networks/networks.py:25-48.  The summation order of a layer is a choice of this build's numerics
 * specification (az_math.h): a chain of fused multiply-adds over the inputs in index order, or,
 * for a small output head (`small_head`, az_small_heads), four interleaved chains. */
static void oz_mlp_forward_one(const oz_mlp* m, const float* x, float* out, int small_head) {
  float bufa[2048], bufb[2048];
  const float* h = x;
  float* dst = bufa;
  for (int L = 0; L < m->n_layers; ++L) {
    int in = m->sizes[L], on = m->sizes[L + 1];
    float* y = (L == m->n_layers - 1) ? out : dst;
    for (int n = 0; n < on; ++n) {
      float acc = 0.0f;
      if (small_head && L == m->n_layers - 1)
        acc = az_head_dot(h, m->w[L] + n, on, in);
      else
        for (int k = 0; k < in; ++k) acc = az_fmaf(h[k], m->w[L][(size_t)k * on + n], acc);
      acc = acc + m->b[L][n];
      if (L != m->n_layers - 1) acc = az_swish(acc);
      y[n] = acc;
    }
    h = y;
    dst = (dst == bufa) ? bufb : bufa;
  }
}

static void oz_forward_one(const oz_model* md, const float* obs, float* logits, float* value) {
  float x[8];
  int od = oz_obs_dim(md->env_id);
  for (int i = 0; i < od; ++i) x[i] = (obs[i] - md->norm_mean[i]) / md->norm_std[i];
  const oz_mlp *p = &md->policy, *v = &md->value;
  const int heads = az_small_heads(p->n_layers, p->sizes[p->n_layers - 1], p->sizes[p->n_layers],
                                   v->n_layers, v->sizes[v->n_layers - 1], v->sizes[v->n_layers]);
  oz_mlp_forward_one(p, x, logits, heads & 1);
  oz_mlp_forward_one(v, x, value, heads & 2);
}

void oz_forward(const oz_model* md, int B, const float* obs, float* logits, float* value) {
  int od = oz_obs_dim(md->env_id);
  int A = md->policy.sizes[md->policy.n_layers], K = md->value.sizes[md->value.n_layers];
  for (int b = 0; b < B; ++b)
    oz_forward_one(md, obs + (size_t)b * od, logits + (size_t)b * A, value + (size_t)b * K);
}

/* ------------------------------------------------------------------ */
/* the search (mctx_dist/_src/search.py:31-114)                         */

static void oz_search_impl(const oz_cfg* c, const oz_tree* t, const float* invalid,
                           const float* root_gumbel, const float* noise,
                           const oz_injected* inj, const oz_model* md, int64_t* sum_depth) {
  int B = c->B, S = c->S, A = c->A, K = c->K;
  int32_t* table = NULL;
  if (c->variant == OZ_VARIANT_GUMBEL) {
    table = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->max_num_considered + 1) * S);
    oz_considered_table(c->max_num_considered, S, table);
  }
  float* zero_invalid = NULL;
  if (!invalid) {
    zero_invalid = (float*)calloc((size_t)B * A, sizeof(float));
    invalid = zero_invalid;
  }
  search_t s = {c, invalid, root_gumbel, table, noise, NULL};
  /* per-simulation keys: rng, simulate_key, expand_key = split(rng, 3) */
  uint32_t* sim_keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)S);
  {
    uint32_t r0 = c->key[0], r1 = c->key[1];
    for (int sim = 0; sim < S; ++sim) {
      uint32_t n0, n1;
      az_key_split(r0, r1, 1u, &sim_keys[2 * sim], &sim_keys[2 * sim + 1]);
      az_key_split(r0, r1, 0u, &n0, &n1);
      r0 = n0;
      r1 = n1;
    }
  }
  int64_t total_depth = 0;
  for (int b = 0; b < B; ++b) {
    lane_t l = oz_lane(t, c, b);
    for (int sim = 0; sim < S; ++sim) {
      uint32_t lk0, lk1;
      az_key_split(sim_keys[2 * sim], sim_keys[2 * sim + 1], (uint32_t)b, &lk0, &lk1);
      int parent, action, nxt, depth;
      oz_simulate(&s, &l, b, sim, lk0, lk1, &parent, &action, &nxt, &depth);
      int node = (nxt == UNVISITED) ? sim + 1 : nxt; /* search.py:97-99 */
      total_depth += depth;
      float logits[OZ_MAX_A], value[OZ_MAX_K], reward, discount;
      if (inj) {
        size_t e = (size_t)b * S + sim;
        for (int a = 0; a < A; ++a) logits[a] = inj->prior_logits[e * A + a];
        for (int k = 0; k < K; ++k) value[k] = inj->value[e * K + k];
        reward = inj->reward[e];
        discount = inj->discount[e];
      } else {
        /* alphazero.py:317-339 on the native environment state */
        env_state_t st;
        env_out_t o;
        for (int i = 0; i < 4; ++i) st.s[i] = l.emb_state[(size_t)parent * 4 + i];
        st.time = l.emb_time[parent];
        oz_env_step_one(md->env_id, &st, action, &o);
        oz_forward_one(md, o.obs, logits, value);
        reward = o.reward;
        discount = md->gamma * (o.truncated != 0.0f ? 1.0f : 1.0f - o.terminated);
        for (int i = 0; i < 4; ++i) l.emb_state[(size_t)node * 4 + i] = st.s[i];
        l.emb_time[node] = st.time;
      }
      oz_expand_write(&l, parent, action, node, logits, value, reward, discount);
      oz_backward(&l, node);
    }
  }
  if (sum_depth) *sum_depth = total_depth;
  free(sim_keys);
  free(table);
  free(zero_invalid);
}

/* mctx_dist/_src/policies.py:346-483, 569-639: the search of stochastic_muzero_policy with the
 * outputs of both node functions injected per (lane, simulation).  `t` has A = AD + C columns
 * and was initialised with the root's logits padded with -inf (policies.py:426-431);
 * `is_decision` [B,N] receives the kind of every node (the root is a decision node, a child is
 * of the other kind than its parent, policies.py:620-623). */
void oz_search_stochastic(const oz_cfg* c, const oz_tree* t, const float* invalid, const float* noise,
                          const oz_injected_stochastic* inj, uint8_t* is_decision, int64_t* sum_depth) {
  int B = c->B, S = c->S, A = c->A, K = c->K, AD = c->num_decision_actions, CN = A - AD, N = S + 1;
  float* zero_invalid = NULL;
  if (!invalid) {
    zero_invalid = (float*)calloc((size_t)B * A, sizeof(float));
    invalid = zero_invalid;
  }
  search_t s = {c, invalid, NULL, NULL, noise, is_decision};
  uint32_t* sim_keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)S);
  {
    uint32_t r0 = c->key[0], r1 = c->key[1];
    for (int sim = 0; sim < S; ++sim) {
      uint32_t n0, n1;
      az_key_split(r0, r1, 1u, &sim_keys[2 * sim], &sim_keys[2 * sim + 1]);
      az_key_split(r0, r1, 0u, &n0, &n1);
      r0 = n0;
      r1 = n1;
    }
  }
  int64_t total_depth = 0;
  for (int b = 0; b < B; ++b) {
    lane_t l = oz_lane(t, c, b);
    uint8_t* kind = is_decision + (size_t)b * N;
    for (int n = 0; n < N; ++n) kind[n] = 0;
    kind[ROOT_INDEX] = 1;
    for (int sim = 0; sim < S; ++sim) {
      uint32_t lk0, lk1;
      az_key_split(sim_keys[2 * sim], sim_keys[2 * sim + 1], (uint32_t)b, &lk0, &lk1);
      int parent, action, nxt, depth;
      oz_simulate(&s, &l, b, sim, lk0, lk1, &parent, &action, &nxt, &depth);
      int node = (nxt == UNVISITED) ? sim + 1 : nxt;
      total_depth += depth;
      float logits[OZ_MAX_A], value[OZ_MAX_K], reward, discount;
      size_t e = (size_t)b * S + sim;
      if (kind[parent]) { /* policies.py:597-603 */
        for (int a = 0; a < AD; ++a) logits[a] = -INFINITY;
        for (int a = 0; a < CN; ++a) logits[AD + a] = inj->chance_logits[e * CN + a];
        for (int k = 0; k < K; ++k) value[k] = inj->afterstate_value[e * K + k];
        reward = 0.0f;
        discount = 1.0f;
      } else { /* policies.py:611-618 */
        for (int a = 0; a < AD; ++a) logits[a] = inj->action_logits[e * AD + a];
        for (int a = 0; a < CN; ++a) logits[AD + a] = -INFINITY;
        for (int k = 0; k < K; ++k) value[k] = inj->value[e * K + k];
        reward = inj->reward[e];
        discount = inj->discount[e];
      }
      uint8_t new_kind = !kind[parent];
      oz_expand_write(&l, parent, action, node, logits, value, reward, discount);
      kind[node] = new_kind;
      oz_backward(&l, node);
    }
  }
  if (sum_depth) *sum_depth = total_depth;
  free(sim_keys);
  free(zero_invalid);
}

/* search with evaluations injected per (lane, simulation); the tree must
 * have been initialised with oz_tree_init. */
void oz_search_injected(const oz_cfg* c, const oz_tree* t, const float* invalid,
                        const float* root_gumbel, const float* noise,
                        const oz_injected* inj, int64_t* sum_depth) {
  oz_search_impl(c, t, invalid, root_gumbel, noise, inj, NULL, sum_depth);
}

/* search with the built-in environment + networks as recurrent_fn. */
void oz_search_model(const oz_cfg* c, const oz_tree* t, const float* invalid,
                     const float* root_gumbel, const float* noise, const oz_model* md,
                     int64_t* sum_depth) {
  oz_search_impl(c, t, invalid, root_gumbel, noise, NULL, md, sum_depth);
}

/* ------------------------------------------------------------------ */
/* policy prologues and epilogues (mctx_dist/_src/policies.py)          */

/* policies.py:486-496 */
static void oz_mask_invalid(float* logits, const float* invalid, int A) {
  if (!invalid) return;
  float mx = logits[0];
  for (int a = 1; a < A; ++a) mx = az_maxf(mx, logits[a]);
  for (int a = 0; a < A; ++a) {
    logits[a] = logits[a] - mx;
    if (invalid[a] != 0.0f) logits[a] = -AZ_FLT_MAX;
  }
}

/* policies.py:519-524 */
static void oz_apply_temperature(float* logits, float temperature, int A) {
  float mx = logits[0];
  for (int a = 1; a < A; ++a) mx = az_maxf(mx, logits[a]);
  float den = az_maxf(AZ_FLT_TINY, temperature);
  for (int a = 0; a < A; ++a) logits[a] = (logits[a] - mx) / den;
}

/* Root priors of muzero_policy (policies.py:80-90, 499-516): Dirichlet mix,
 * log, mask.  `dirichlet` [B,A] is the injected noise sample. */
void oz_root_muzero(int B, int A, const float* prior_logits, const float* dirichlet,
                    float dirichlet_fraction, const float* invalid, float* out_logits) {
  for (int b = 0; b < B; ++b) {
    float p[OZ_MAX_A];
    oz_softmax(prior_logits + (size_t)b * A, A, p);
    float* o = out_logits + (size_t)b * A;
    for (int a = 0; a < A; ++a) {
      float noisy = (1.0f - dirichlet_fraction) * p[a] + dirichlet_fraction * dirichlet[(size_t)b * A + a];
      o[a] = az_logf(az_maxf(noisy, AZ_FLT_TINY));
    }
    oz_mask_invalid(o, invalid ? invalid + (size_t)b * A : NULL, A);
  }
}

/* Root priors of sampled_muzero_policy (policies.py:184-200, 534-546).
 * Either `counts` [B,A] is injected, or the draws are made from `key`
 * (= sample_dist_key) as argmax(logits + gumbel) with gumbel bits at flat
 * index (b*num_samples + s)*A + a.  `exponent` = 1 - 1/temperature. */
void oz_root_sampled(int B, int A, const float* prior_logits, const float* dirichlet,
                     float dirichlet_fraction, const float* invalid, int num_samples,
                     float temperature, float exponent, const float* counts_in,
                     const uint32_t* key, float* out_probs, float* out_counts) {
  float* noisy = (float*)malloc(sizeof(float) * (size_t)B * A);
  oz_root_muzero(B, A, prior_logits, dirichlet, dirichlet_fraction, invalid, noisy);
  for (int b = 0; b < B; ++b) {
    const float* L = noisy + (size_t)b * A;
    float sd[OZ_MAX_A], counts[OZ_MAX_A], prior[OZ_MAX_A];
    for (int a = 0; a < A; ++a) sd[a] = L[a];
    oz_apply_temperature(sd, temperature, A);
    if (counts_in) {
      for (int a = 0; a < A; ++a) counts[a] = counts_in[(size_t)b * A + a];
    } else {
      for (int a = 0; a < A; ++a) counts[a] = 0.0f;
      for (int s = 0; s < num_samples; ++s) {
        float z[OZ_MAX_A];
        for (int a = 0; a < A; ++a) {
          uint32_t idx = (uint32_t)(((size_t)b * num_samples + s) * A + a);
          z[a] = az_gumbel_from_bits(az_random_bits32(key[0], key[1], idx)) + sd[a];
        }
        counts[oz_argmax(z, A)] += 1.0f;
      }
    }
    oz_softmax(L, A, prior);
    for (int a = 0; a < A; ++a) {
      out_probs[(size_t)b * A + a] = (counts[a] / (float)num_samples) * az_powf(prior[a], exponent);
      if (out_counts) out_counts[(size_t)b * A + a] = counts[a];
    }
  }
  free(noisy);
}

/* Root logits of gumbel_muzero_policy (policies.py:291-297): mask; the
 * Gumbel sample is injected or drawn from `key` (= gumbel_rng) at flat index
 * b*A + a and scaled. */
void oz_root_gumbel(int B, int A, const float* prior_logits, const float* invalid,
                    const float* gumbel_in, const uint32_t* key, float gumbel_scale,
                    float* out_logits, float* out_gumbel) {
  for (int b = 0; b < B; ++b) {
    float* o = out_logits + (size_t)b * A;
    for (int a = 0; a < A; ++a) o[a] = prior_logits[(size_t)b * A + a];
    oz_mask_invalid(o, invalid ? invalid + (size_t)b * A : NULL, A);
    for (int a = 0; a < A; ++a) {
      size_t i = (size_t)b * A + a;
      float g = gumbel_in ? gumbel_in[i]
                          : az_gumbel_from_bits(az_random_bits32(key[0], key[1], (uint32_t)i));
      out_gumbel[i] = gumbel_scale * g;
    }
  }
}

/* Epilogue of muzero_policy / sampled_muzero_policy (policies.py:113-122,
 * 225-234; tree.py:102-120): visit probabilities and the categorical draw
 * argmax(gumbel + logits/T).  `gumbel` [B,A] injected or drawn from `key`. */
void oz_epilogue_visits(const oz_cfg* c, const oz_tree* t, float temperature,
                        const float* gumbel_in, const uint32_t* key, int32_t* action,
                        float* action_weights) {
  int B = c->B, A = c->A;
  for (int b = 0; b < B; ++b) {
    lane_t l = oz_lane(t, c, b);
    float counts[OZ_MAX_A], lg[OZ_MAX_A], z[OZ_MAX_A];
    float total = 0.0f;
    for (int a = 0; a < A; ++a) {
      counts[a] = (float)l.cvisits[a];
      total = total + counts[a];
    }
    for (int a = 0; a < A; ++a) {
      float p = counts[a] / az_maxf(total, 1.0f);
      if (!(total > 0.0f)) p = (float)(1.0 / (double)A);
      action_weights[(size_t)b * A + a] = p;
      lg[a] = az_logf(az_maxf(p, AZ_FLT_TINY));
    }
    oz_apply_temperature(lg, temperature, A);
    for (int a = 0; a < A; ++a) {
      size_t i = (size_t)b * A + a;
      float g = gumbel_in ? gumbel_in[i]
                          : az_gumbel_from_bits(az_random_bits32(key[0], key[1], (uint32_t)i));
      z[a] = g + lg[a];
    }
    action[b] = oz_argmax(z, A);
  }
}

/* Epilogue of gumbel_muzero_policy (policies.py:321-343).  `root_logits` are
 * the masked root logits, `gumbel` the scaled root Gumbel sample. */
void oz_epilogue_gumbel(const oz_cfg* c, const oz_tree* t, const float* root_logits,
                        const float* gumbel, const float* invalid, int32_t* action,
                        float* action_weights) {
  int B = c->B, A = c->A;
  for (int b = 0; b < B; ++b) {
    lane_t l = oz_lane(t, c, b);
    const int32_t* vis = l.cvisits;
    int32_t considered_visit = vis[0];
    for (int a = 1; a < A; ++a)
      if (vis[a] > considered_visit) considered_visit = vis[a];
    float cq[OZ_MAX_A], score[OZ_MAX_A], x[OZ_MAX_A];
    oz_qtransform(&l, c, ROOT_INDEX, cq);
    oz_score_considered(considered_visit, gumbel + (size_t)b * A, root_logits + (size_t)b * A, cq,
                        vis, A, score);
    action[b] = oz_masked_argmax(score, invalid ? invalid + (size_t)b * A : NULL, A);
    for (int a = 0; a < A; ++a) x[a] = root_logits[(size_t)b * A + a] + cq[a];
    oz_mask_invalid(x, invalid ? invalid + (size_t)b * A : NULL, A);
    oz_softmax(x, A, action_weights + (size_t)b * A);
  }
}

/* tree.py:102-120: SearchSummary of the root. */
void oz_summary(const oz_cfg* c, const oz_tree* t, float* visit_counts, float* visit_probs,
                float* value, float* qvalues) {
  int B = c->B, A = c->A, K = c->K;
  for (int b = 0; b < B; ++b) {
    lane_t l = oz_lane(t, c, b);
    float total = 0.0f;
    for (int a = 0; a < A; ++a) {
      visit_counts[(size_t)b * A + a] = (float)l.cvisits[a];
      total = total + (float)l.cvisits[a];
    }
    for (int a = 0; a < A; ++a) {
      float p = (float)l.cvisits[a] / az_maxf(total, 1.0f);
      if (!(total > 0.0f)) p = (float)(1.0 / (double)A);
      visit_probs[(size_t)b * A + a] = p;
      for (int k = 0; k < K; ++k)
        qvalues[((size_t)b * A + a) * K + k] = l.crew[a] + l.cdisc[a] * l.cval[(size_t)a * K + k];
    }
    for (int k = 0; k < K; ++k) value[(size_t)b * K + k] = l.nval[k];
  }
}

/* alphazero.py:367-379: greedy action over searched actions, search value,
 * log-probability of the chosen action under the prior. */
void oz_actor_post(const oz_cfg* c, const oz_tree* t, const float* action_weights,
                   const int32_t* sampled_action, int deterministic, const float* prior_logits,
                   int32_t* action, float* search_value, float* prior_log_prob) {
  int B = c->B, A = c->A, K = c->K;
  for (int b = 0; b < B; ++b) {
    lane_t l = oz_lane(t, c, b);
    float q[OZ_MAX_A], masked[OZ_MAX_A];
    oz_mean_qvalues(&l, ROOT_INDEX, q);
    for (int a = 0; a < A; ++a)
      masked[a] = action_weights[(size_t)b * A + a] != 0.0f ? q[a] : -AZ_INF;
    int best = oz_argmax(masked, A);
    int act = deterministic ? best : sampled_action[b];
    action[b] = act;
    for (int k = 0; k < K; ++k) search_value[(size_t)b * K + k] = l.nval[k];
    const float* pl = prior_logits + (size_t)b * A;
    float mx = pl[0];
    for (int a = 1; a < A; ++a) mx = az_maxf(mx, pl[a]);
    float se = 0.0f;
    for (int a = 0; a < A; ++a) se = se + az_expf(pl[a] - mx);
    prior_log_prob[b] = (pl[act] - mx) - az_logf(se);
  }
}

/* ------------------------------------------------------------------ */
/* the real environment step: AutoResetWrapper(VmapWrapper(EpisodeWrapper(
 * GymnaxToBraxWrapper(env)))).step  (envs/brax_wrappers.py:81-97, 109-134;
 * gymnax/gymnax/wrappers/brax.py:57-70; environment.py:26-48). */
void oz_real_env_step(int env_id, int B, int episode_length, float* state, int32_t* time,
                      float* obs, float* real_obs, float* reward, float* done,
                      float* truncation, float* steps, uint32_t* rng, const int32_t* action) {
  int od = oz_obs_dim(env_id);
  for (int b = 0; b < B; ++b) {
    /* AutoResetWrapper.step: zero `steps` where the previous step was done */
    float st_steps = done[b] != 0.0f ? 0.0f : steps[b];
    /* GymnaxToBraxWrapper.step: rng, step_rng = split(info._rng) */
    uint32_t r0, r1;
    az_key_split(rng[2 * b], rng[2 * b + 1], 0u, &r0, &r1);
    rng[2 * b] = r0;
    rng[2 * b + 1] = r1;
    env_state_t st;
    env_out_t o;
    for (int i = 0; i < 4; ++i) st.s[i] = state[(size_t)b * 4 + i];
    st.time = time[b];
    oz_env_step_one(env_id, &st, action[b], &o);
    /* EpisodeWrapper.step */
    st_steps = st_steps + 1.0f;
    float env_done = o.terminated;
    float d = st_steps >= (float)episode_length ? 1.0f : env_done;
    truncation[b] = st_steps >= (float)episode_length ? 1.0f - env_done : 0.0f;
    steps[b] = st_steps;
    reward[b] = o.reward;
    done[b] = d;
    for (int i = 0; i < od; ++i) real_obs[(size_t)b * od + i] = o.obs[i];
    /* AutoResetWrapper: reset from the stored key, select where done */
    if (d != 0.0f) {
      env_state_t rs;
      oz_env_reset_one(env_id, r0, r1, &rs);
      st = rs;
      oz_env_obs(env_id, &rs, o.obs);
    }
    for (int i = 0; i < 4; ++i) state[(size_t)b * 4 + i] = st.s[i];
    time[b] = st.time;
    for (int i = 0; i < od; ++i) obs[(size_t)b * od + i] = o.obs[i];
  }
}

/* ------------------------------------------------------------------ */
/* vector entry points of the numerics specification, for host/device   */
/* bit-comparison tests                                                 */

void oz_math_apply(int fn, int n, const float* x, float* y) {
  for (int i = 0; i < n; ++i) {
    float v = x[i];
    switch (fn) {
      case 0: y[i] = az_expf(v); break;
      case 1: y[i] = az_logf(v); break;
      case 2: y[i] = az_sinf(v); break;
      case 3: y[i] = az_cosf(v); break;
      case 4: y[i] = az_swish(v); break;
      case 5: y[i] = az_sqrtf(v); break;
      case 6: y[i] = 1.0f / v; break;
      case 7: y[i] = az_gumbel_from_bits(az_f2u(v)); break;
      case 9: y[i] = az_trig_(v, 0); break;  /* select-only forms of sin / cos */
      case 10: y[i] = az_trig_(v, 1); break;
      default: y[i] = v;
    }
  }
}

/* x[i]^e with the shared pow (the replay sampler's importance weights). */
void oz_powf(int n, const float* x, float e, float* y) {
  for (int i = 0; i < n; ++i) y[i] = az_powf(x[i], e);
}

/* jax.random.bits(key, (n,), uint32) and jax.random.uniform(key, (n,)) in [0, 1). */
void oz_random_bits(const uint32_t* key, int n, uint32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = az_random_bits32(key[0], key[1], (uint32_t)i);
}
void oz_uniform01(const uint32_t* key, int n, float* out) {
  for (int i = 0; i < n; ++i)
    out[i] = az_uniform_from_bits(az_random_bits32(key[0], key[1], (uint32_t)i), 0.0f, 1.0f);
}

void oz_threefry(int n, const uint32_t* key, const uint32_t* ctr, uint32_t* out) {
  for (int i = 0; i < n; ++i)
    az_threefry2x32(key[2 * i], key[2 * i + 1], ctr[2 * i], ctr[2 * i + 1], &out[2 * i],
                    &out[2 * i + 1]);
}

void oz_key_split(const uint32_t* key, int n, uint32_t* out) {
  for (int i = 0; i < n; ++i) az_key_split(key[0], key[1], (uint32_t)i, &out[2 * i], &out[2 * i + 1]);
}
