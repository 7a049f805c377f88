// az_coop.h -- cooperative (32 threads per lane) form of the tree search.
//
// The literal search walks root -> leaf re-evaluating the selection rule at every
// level (d dependent round trips of loads + exp/div arithmetic per simulation) and
// then walks leaf -> root again for the backup.  The selection at a node depends only
// on that node's own statistics, and those change only when the node lies on a
// backup path.  So this form keeps, per node, the action the rule currently selects
// together with the child it leads to (`sel`, in LDS), and restructures a simulation
// as
//   1. chase   : one thread per lane follows `sel` from the root -- one dependent LDS
//                read per level -- and records the path;
//   2. evaluate: the caller's recurrent_fn (injected arrays, or env + MLP);
//   3. update  : the 32 threads of the lane take one path node each: load the node's
//                row (one round trip, all nodes in parallel; the loads only need the
//                path, so they can be issued before the evaluation), recompute the
//                value chain from the leaf down to their node, apply the backup to
//                their node, and re-evaluate the selection rule on the updated row.
// Every floating-point operation is the same, on the same operands and in the same
// order, as in the literal form (az_device.h), so the resulting tree is bit-identical.
//
// Per-visit randomness: the PUCT rules add 1e-7 * uniform noise to break ties.  A cached
// pick is marked ambiguous when that noise could change it (the top two scores closer
// than the noise), and the chase then evaluates that node exactly, with the visit's own
// noise, as the literal form does; everywhere else the noise cannot matter.
//
// The kernels are instantiated twice (NOISY): the form without per-visit noise (Gumbel, or
// PUCT with the noise disabled) carries none of that code.
//
// Applicability: A <= 8, K <= 8, N <= 65535.
#ifndef AZ_COOP_H_
#define AZ_COOP_H_

#include "az_device.h"

#define AZ_GROUP 32        // threads per lane
#define AZ_WG_LANES 16     // lanes per workgroup
#define AZ_TREE_THREADS (AZ_GROUP * AZ_WG_LANES)

// Words of LDS per lane: sel[N], edge_r[N], edge_d[N], path[S+1], pr[S], pd[S], hdr[8],
// root_g[8], root_inv[8], root_tab[S].  With `borrowed`, pr and pd live in memory the
// caller lends (2*S floats per lane, free between evaluations) and are not counted.
__host__ __device__ inline int az_coop_lane_words(int S, bool borrowed = false) {
  return (borrowed ? 5 : 7) * S + 28;
}

struct CoopLds {
  int32_t* sel;   // [N]   (action << 16) | (child + 1) currently selected at each node
  float* edge_r;  // [N]   reward of the edge leading INTO each node
  float* edge_d;  // [N]   discount of that edge
  int32_t* path;  // [S+1] node | action << 16 per path position
  float* pr;      // [S]   edge reward by path position
  float* pd;      // [S]   edge discount by path position
  int32_t* hdr;   // [8]   depth, new node, parent, action, reward bits, discount bits, -, table row
  float* root_g;      // [8] the root's Gumbel noise
  float* root_inv;    // [8] the root's invalid-action mask
  int32_t* root_tab;  // [S] the row of the considered-visits table this lane uses
  AZ_DEV CoopLds(float* base, int lane, int S, float* borrowed = nullptr) {
    float* p = base + (size_t)lane * az_coop_lane_words(S, borrowed != nullptr);
    const int N = S + 1;
    sel = reinterpret_cast<int32_t*>(p);
    edge_r = p + N;
    edge_d = p + 2 * N;
    path = reinterpret_cast<int32_t*>(p + 3 * N);
    float* q = p + 4 * N;
    if (borrowed) {
      pr = borrowed + (size_t)lane * 2 * S;
    } else {
      pr = q;
      q += 2 * S;
    }
    pd = pr + S;
    hdr = reinterpret_cast<int32_t*>(q);
    root_g = q + 8;
    root_inv = root_g + 8;
    root_tab = reinterpret_cast<int32_t*>(root_inv + 8);
  }
  // Copies of the root-only selection inputs (filled by az_coop_init): the root is
  // re-selected every simulation, and reading these from global memory again would
  // put two dependent round trips on the critical path of every update.
  AZ_DEV RootCache root_cache(const SearchCtx& c) const {
    RootCache rc;
    rc.gumbel = root_g;
    rc.invalid = root_inv;
    rc.has_invalid = c.in.invalid_actions != nullptr;
    rc.table_row = root_tab;
    rc.table_nc = hdr[7];
    return rc;
  }
};

// The statistics arrays of a workgroup's lanes staged in LDS for the whole search (the
// tree-only kernel, when they fit): the search then touches HBM twice per array -- one
// coalesced read of the fresh tree and one coalesced write of the finished one (the 16 lanes
// of a workgroup are contiguous in every [B, N, ...] array) -- instead of a cache line per
// node visit.  `view()` is an az_tree over the copies, indexed by the lane's position in the workgroup.
struct TreeLds {
  int32_t *nv, *par, *afp, *cidx, *cvis;
  float *raw, *nval, *clog, *crew, *cdisc, *cval;
  int N, A, K;
  __host__ __device__ static inline size_t words(int N, int A, int K) {
    return (size_t)AZ_WG_LANES * N * (3 + 2 * K + A * (5 + K));
  }
  AZ_DEV TreeLds(float* base, int N_, int A_, int K_) : N(N_), A(A_), K(K_) {
    const int L = AZ_WG_LANES * N;
    int32_t* ip = reinterpret_cast<int32_t*>(base);
    nv = ip;
    par = ip + L;
    afp = ip + 2 * L;
    cidx = ip + 3 * L;
    cvis = cidx + L * A;
    float* fp = reinterpret_cast<float*>(cvis + L * A);
    raw = fp;
    nval = raw + L * K;
    clog = nval + L * K;
    crew = clog + L * A;
    cdisc = crew + L * A;
    cval = cdisc + L * A;
  }
  // An az_tree over the LDS copies: lane `ln` of the workgroup is its batch row `ln`.
  // (No negative base offsets here: a pointer into LDS must stay inside the LDS window.)
  AZ_DEV az_tree view() const {
    az_tree v{};
    v.node_visits = nv;
    v.parents = par;
    v.action_from_parent = afp;
    v.children_index = cidx;
    v.children_visits = cvis;
    v.raw_values = raw;
    v.node_values = nval;
    v.children_prior_logits = clog;
    v.children_rewards = crew;
    v.children_discounts = cdisc;
    v.children_values = cval;
    v.emb_state = nullptr;
    v.emb_time = nullptr;
    return v;
  }
  // copies the lanes [b0, b0 + lanes) between the global tree and LDS (workgroup-wide)
  AZ_DEV void copy(const az_tree& t, int b0, int lanes, bool to_lds, int tid, int nthreads) const {
    const size_t o = (size_t)b0 * N;
    const int n1 = lanes * N, nA = n1 * A, nK = n1 * K, nAK = nA * K;
#define AZ_TL_COPY(lds_arr, glob_arr, off, count)                          \
  for (int i = tid; i < (count); i += nthreads) {                          \
    if (to_lds)                                                            \
      lds_arr[i] = (glob_arr)[(off) + i];                                  \
    else                                                                   \
      (glob_arr)[(off) + i] = lds_arr[i];                                  \
  }
    AZ_TL_COPY(nv, t.node_visits, o, n1)
    AZ_TL_COPY(par, t.parents, o, n1)
    AZ_TL_COPY(afp, t.action_from_parent, o, n1)
    AZ_TL_COPY(cidx, t.children_index, o * A, nA)
    AZ_TL_COPY(cvis, t.children_visits, o * A, nA)
    AZ_TL_COPY(raw, t.raw_values, o * K, nK)
    AZ_TL_COPY(nval, t.node_values, o * K, nK)
    AZ_TL_COPY(clog, t.children_prior_logits, o * A, nA)
    AZ_TL_COPY(crew, t.children_rewards, o * A, nA)
    AZ_TL_COPY(cdisc, t.children_discounts, o * A, nA)
    AZ_TL_COPY(cval, t.children_values, o * A * K, nAK)
#undef AZ_TL_COPY
  }
};

// Every per-node word of one node, in registers.
template <int AMAX, int KMAX>
struct FullRow {
  int32_t nv;
  int32_t cidx[AMAX], vis[AMAX];
  float logits[AMAX], rew[AMAX], disc[AMAX];
  float cval[AMAX][KMAX];
  float raw[KMAX], nval[KMAX];
};

template <int AMAX, int KMAX>
AZ_DEV void az_load_full_row(const Lane<AMAX, KMAX>& l, int node, FullRow<AMAX, KMAX>& r) {
  const int A = l.A, K = l.K;
  r.nv = l.nv[node];
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      r.cidx[a] = l.cidx[node * A + a];
      r.vis[a] = l.cvis[node * A + a];
      r.logits[a] = l.clog[node * A + a];
      r.rew[a] = l.crew[node * A + a];
      r.disc[a] = l.cdisc[node * A + a];
AZ_UNROLL_K
      for (int k = 0; k < AZ_TRIP_K; ++k)
        if (k < K) r.cval[a][k] = l.cval[((size_t)node * A + a) * K + k];
    }
AZ_UNROLL_K
  for (int k = 0; k < AZ_TRIP_K; ++k)
    if (k < K) {
      r.raw[k] = l.raw[(size_t)node * K + k];
      r.nval[k] = l.nval[(size_t)node * K + k];
    }
}

// The selection inputs of a row held in registers; same arithmetic as az_load_row.
template <int AMAX, int KMAX>
AZ_DEV void az_node_row(const FullRow<AMAX, KMAX>& f, int A, int K, NodeRow<AMAX>& r) {
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      r.vis[a] = f.vis[a];
      r.logits[a] = f.logits[a];
      float s = 0.0f;
AZ_UNROLL_K
      for (int k = 0; k < AZ_TRIP_K; ++k)
        if (k < K) s = s + (f.rew[a] + f.disc[a] * f.cval[a][k]);
      r.q[a] = s / (float)K;
    }
  float sv = 0.0f, sr = 0.0f;
AZ_UNROLL_K
  for (int k = 0; k < AZ_TRIP_K; ++k)
    if (k < K) {
      sv = sv + f.nval[k];
      sr = sr + f.raw[k];
    }
  r.vmean = sv / (float)K;
  r.rawmean = sr / (float)K;
  r.node_visit = f.nv;
}

// The action the search's selection rule picks at a node (root: depth 0), evaluated
// WITHOUT per-visit noise.  `ambiguous` tells whether the PUCT tie-break noise
// (1e-7 * u, u in [0, 1)) could change the pick: it cannot when the best score leads every
// other one by more than the noise plus the rounding of the additions; otherwise the chase
// re-evaluates the node with the visit's actual noise (az_coop_exact_select).
template <int AMAX, bool NOISY>
AZ_DEV int az_select_cached(const SearchCtx& c, const CoopLds& s, int A, int b, bool root,
                            const NodeRow<AMAX>& r, bool& ambiguous) {
  const RootCache rc = s.root_cache(c);
  ambiguous = false;
  if (c.d.variant == AZ_VARIANT_GUMBEL)
    return root ? az_select_gumbel_root<AMAX>(c, A, r, rc) : az_select_gumbel_interior<AMAX>(c, A, r);
  float score[AMAX], noise[AMAX];
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a) noise[a] = 1e-7f * 0.0f;
  az_puct_scores<AMAX>(c, A, r, score);
  const int best = az_puct_argmax<AMAX>(A, root ? 0 : 1, score, noise, rc);
  if (NOISY && c.d.noise_mode != AZ_NOISE_ZERO) {
    const bool use_mask = root && rc.has_invalid;
    float bs = 0.0f;
    bool best_masked = false;
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a == best) {
        bs = score[a];
        best_masked = use_mask && rc.invalid[a] != 0.0f;
      }
    if (best_masked || !(bs == bs)) ambiguous = true;  // everything masked, or NaN: let the exact rule decide
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A && a != best && !(use_mask && rc.invalid[a] != 0.0f)) {
        const float x = score[a];
        float m = bs < 0.0f ? -bs : bs;
        const float ax = x < 0.0f ? -x : x;
        m = ax > m ? ax : m;
        m = m > 1.0f ? m : 1.0f;
        if (!(bs - x > 4e-7f * m)) ambiguous = true;
      }
  }
  return best;
}

#define AZ_SEL_AMBIG (1 << 30)
AZ_DEV int32_t az_pack_sel(int action, int child, bool ambiguous = false) {
  return (action << 16) | (child + 1) | (ambiguous ? AZ_SEL_AMBIG : 0);
}

// The selection at `node` with the tie-break noise of visit (sim, depth), as the literal
// search evaluates it (az_simulate_lane).  (k0, k1) is the lane's simulate key advanced to
// `key_depth` levels; it is advanced further as needed.
template <int AMAX, int KMAX>
AZ_DEV int az_coop_exact_select(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const CoopLds& s, int b,
                                int sim, int node, int depth, uint32_t& k0, uint32_t& k1,
                                int& key_depth, int& child) {
  const int A = l.A;
  NodeRow<AMAX> r;
  az_load_row<AMAX, KMAX>(l, node, r);
  if (c.d.noise_mode == AZ_NOISE_THREEFRY)
    for (; key_depth < depth; ++key_depth) {
      uint32_t n0, n1;
      az_key_split(k0, k1, 0u, &n0, &n1);
      k0 = n0;
      k1 = n1;
    }
  float noise[AMAX];
  az_puct_noise<AMAX>(c, A, b, sim, depth, k0, k1, noise);
  key_depth = depth + 1;
  const int act = az_select_puct<AMAX>(c, A, depth, r, noise, s.root_cache(c));
  child = l.cidx[node * A + act];
  return act;
}

// Before the first simulation: selection at the root of the fresh tree.
template <int AMAX, int KMAX, bool NOISY>
AZ_DEV void az_coop_init(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const CoopLds& s, int b) {
  const int A = l.A;
  for (int a = 0; a < A; ++a) {
    s.root_g[a] = c.in.root_gumbel ? c.in.root_gumbel[(size_t)b * A + a] : 0.0f;
    s.root_inv[a] = c.in.invalid_actions ? c.in.invalid_actions[(size_t)b * A + a] : 0.0f;
  }
  s.hdr[7] = -1;
  if (c.d.variant == AZ_VARIANT_GUMBEL && c.in.table) {
    float nvalid = 0.0f;
    for (int a = 0; a < A; ++a) nvalid = nvalid + (1.0f - s.root_inv[a]);
    const int32_t nc = az_num_considered(c.d, nvalid);
    for (int i = 0; i < c.d.S; ++i) s.root_tab[i] = c.in.table[(size_t)nc * c.d.S + i];
    s.hdr[7] = nc;
  }
  FullRow<AMAX, KMAX> f;
  NodeRow<AMAX> r;
  az_load_full_row<AMAX, KMAX>(l, AZ_ROOT, f);
  az_node_row<AMAX, KMAX>(f, A, l.K, r);
  bool ambiguous;
  const int a = az_select_cached<AMAX, NOISY>(c, s, A, b, true, r, ambiguous);
  int child = AZ_UNVISITED;
AZ_UNROLL_A
  for (int i = 0; i < AZ_TRIP_A; ++i)
    if (i == a) child = f.cidx[i];
  s.sel[AZ_ROOT] = az_pack_sel(a, child, ambiguous);
}

// Step 1 (one thread per lane): follow the cached selections from the root
// (search.py:127-187) and record the path.  Only the selection word is on the
// dependent chain; the edge rewards / discounts of the path are gathered afterwards
// by all threads (az_coop_fill_edges).  Returns the depth of the expanded node.
// `sim_keys` ([S][2] words, or null): the per-simulation simulate keys (search.py:90), for
// selections whose cached pick the tie-break noise could change.
template <int AMAX, int KMAX, bool NOISY>
AZ_DEV int az_coop_chase(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const CoopLds& s, int b, int sim,
                         int max_depth, const uint32_t* sim_keys) {
  int node = AZ_ROOT, dep = 0, act, nxt;
  uint32_t k0 = 0u, k1 = 0u;
  int key_depth = -1;  // -1: the lane's key of this simulation has not been derived yet
  for (;;) {
    const int32_t e = s.sel[node];
    act = (e >> 16) & 0xff;
    nxt = (e & 0xffff) - 1;
    if (NOISY && (e & AZ_SEL_AMBIG)) {
      if (key_depth < 0) {
        if (sim_keys) az_key_split(sim_keys[2 * sim], sim_keys[2 * sim + 1], (uint32_t)b, &k0, &k1);
        key_depth = 0;
      }
      act = az_coop_exact_select<AMAX, KMAX>(c, l, s, b, sim, node, dep, k0, k1, key_depth, nxt);
    }
    s.path[dep] = node | (act << 16);
    dep += 1;
    if (nxt < 0 || dep >= max_depth) break;
    node = nxt;
  }
  s.hdr[0] = dep;
  s.hdr[1] = nxt < 0 ? sim + 1 : nxt;  // search.py:97-99
  s.hdr[2] = node;
  s.hdr[3] = act;
  return dep;
}

// Workgroup barrier that publishes LDS writes only: global loads that were requested before it
// stay in flight (a plain __syncthreads() waits for them).
AZ_DEV void az_coop_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The same walk for the rules without per-visit noise, written so that every lane of the wave
// runs every iteration under a predicate and the loop ends on one scalar test (no lane ever
// leaves the loop early): the walk is one dependent LDS read per level, and in the plain form
// above the execution-mask bookkeeping of the divergent loop costs more than that read.
AZ_DEV int az_coop_chase_lean(const CoopLds& s, int sim, int max_depth, bool active) {
  // One wave alone on its SIMD issues an instruction every four cycles or so, so the cost of a
  // level is its instruction count as much as the LDS round trip: the loop carries only the
  // node and the depth; the last edge is read back from the recorded path afterwards.
  int node = active ? AZ_ROOT : -1;  // -1: this lane's walk has ended
  int dep = 0;
  s.hdr[6] = 0;  // what a finished lane reads instead of a selection word: "no child"
  // four levels per exit test: the test (vector compare -> scalar branch) is as long as the LDS
  // round trip it guards, and a finished lane's extra levels are harmless
  while (__builtin_amdgcn_ballot_w64(node >= 0) != 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = node >= 0;
      const int32_t e = *(live ? s.sel + node : s.hdr + 6);
      const int a = (e >> 16) & 0xff, n = (e & 0xffff) - 1;  // n == -1: unvisited edge, or finished
      s.path[dep] = node | (a << 16);  // a finished lane scribbles on the entry past its path's end
      dep += live ? 1 : 0;
      node = dep < max_depth ? n : -1;
    }
  }
  if (active) {
    const int32_t pe = s.path[dep - 1];
    const int last = pe & 0xffff, act = (pe >> 16) & 0xff;
    const int nxt = (s.sel[last] & 0xffff) - 1;
    s.hdr[0] = dep;
    s.hdr[1] = nxt < 0 ? sim + 1 : nxt;  // search.py:97-99
    s.hdr[2] = last;
    s.hdr[3] = act;
  }
  return dep;
}

// The simulate keys of all S simulations into `sim_keys` ([S][2] words of LDS): the rng
// chain rng, simulate_key, expand_key = split(rng, 3) (search.py:90) is sequential, so one
// thread walks it and the others derive their simulation's key from it.  Workgroup-wide
// call (contains barriers).
AZ_DEV void az_coop_sim_keys(const az_search_desc& d, uint32_t* sim_keys, int tid, int nthreads) {
  if (tid == 0) {
    uint32_t r0 = d.key[0], r1 = d.key[1];
    for (int i = 0; i < d.S; ++i) {
      sim_keys[2 * i] = r0;
      sim_keys[2 * i + 1] = r1;
      uint32_t n0, n1;
      az_key_split(r0, r1, 0u, &n0, &n1);
      r0 = n0;
      r1 = n1;
    }
  }
  __syncthreads();
  for (int i = tid; i < d.S; i += nthreads) {
    uint32_t s0, s1;
    az_key_split(sim_keys[2 * i], sim_keys[2 * i + 1], 1u, &s0, &s1);
    sim_keys[2 * i] = s0;
    sim_keys[2 * i + 1] = s1;
  }
  __syncthreads();
}

// Thread j of the lane's group copies the reward / discount of the path edges
// j, j + AZ_GROUP, ... between recorded nodes (the edge into the evaluated node comes
// from the evaluation itself).  Call after the chase is visible to the group and
// before az_coop_apply, with a barrier on either side.
AZ_DEV void az_coop_fill_edges(const CoopLds& s, int j, int dep) {
  for (int q = j; q < dep - 1; q += AZ_GROUP) {
    const int child = s.path[q + 1] & 0xffff;
    s.pr[q] = s.edge_r[child];
    s.pd[q] = s.edge_d[child];
  }
}

// Largest path depth over the workgroup's lanes (uniform across the workgroup).
AZ_DEV int az_coop_max_depth(float* lds_base, int S, bool borrowed = false) {
  int m = 0;
#pragma unroll
  for (int i = 0; i < AZ_WG_LANES; ++i) {
    const int d = reinterpret_cast<const int32_t*>(
        lds_base + (size_t)i * az_coop_lane_words(S, borrowed) + 4 * (S + 1) + (borrowed ? 0 : 2 * S))[0];
    m = d > m ? d : m;
  }
  return m;
}

// Step 3 for path position p (0 = root ... dep-1 = parent, dep = the expanded node):
// phase A loads what the position needs, phase B (after a workgroup barrier) applies
// expand (search.py:231-244, 304-342) and backward (search.py:265-286) to the node
// at that position and refreshes its cached selection.
template <int AMAX, int KMAX>
struct CoopPos {
  FullRow<AMAX, KMAX> row;
  float child_old[KMAX];
  int32_t child_cnt;
  int node, act;
};

template <int AMAX, int KMAX>
AZ_DEV void az_coop_load(const Lane<AMAX, KMAX>& l, const CoopLds& s, int p, int dep, int newn,
                         CoopPos<AMAX, KMAX>& P) {
  const int K = l.K;
  if (p < dep) {
    const int32_t e = s.path[p];
    P.node = e & 0xffff;
    P.act = e >> 16;
  } else {
    P.node = newn;
    P.act = 0;
  }
  az_load_full_row<AMAX, KMAX>(l, P.node, P.row);
  P.child_cnt = 0;
  if (p + 1 < dep) {
    const int child = s.path[p + 1] & 0xffff;
    P.child_cnt = l.nv[child];
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k)
      if (k < K) P.child_old[k] = l.nval[(size_t)child * K + k];
  }
}

template <int AMAX, int KMAX, bool NOISY>
AZ_DEV void az_coop_apply(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const CoopLds& s, int b,
                          int p, int dep, int newn, int parent, int action,
                          const float* ev_logits, const float* ev_value, float reward,
                          float discount, CoopPos<AMAX, KMAX>& P) {
  const int A = l.A, K = l.K;
  FullRow<AMAX, KMAX>& f = P.row;
  const int node = P.node;
  if (p == dep) {
    // the evaluated node (update_tree_node, search.py:304-342)
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) {
        f.logits[a] = ev_logits[a];
        l.clog[node * A + a] = f.logits[a];
      }
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k)
      if (k < K) {
        f.raw[k] = ev_value[k];
        f.nval[k] = ev_value[k];
        l.raw[(size_t)node * K + k] = f.raw[k];
        l.nval[(size_t)node * K + k] = f.nval[k];
      }
    f.nv = f.nv + 1;
    l.nv[node] = f.nv;
    l.par[node] = parent;
    l.afp[node] = action;
    s.edge_r[node] = reward;
    s.edge_d[node] = discount;
  } else {
    const int act = P.act;
    if (p == dep - 1) {
      // the edge parent -> evaluated node (search.py:235-241)
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a == act) {
          f.cidx[a] = newn;
          f.rew[a] = reward;
          f.disc[a] = discount;
        }
      l.cidx[node * A + act] = newn;
      l.crew[node * A + act] = reward;
      l.cdisc[node * A + act] = discount;
    }
    // backward (search.py:265-286): value chain from the leaf down to this node
    float lv[KMAX], child_new[KMAX];
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k)
      if (k < K) lv[k] = ev_value[k];
    // the edge into the evaluated node first, then the recorded edges q = dep-2 ... p+1,
    // read from LDS four at a time so that a round trip covers four dependent steps
    if (dep - 1 > p) {
AZ_UNROLL_K
      for (int k = 0; k < AZ_TRIP_K; ++k)
        if (k < K) lv[k] = reward + discount * lv[k];
    }
    for (int q = dep - 2; q > p; q -= 4) {
      float rr[4], dd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qi = q - i > 0 ? q - i : 0;
        rr[i] = s.pr[qi];
        dd[i] = s.pd[qi];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (q - i > p) {
AZ_UNROLL_K
          for (int k = 0; k < AZ_TRIP_K; ++k)
            if (k < K) lv[k] = rr[i] + dd[i] * lv[k];
        }
    }
    const float cnt_child = (float)P.child_cnt;
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k)
      if (k < K)
        child_new[k] = p + 1 == dep ? ev_value[k]
                                    : (P.child_old[k] * cnt_child + lv[k]) / (cnt_child + 1.0f);
    const float r = p == dep - 1 ? reward : s.pr[p];
    const float dsc = p == dep - 1 ? discount : s.pd[p];
    const float cnt = (float)f.nv;
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k)
      if (k < K) {
        lv[k] = r + dsc * lv[k];
        f.nval[k] = (f.nval[k] * cnt + lv[k]) / (cnt + 1.0f);
        l.nval[(size_t)node * K + k] = f.nval[k];
        l.cval[((size_t)node * A + act) * K + k] = child_new[k];
      }
    int32_t newvis = 0;
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a == act) {
AZ_UNROLL_K
        for (int k = 0; k < AZ_TRIP_K; ++k)
          if (k < K) f.cval[a][k] = child_new[k];
        f.vis[a] = f.vis[a] + 1;
        newvis = f.vis[a];
      }
    l.cvis[node * A + act] = newvis;
    f.nv = f.nv + 1;
    l.nv[node] = f.nv;
  }
  // refresh the cached selection of this node
  NodeRow<AMAX> r;
  az_node_row<AMAX, KMAX>(f, A, K, r);
  bool ambiguous;
  const int best = az_select_cached<AMAX, NOISY>(c, s, A, b, p == 0, r, ambiguous);
  int child = AZ_UNVISITED;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a == best) child = f.cidx[a];
  s.sel[node] = az_pack_sel(best, child, ambiguous);
}

#endif  // AZ_COOP_H_
