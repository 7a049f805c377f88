// az_wave.h -- cooperative form of the tree search: 32 threads per lane, two lanes per wave,
// and nothing inside a simulation that crosses a wave.
//
// The literal search (az_device.h) walks root -> leaf re-evaluating the selection rule at every
// level and then walks leaf -> root again for the backup.  The selection at a node depends only
// on that node's own statistics, and those change only when the node lies on a backup path.
// So this form keeps, per node, the action the rule currently selects together with the child it
// leads to (`selx`, in LDS), and a simulation becomes
//   1. chase   : follow `selx` from the root -- one dependent LDS read per level -- and record
//                the path (search.py:127-187);
//   2. evaluate: the caller's recurrent_fn (injected arrays, or environment + networks);
//   3. update  : the 32 threads of the lane take one path position each (deeper paths take
//                several passes, deepest block first): load the node's own row, run the backup's
//                value chain (search.py:265-286) as a systolic pass over the wave's lanes --
//                position p+1 hands reward + discount * value to position p through a
//                one-lane DPP shift -- apply expand / backward to the row, and re-evaluate the
//                selection rule on the updated row.
// A thread reads no row but its own: the updated value of its child arrives by the same DPP
// shift.  Both lanes of a wave and all their positions live in that wave, so the simulation
// loop needs no workgroup barrier; LDS accesses of one wave execute in program order.
// Every floating-point operation is the same, on the same operands and in the same order, as in
// the literal form, so the resulting tree is bit-identical.
//
// Per-visit randomness: the PUCT rules add 1e-7 * uniform noise to break ties.  A cached pick is
// marked ambiguous when that noise could change it (the top two scores closer than the noise),
// and the chase then evaluates that node exactly, with the visit's own noise, as the literal
// form does; everywhere else the noise cannot matter.  The kernels are instantiated twice
// (NOISY): the form without per-visit noise (Gumbel, or PUCT with the noise disabled) carries
// none of that code.
//
// Applicability: A <= 8 and K <= 8, or K = 16 with A = 2 or 3 (vectors in registers);
// N <= 65535 (16-bit node codes).
#ifndef AZ_WAVE_H_
#define AZ_WAVE_H_

#include "az_device.h"

#define AZ_POS 32          // path positions (threads) per lane and pass
#define AZ_WG_LANES 16     // lanes per workgroup (8 waves x 2 lanes)
#define AZ_TREE_THREADS (AZ_POS * AZ_WG_LANES)
#define AZ_SEL_AMBIG (1 << 30)

// Hook of the diagnostic (cycle-stamp) builds; expands to nothing otherwise.
#ifndef AZ_WAVE_STAMP
#define AZ_WAVE_STAMP(k) \
  do {                   \
  } while (0)
#endif

#ifdef AZ_STAMP
// Diagnostic build only: [0] walks repeated level by level, [1] exact selections, [2] key splits
// of those, [3] walks in all.
static __device__ unsigned long long az_dbg_flag_counts[4];
#define AZ_DBG_COUNT(i, n) atomicAdd(&az_dbg_flag_counts[i], (unsigned long long)(n))
#else
#define AZ_DBG_COUNT(i, n) \
  do {                     \
  } while (0)
#endif

// Compiler-level ordering point between LDS / memory accesses of different threads of ONE wave
// (the hardware already executes a wave's memory instructions in program order).
AZ_DEV void az_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Value of wave lane i + 1 (0 for lane 63): one VALU instruction (DPP wave_shl:1).
AZ_DEV float az_from_next_lane(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}

// Per-lane words of LDS.  selx[k] is the cached selection of node k - 1, four words:
//   .x = (action << 16) | (child + 1) [| AZ_SEL_AMBIG]   the node's own selection
//   .y, .z, .w = the .x words of the next three nodes that selection chain leads to (0 from the
//                first unvisited child on)
// so that the walk covers four levels per LDS round trip.  selx[0] = 0 is what a finished walk
// keeps reading.  path[t] is the selection word (.x form) of the node at level t.
typedef __attribute__((ext_vector_type(4))) int az_i32x4;
typedef __attribute__((ext_vector_type(4))) float az_f32x4_;
struct WaveLds {
  az_i32x4* selx;     // [S + 2]
  int32_t* path;      // [S + 8]
  float* root_g;      // [8] the root's Gumbel noise
  float* root_inv;    // [8] the root's invalid-action mask
  float* root_glg;    // [8] gumbel + (logits - max logits) of the root (seq_halving.py:30-32)
  int32_t* hdr;       // [8] 4, 5: reward / discount bits of the evaluation; 7: table row
  int32_t* root_tab;  // [S] the row of the considered-visits table this lane uses
  __host__ __device__ static inline int lane_words(int S) { return (6 * S + 48 + 3) & ~3; }  // 16-byte rows
  AZ_DEV WaveLds(float* base, int lane, int S) {
    float* p = base + (size_t)lane * lane_words(S);
    selx = reinterpret_cast<az_i32x4*>(p);
    path = reinterpret_cast<int32_t*>(p) + 4 * (S + 2);
    root_g = p + 5 * S + 16;
    root_inv = root_g + 8;
    root_glg = root_inv + 8;
    hdr = reinterpret_cast<int32_t*>(root_glg + 8);
    root_tab = hdr + 8;
  }
  // Copies of the root-only selection inputs (filled by az_wave_init): the root is re-selected
  // every simulation.
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

// Words of the per-simulation key table behind the lanes' WaveLds blocks (two per simulation,
// rounded so that what follows it starts on a 16-byte boundary).
__host__ __device__ inline int az_key_words(int S) { return (2 * S + 3) & ~3; }

// The statistics arrays of a workgroup's lanes staged in LDS for the whole search (when they
// fit): the search then touches HBM twice per array -- one coalesced read of the fresh tree and
// one coalesced write of the finished one (the 16 lanes of a workgroup are contiguous in every
// [B, N, ...] array) -- instead of a cache line per node visit.  `view()` is an az_tree over the
// copies, indexed by the lane's position in the workgroup.
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
  // copies the lanes [b0, b0 + lanes) between the global tree and LDS (workgroup-wide).
  // 16 bytes per access, and the first access of every array is issued before any of them is
  // waited for: a plain word-by-word loop per array pays the memory latency once per word and
  // thread (measured: 32 k cycles each way for 70 KB, a sixth of the whole search).
  template <bool TO_LDS>
  AZ_DEV void copy(const az_tree& t, int b0, int lanes, int tid, int nthreads) const {
    const size_t o = (size_t)b0 * N;
    const int n1 = lanes * N, nA = n1 * A, nK = n1 * K, nAK = nA * K;
    constexpr int kArrays = 11;
    // (int32 and float arrays alike are moved as words)
    uint32_t* const in_lds[kArrays] = {(uint32_t*)nv,   (uint32_t*)par,  (uint32_t*)afp,  (uint32_t*)cidx,
                                       (uint32_t*)cvis, (uint32_t*)raw,  (uint32_t*)nval, (uint32_t*)clog,
                                       (uint32_t*)crew, (uint32_t*)cdisc, (uint32_t*)cval};
    uint32_t* const in_glob[kArrays] = {(uint32_t*)(t.node_visits + o),
                                        (uint32_t*)(t.parents + o),
                                        (uint32_t*)(t.action_from_parent + o),
                                        (uint32_t*)(t.children_index + o * A),
                                        (uint32_t*)(t.children_visits + o * A),
                                        (uint32_t*)(t.raw_values + o * K),
                                        (uint32_t*)(t.node_values + o * K),
                                        (uint32_t*)(t.children_prior_logits + o * A),
                                        (uint32_t*)(t.children_rewards + o * A),
                                        (uint32_t*)(t.children_discounts + o * A),
                                        (uint32_t*)(t.children_values + o * A * K)};
    const int count[kArrays] = {n1, n1, n1, nA, nA, nK, nK, nA, nA, nA, nAK};
    // every array's part starts on a 16-byte boundary on both sides: 16 lanes x N nodes x 4 bytes
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 first[kArrays];
#pragma unroll
    for (int k = 0; k < kArrays; ++k) {
      const u32x4* src = reinterpret_cast<const u32x4*>(TO_LDS ? in_glob[k] : in_lds[k]);
      if (tid < (count[k] >> 2)) first[k] = src[tid];
    }
#pragma unroll
    for (int k = 0; k < kArrays; ++k) {
      u32x4* dst = reinterpret_cast<u32x4*>(TO_LDS ? in_lds[k] : in_glob[k]);
      if (tid < (count[k] >> 2)) dst[tid] = first[k];
    }
#pragma unroll
    for (int k = 0; k < kArrays; ++k) {
      const uint32_t* src = TO_LDS ? in_glob[k] : in_lds[k];
      uint32_t* dst = TO_LDS ? in_lds[k] : in_glob[k];
      const int n4 = count[k] >> 2;
      for (int i = tid + nthreads; i < n4; i += nthreads)
        reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
      for (int i = 4 * n4 + tid; i < count[k]; i += nthreads) dst[i] = src[i];
    }
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

// Array-of-structures rows in LDS for two actions and one value atom (the tree-only kernel's
// form of the staged tree): the 15 words of a node -- plus its prior probabilities,
// max(tiny, softmax(prior logits)) -- side by side in 64 bytes, so that a thread fetches its row
// with four 16-byte reads from one address and puts it back with four 16-byte writes, instead of
// nine reads / up to eleven writes with an address computation each.
//   w0 node_visits   w1 (child0 + 1) | (child1 + 1) << 16   w2 visits0   w3 visits1
//   w4 reward0  w5 reward1  w6 discount0  w7 discount1
//   w8 child value0  w9 child value1  w10 logits0  w11 logits1
//   w12 prob0  w13 prob1  w14 raw value  w15 node value
#define AZ_AOS_WORDS 20  // row stride: 16 words + 4 of padding -- rows 64 bytes apart would put the same
                         // 16-byte part of every row on two groups of four banks (measured: row loads
                         // 0.56 k -> 0.8 k cycles); with 80 bytes eight consecutive rows cover all 32 banks
#define AZ_AOS_LOGITS 10
#define AZ_AOS_PROBS 12
#define AZ_AOS_RAW 14
AZ_DEV void az_aos21_load(const float* row, FullRow<2, 1>& f) {
  const az_i32x4 a = *reinterpret_cast<const az_i32x4*>(row);
  const az_f32x4_ b = *reinterpret_cast<const az_f32x4_*>(row + 4);
  const az_f32x4_ c = *reinterpret_cast<const az_f32x4_*>(row + 8);
  const az_f32x4_ d = *reinterpret_cast<const az_f32x4_*>(row + 12);
  f.nv = a.x;
  f.cidx[0] = (a.y & 0xffff) - 1;
  f.cidx[1] = (int32_t)((uint32_t)a.y >> 16) - 1;
  f.vis[0] = a.z;
  f.vis[1] = a.w;
  f.rew[0] = b.x;
  f.rew[1] = b.y;
  f.disc[0] = b.z;
  f.disc[1] = b.w;
  f.cval[0][0] = c.x;
  f.cval[1][0] = c.y;
  f.logits[0] = c.z;
  f.logits[1] = c.w;
  f.raw[0] = d.z;
  f.nval[0] = d.w;
}
AZ_DEV void az_aos21_store(float* row, const FullRow<2, 1>& f, float prob0, float prob1) {
  const az_i32x4 a = {f.nv, (f.cidx[0] + 1) | ((f.cidx[1] + 1) << 16), f.vis[0], f.vis[1]};
  *reinterpret_cast<az_i32x4*>(row) = a;
  *reinterpret_cast<az_f32x4_*>(row + 4) = az_f32x4_{f.rew[0], f.rew[1], f.disc[0], f.disc[1]};
  *reinterpret_cast<az_f32x4_*>(row + 8) = az_f32x4_{f.cval[0][0], f.cval[1][0], f.logits[0], f.logits[1]};
  *reinterpret_cast<az_f32x4_*>(row + 12) = az_f32x4_{prob0, prob1, f.raw[0], f.nval[0]};
}

// The selection inputs of a row held in registers; same arithmetic as az_load_row (a mean over
// one atom divides by 1.0f, which changes no bit and is left out).
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
      r.q[a] = KMAX == 1 ? s : s / (float)K;
    }
  float sv = 0.0f, sr = 0.0f;
AZ_UNROLL_K
  for (int k = 0; k < AZ_TRIP_K; ++k)
    if (k < K) {
      sv = sv + f.nval[k];
      sr = sr + f.raw[k];
    }
  r.vmean = KMAX == 1 ? sv : sv / (float)K;
  r.rawmean = KMAX == 1 ? sr : sr / (float)K;
  r.node_visit = f.nv;
}

// The action the search's selection rule picks at a node (root: depth 0), evaluated WITHOUT
// per-visit noise.  Root and interior positions share a wave, so the two Gumbel rules are
// written as one qtransform followed by two short tails instead of two whole rules.
// `ambiguous` tells whether the PUCT tie-break noise (1e-7 * u, u in [0, 1)) could change the
// pick: it cannot when the best score leads every other one by more than the noise plus the
// rounding of the additions; otherwise the chase re-evaluates the node with the visit's actual
// noise (az_wave_exact_select).
template <int AMAX, bool NOISY>
AZ_DEV int az_select_cached(const SearchCtx& c, const RootCache& rc, int A, bool root,
                            const NodeRow<AMAX>& r, bool& ambiguous) {
  ambiguous = false;
  if (c.d.variant == AZ_VARIANT_GUMBEL) {
    float cq[AMAX];
    az_qtransform<AMAX>(c.d, A, r, cq);
    int best = az_gumbel_interior_tail<AMAX>(A, r, cq);
    if (root) best = az_gumbel_root_tail<AMAX>(c, A, r, cq, rc);
    return best;
  }
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

AZ_DEV int32_t az_pack_sel(int action, int child, bool ambiguous = false) {
  return (action << 16) | (child + 1) | (ambiguous ? AZ_SEL_AMBIG : 0);
}

// The selection at `node` with the tie-break noise of visit (sim, depth), as the literal search
// evaluates it (az_simulate_lane).  (k0, k1) is the lane's simulate key advanced to `key_depth`
// levels; it is advanced further as needed.
template <int AMAX, int KMAX>
AZ_DEV int az_wave_exact_select(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const RootCache& rc,
                                int b, int sim, int node, int depth, uint32_t& k0, uint32_t& k1,
                                int& key_depth, int& child) {
  const int A = l.A;
  NodeRow<AMAX> r;
  az_load_row<AMAX, KMAX>(l, node, r);
  AZ_DBG_COUNT(1, 1);
  if (c.d.noise_mode == AZ_NOISE_THREEFRY)
    AZ_DBG_COUNT(2, depth - key_depth);
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
  const int act = az_select_puct<AMAX>(c, A, depth, r, noise, rc);
  child = l.cidx[node * A + act];
  return act;
}

// Before the first simulation (all threads of a wave): the root-only inputs into LDS and the
// selection at the root of the fresh tree.
// `aos_rows` (or null): the lane's array-of-structures rows; the root's row is read from there.
template <int AMAX, int KMAX, bool NOISY>
AZ_DEV void az_wave_init(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const WaveLds& s, int b, int j,
                         bool active, const float* aos_rows = nullptr) {
  const int A = l.A;
  // The lane's row of the considered-visits table, by all of the lane's threads: S words that one
  // thread would fetch one memory latency after the other.
  int32_t nc = -1;
  if (active && c.d.variant == AZ_VARIANT_GUMBEL && c.in.table) {
    float nvalid = 0.0f;
    for (int a = 0; a < A; ++a)
      nvalid = nvalid + (1.0f - (c.in.invalid_actions ? c.in.invalid_actions[(size_t)b * A + a] : 0.0f));
    nc = az_num_considered(c.d, nvalid);
    for (int i = j; i < c.d.S; i += AZ_POS) s.root_tab[i] = c.in.table[(size_t)nc * c.d.S + i];
  }
  az_wave_sync();
  if (j != 0 || !active) return;
  for (int a = 0; a < A; ++a) {
    s.root_g[a] = c.in.root_gumbel ? c.in.root_gumbel[(size_t)b * A + a] : 0.0f;
    s.root_inv[a] = c.in.invalid_actions ? c.in.invalid_actions[(size_t)b * A + a] : 0.0f;
  }
  FullRow<AMAX, KMAX> f;
  if constexpr (AMAX == 2 && KMAX == 1) {
    if (aos_rows != nullptr)
      az_aos21_load(aos_rows, f);
    else
      az_load_full_row<AMAX, KMAX>(l, AZ_ROOT, f);
  } else {
    az_load_full_row<AMAX, KMAX>(l, AZ_ROOT, f);
  }
  {  // the part of the root's score that no simulation changes (az_score_considered)
    float mx = f.logits[0];
AZ_UNROLL_A
    for (int a = 1; a < AZ_TRIP_A; ++a)
      if (a < A) mx = az_maxf(mx, f.logits[a]);
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) s.root_glg[a] = s.root_g[a] + (f.logits[a] - mx);
  }
  s.hdr[7] = nc;
  NodeRow<AMAX> r;
  az_node_row<AMAX, KMAX>(f, A, l.K, r);
  bool ambiguous;
  const int a = az_select_cached<AMAX, NOISY>(c, s.root_cache(c), A, true, r, ambiguous);
  int child = AZ_UNVISITED;
AZ_UNROLL_A
  for (int i = 0; i < AZ_TRIP_A; ++i)
    if (i == a) child = f.cidx[i];
  s.selx[AZ_ROOT + 1] = az_i32x4{az_pack_sel(a, child, ambiguous), 0, 0, 0};  // no child of a fresh root is visited
}

// The simulate keys of all S simulations into `sim_keys` ([S][2] words of LDS): the rng chain
// rng, simulate_key, expand_key = split(rng, 3) (search.py:90) is sequential, so one thread walks
// it and the others derive their simulation's key from it.  Workgroup-wide call (barriers).
// `given` (or null): the finished keys, worked out by the host (az_sim_keys); copied as they are.
AZ_DEV void az_wave_sim_keys(const az_search_desc& d, uint32_t* sim_keys, int tid, int nthreads,
                             const uint32_t* given = nullptr) {
  if (given != nullptr) {
    for (int i = tid; i < 2 * d.S; i += nthreads) sim_keys[i] = given[i];
    __syncthreads();
    return;
  }
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

// What the walk leaves behind for the update.
struct WavePath {
  int dep;         // levels walked = depth of the evaluated node (0: inactive lane)
  int dep_max;     // the larger `dep` of the wave's two lanes (wave-uniform)
  int32_t e_prev;  // the selection word the walk read at level j - 1 (j = the thread's position in pass 0)
  int32_t e_mine;  // ... and at level j
};

// Depth of the recorded path of both lanes of the wave after a walk of `t_exit` levels
// (entries [0, t_exit) of `path` are this simulation's; the walk wrote 0 past a lane's end).
AZ_DEV void az_wave_path_depth(const WaveLds& s, int h, int j, int t_exit, int max_depth,
                               bool active, WavePath& w) {
  // position p lies on the path iff the walk reached it: p == 0, or level p - 1 led to a
  // visited child and the depth cap (search.py:183) allowed another level
  bool on = active && j < t_exit && (j == 0 || ((w.e_prev & 0xffff) != 0 && j < max_depth));
  unsigned long long m = __builtin_amdgcn_ballot_w64(on);
  int d0 = __builtin_popcount((unsigned)m), d1 = __builtin_popcount((unsigned)(m >> 32));
  for (int base = AZ_POS; base < t_exit; base += AZ_POS) {  // paths deeper than one pass
    const int p = base + j;
    on = active && p < t_exit && (s.path[p - 1] & 0xffff) != 0 && p < max_depth;
    m = __builtin_amdgcn_ballot_w64(on);
    d0 += __builtin_popcount((unsigned)m);
    d1 += __builtin_popcount((unsigned)(m >> 32));
  }
  w.dep = h ? d1 : d0;
  w.dep_max = d0 > d1 ? d0 : d1;
}

// Step 1 for the rules without per-visit noise: thread 0 of each lane follows the cached
// selections from the root, four levels per LDS round trip, and records the selection word of
// every level.  Only the LDS read, one mask and the address are on the dependent chain; the loop
// ends on one scalar test per eight levels, a lane whose walk has ended keeps reading
// selx[0] = 0 meanwhile, and levels past the depth cap (search.py:183) are walked too and
// discarded by az_wave_path_depth.
// `between`: work of the caller that does not depend on the path (requesting the next evaluation),
// placed between the issue of the path's read-back and its first use.
template <typename Between>
AZ_DEV WavePath az_wave_chase_lean(const WaveLds& s, int h, int j, int sim, int max_depth, bool active,
                                   Between&& between) {
  (void)sim;  // named by the stamp hook of diagnostic builds
  int t = 0;
  if (j == 0) {
    int k = active ? 1 : 0;
    az_i32x4* path4 = reinterpret_cast<az_i32x4*>(s.path);
    do {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const az_i32x4 e = s.selx[k];  // this level and the next three
        path4[t / 4 + u] = e;
        k = e.w & 0xffff;
      }
      t += 8;
    } while (__builtin_amdgcn_ballot_w64(k != 0) != 0 && t < max_depth);
  }
  t = __builtin_amdgcn_readlane(t, 0);
  az_wave_sync();
  AZ_WAVE_STAMP(1);
  WavePath w;
  w.e_prev = s.path[j > 0 ? j - 1 : 0];
  w.e_mine = s.path[j];
  between();
  az_wave_path_depth(s, h, j, t, max_depth, active, w);
  return w;
}
AZ_DEV WavePath az_wave_chase_lean(const WaveLds& s, int h, int j, int sim, int max_depth, bool active) {
  return az_wave_chase_lean(s, h, j, sim, max_depth, active, [] {});
}

// The same walk for the PUCT rules with per-visit noise: a cached pick that the noise could
// change is re-evaluated with the visit's own noise.  `sim_keys` ([S][2] words, or null): the
// per-simulation simulate keys (search.py:90).
template <int AMAX, int KMAX>
AZ_DEV WavePath az_wave_chase_noisy(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const WaveLds& s,
                                    int b, int h, int j, int sim, int max_depth, bool active,
                                    const uint32_t* sim_keys) {
  int dep = 0;
  if (active && j == 0) {
    const RootCache rc = s.root_cache(c);
    int node = AZ_ROOT;
    uint32_t k0 = 0u, k1 = 0u;
    int key_depth = -1;  // -1: the lane's key of this simulation has not been derived yet
    for (;;) {
      int32_t e = s.selx[node + 1].x;
      if (e & AZ_SEL_AMBIG) {
        if (key_depth < 0) {
          if (sim_keys) az_key_split(sim_keys[2 * sim], sim_keys[2 * sim + 1], (uint32_t)b, &k0, &k1);
          key_depth = 0;
        }
        int child;
        const int act = az_wave_exact_select<AMAX, KMAX>(c, l, rc, b, sim, node, dep, k0, k1, key_depth, child);
        e = az_pack_sel(act, child);
      }
      s.path[dep] = e;
      dep += 1;
      const int nxt = (e & 0xffff) - 1;
      if (nxt < 0 || dep >= max_depth) break;
      node = nxt;
    }
  }
  az_wave_sync();
  const int d0 = __builtin_amdgcn_readlane(dep, 0), d1 = __builtin_amdgcn_readlane(dep, 32);
  WavePath w;
  w.dep = h ? d1 : d0;
  w.dep_max = d0 > d1 ? d0 : d1;
  w.e_prev = s.path[j > 0 ? j - 1 : 0];  // entries past the path's end are stale and not used
  w.e_mine = s.path[j];
  return w;
}

// Step 1 for the rules with per-visit noise.  A cached selection that the noise could change
// carries a flag, and hardly any does (a few in a million): the walk is therefore the lean one,
// flags ignored, and only when a flagged entry turns up on a path of the wave is the walk
// repeated level by level with every flagged level decided by the exact rule.
template <int AMAX, int KMAX, typename Between>
AZ_DEV WavePath az_wave_chase_flagged(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const WaveLds& s,
                                      int b, int h, int j, int sim, int max_depth, bool active,
                                      const uint32_t* sim_keys, Between&& between) {
  WavePath w = az_wave_chase_lean(s, h, j, sim, max_depth, active, between);
  bool flagged = active && j < w.dep && (w.e_mine & AZ_SEL_AMBIG) != 0;
  for (int base = AZ_POS; base < w.dep_max; base += AZ_POS) {  // paths deeper than one pass
    const int p = base + j;
    flagged = flagged || (active && p < w.dep && (s.path[p] & AZ_SEL_AMBIG) != 0);
  }
  if (j == 0 && active) AZ_DBG_COUNT(3, 1);
  if (__builtin_amdgcn_ballot_w64(flagged) == 0) return w;
  if (j == 0 && active) AZ_DBG_COUNT(0, 1);
  return az_wave_chase_noisy<AMAX, KMAX>(c, l, s, b, h, j, sim, max_depth, active, sim_keys);
}

// The node the evaluation expands: parent = the last node of the path, action = the action
// taken there, node = the child it leads to, or node `sim + 1` when that edge is unvisited
// (search.py:97-99).  Valid in every thread of an active lane.
struct WaveLeaf {
  int parent, action, node;
};
AZ_DEV WaveLeaf az_wave_leaf(const WaveLds& s, int dep, int sim) {
  WaveLeaf w;
  const int32_t e = s.path[dep - 1];
  const int kc = e & 0xffff;
  w.action = (e >> 16) & 0xff;
  w.node = kc == 0 ? sim + 1 : kc - 1;
  w.parent = dep > 1 ? (s.path[dep - 2] & 0xffff) - 1 : AZ_ROOT;
  return w;
}

// What path position p of a lane works on: its node, the action the path takes there, and the
// node's row.  Positions past the path's end (and inactive lanes) read the root's row, which is
// always valid memory, and are kept from writing by the caller.
template <int AMAX, int KMAX>
struct WaveRow {
  FullRow<AMAX, KMAX> f;
  int node, act;
  int e_mine_code;  // code (index + 1) of the path's next node; meaningful below the parent
};

template <int AMAX, int KMAX>
AZ_DEV void az_wave_load(const Lane<AMAX, KMAX>& l, bool active, int p, int dep, int sim, int32_t e_prev,
                         int32_t e_mine, WaveRow<AMAX, KMAX>& w, const float* aos_rows = nullptr) {
  const bool mine = active && p <= dep, leaf = p == dep;
  const int kc = p > 0 ? (e_prev & 0xffff) : 1;
  const int on_path = kc - 1;
  const int fresh = kc == 0 ? sim + 1 : kc - 1;  // search.py:97-99
  w.node = mine ? (leaf ? fresh : on_path) : AZ_ROOT;
  w.act = mine && !leaf ? (e_mine >> 16) & 0xff : 0;
  w.e_mine_code = e_mine & 0xffff;
  if constexpr (AMAX == 2 && KMAX == 1) {
    if (aos_rows != nullptr) {
      az_aos21_load(aos_rows + w.node * AZ_AOS_WORDS, w.f);
      return;
    }
  }
  az_load_full_row<AMAX, KMAX>(l, w.node, w.f);
}

// The Gumbel rules with qtransform_completed_by_mix_value at its defaults
// (action_selection.py:148-240, qtransforms.py:90-198), split in two so that the part that does
// not depend on the backup -- the prior softmax, everything derived from the visit counts --
// can be issued ahead of it and overlap with it.  The operations, their operands and their
// order are those of az_qt_completed_mix / az_gumbel_interior_tail / az_gumbel_root_tail; only
// their placement differs, and nothing is evaluated under a branch.
template <int AMAX>
struct GumbelPre {
  float probs[AMAX];  // max(tiny, softmax(prior logits))
  float ratio[AMAX];  // visits / (1 + sum of visits)
  float sum_probs, fsum, fsum1, scale;
  int32_t sum_visits;
  int32_t considered_visit;  // the root rule's table entry for this visit count (seq_halving.py:36-70)
};

// IEEE 754-2019 maximum / minimum (one instruction each on gfx950).  Like jnp.maximum / minimum
// (az_maxf / az_minf) they return NaN when either operand is NaN; they differ from them in the
// sign of a zero result and in the NaN's payload only, neither of which reaches the picked action.
AZ_DEV float az_max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
AZ_DEV float az_min_nan(float a, float b) { return __builtin_elementwise_minimum(a, b); }

// jax.nn.softmax as az_softmax, with the one-instruction maximum.
template <int AMAX>
AZ_DEV void az_softmax_fast(const float (&x)[AMAX], int A, float (&out)[AMAX]) {
  if (AMAX == 2 && A == 2) {
    // Two actions: the larger one's exponential is exp(m - m), i.e. 1 (or NaN when m is not
    // finite: t below), so only the smaller one's is evaluated; 1 + e and e + 1 are the same sum.
    const float m = az_max_nan(x[0], x[1]), lo = az_min_nan(x[0], x[1]);
    const float t = m - m;               // +0, or NaN for an infinite or NaN maximum
    const float e = az_expf(lo - m) + t;  // (e + 0 is e)
    const float one = 1.0f + t;
    const bool first = x[0] >= x[1];  // equal inputs: both exponentials are 1 either way
    const float e0 = first ? one : e, e1 = first ? e : one;
    const float s = e0 + e1;
    out[0] = e0 / s;
    out[1] = e1 / s;
    return;
  }
  float m = x[0];
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a)
    if (a < A) m = az_max_nan(m, x[a]);
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

// An optimisation barrier: the value is computed here, unconditionally.
AZ_DEV float az_pin(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// vis: the children's visit counts AFTER this simulation's backup (they are known from the path).
// The prior part of the rule, max(tiny, softmax(prior logits)) (qtransforms.py:121-123): a
// function of the node's prior logits alone, which change only when the node is expanded.
template <int AMAX>
AZ_DEV void az_prior_probs(const float (&logits)[AMAX], int A, float (&probs)[AMAX]) {
  az_softmax_fast<AMAX>(logits, A, probs);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) probs[a] = az_max_nan(AZ_FLT_TINY, probs[a]);
}

// `cached_probs` (or null): az_prior_probs of these logits, worked out earlier.
template <int AMAX>
AZ_DEV void az_gumbel_pre(const az_search_desc& d, int A, const float (&logits)[AMAX],
                          const int32_t (&vis)[AMAX], GumbelPre<AMAX>& g, const int32_t* table_row,
                          const float* cached_probs = nullptr) {
  if (cached_probs != nullptr) {
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) g.probs[a] = cached_probs[a];
  } else {
    az_prior_probs<AMAX>(logits, A, g.probs);
  }
  int32_t sum_visits = 0, maxvisit = vis[0];
  g.sum_probs = 0.0f;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      sum_visits += vis[a];
      if (a > 0 && vis[a] > maxvisit) maxvisit = vis[a];
      g.sum_probs = g.sum_probs + (vis[a] > 0 ? g.probs[a] : 0.0f);
    }
  g.sum_visits = sum_visits;
  {
    // requested here by every thread (only the root's is used) so that the read is not waited
    // for inside the root's branch behind the backup; every index is inside the row
    int32_t sim_index = sum_visits;
    if (sim_index > d.S - 1) sim_index = d.S - 1;
    if (sim_index < 0) sim_index = 0;
    g.considered_visit = table_row[sim_index];
  }
  g.fsum = (float)sum_visits;
  g.fsum1 = (float)(sum_visits + 1);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) g.ratio[a] = (float)vis[a] / g.fsum1;
  g.scale = (d.maxvisit_init + (float)maxvisit) * d.value_scale;
}

// Pins the results of az_gumbel_pre where this is called: without it the compiler sinks the
// whole computation to its uses behind the backup.
template <int AMAX>
AZ_DEV void az_gumbel_pre_pin(int A, GumbelPre<AMAX>& g) {
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      g.ratio[a] = az_pin(g.ratio[a]);
      g.probs[a] = az_pin(g.probs[a]);
    }
  g.sum_probs = az_pin(g.sum_probs);
  g.scale = az_pin(g.scale);
  asm volatile("" : "+v"(g.considered_visit));
}

// q: mean over atoms of reward + discount * child value per action; raw: mean raw value.
template <int AMAX>
AZ_DEV int az_gumbel_post(const SearchCtx& c, int A, const GumbelPre<AMAX>& g, const float (&logits)[AMAX],
                          const int32_t (&vis)[AMAX], const float (&q)[AMAX], float raw, bool root,
                          const RootCache& rc, const float* glg) {
  const az_search_desc& d = c.d;
  float weighted_q = 0.0f;
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) {
      const float t = az_pin((g.probs[a] * q[a]) / g.sum_probs);
      weighted_q = weighted_q + (vis[a] > 0 ? t : 0.0f);
    }
  const float value = (raw + g.fsum * weighted_q) / g.fsum1;
  float cq[AMAX];
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) cq[a] = vis[a] > 0 ? q[a] : value;
  float mn = cq[0], mx = cq[0];
AZ_UNROLL_A
  for (int a = 1; a < AZ_TRIP_A; ++a)
    if (a < A) {
      mn = az_min_nan(mn, cq[a]);
      mx = az_max_nan(mx, cq[a]);
    }
  const float den = az_max_nan(mx - mn, d.q_epsilon);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) cq[a] = g.scale * ((cq[a] - mn) / den);
  // interior rule (action_selection.py:203-240)
  float x[AMAX], p2[AMAX], to_argmax[AMAX];
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) x[a] = logits[a] + cq[a];
  az_softmax_fast<AMAX>(x, A, p2);
AZ_UNROLL_A
  for (int a = 0; a < AZ_TRIP_A; ++a)
    if (a < A) to_argmax[a] = p2[a] - g.ratio[a];
  int best = az_argmax<AMAX>(to_argmax, A);
  if (root) {
    // root rule (action_selection.py:148-200, seq_halving.py:23-33): the number of considered
    // actions is fixed per search, so the table row is the cached one, and gumbel + (logits -
    // max logits) was added up once by az_wave_init
    const int32_t considered_visit = g.considered_visit;
    float score[AMAX];
AZ_UNROLL_A
    for (int a = 0; a < AZ_TRIP_A; ++a)
      if (a < A) {
        const float penalty = vis[a] == considered_visit ? 0.0f : -AZ_INF;
        score[a] = az_max_nan(-1e9f, glg[a] + cq[a]) + penalty;
        if (rc.has_invalid && rc.invalid[a] != 0.0f) score[a] = -AZ_INF;
      }
    best = az_argmax<AMAX>(score, A);
  }
  return best;
}

// Value of wave lane i + 1 (0 for lane 63), in a form the compiler can fold into the consuming
// multiply as its DPP operand.
AZ_DEV float az_from_next_lane_keep(float x) {
  const int v = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true));
}

// Step 3: expand (search.py:231-244, 304-342) and backward (search.py:265-286) applied to every
// node of the recorded path plus the evaluated node, and their cached selections refreshed.
// Position p of the path (0 = root ... dep-1 = parent, dep = the evaluated node) belongs to
// thread p % 32 of the lane in pass p / 32.  ev_logits [A] / ev_value [K] / reward / discount
// are the recurrent_fn's outputs for this lane (any address space; read by every thread).
// `pre` (or null): the rows of pass 0, loaded by az_wave_load ahead of the evaluation.
// `node_probs` (or null; Gumbel rules): [S + 1][A] words of LDS per lane with az_prior_probs of
// every node's prior logits, slot sim + 1 filled by the caller with those of simulation sim's
// evaluation before that simulation's update (a fresh node IS node sim + 1).
//
// The backup's value chain  lv(p) = reward(p) + discount(p) * lv(p + 1)  runs as a fixed-point
// iteration over the wave's lanes: every step, every thread recomputes its value from the one
// its next lane held a step earlier (one DPP multiply-add pair).  The deepest path position
// reads a constant (the evaluated node's value), so after u steps the u deepest positions hold
// their final values and recomputing them changes nothing; no thread needs to know when its
// turn is.  Threads past the path's end hand the constant through unchanged (-0.0f + 1.0f * x
// is x for every x), and thread 31 of each lane is pinned so that nothing leaks in from the
// neighbouring lane of the wave.
// AOS (two actions, one atom, Gumbel rules): the lane's rows are array-of-structures rows at
// `aos_rows` (az_aos21_load / _store): a thread writes its whole row back once instead of field
// by field, `l` only provides parents / action_from_parent, and the prior probabilities of node n
// are words AZ_AOS_PROBS .. of row n (`node_probs` is not used).
template <int AMAX, int KMAX, bool NOISY, bool FAST_GUMBEL = false, bool AOS = false>
AZ_DEV void az_wave_update(const SearchCtx& c, const Lane<AMAX, KMAX>& l, const WaveLds& s, bool active,
                           int h, int j, int sim, const WavePath& wp, const float* ev_logits,
                           const float* ev_value, float reward, float discount,
                           const WaveRow<AMAX, KMAX>* pre = nullptr, float* node_probs = nullptr,
                           float* aos_rows = nullptr) {
  static_assert(!AOS || (AMAX == 2 && KMAX == 1 && FAST_GUMBEL), "array-of-structures rows: 2 actions, 1 atom, Gumbel");
  const int A = l.A, K = l.K;
  const int dep = wp.dep;
  const RootCache rc = s.root_cache(c);
  const bool top = j == AZ_POS - 1;
  float ev[KMAX], carry_x[KMAX], carry_child[KMAX];
  int32_t carry_sel[3] = {0, 0, 0};
AZ_UNROLL_K
  for (int k = 0; k < AZ_TRIP_K; ++k) {
    ev[k] = k < K ? ev_value[k] : 0.0f;
    carry_x[k] = ev[k];  // backward starts from the evaluated node's value
    carry_child[k] = 0.0f;
  }
  for (int it = wp.dep_max / AZ_POS; it >= 0; --it) {
    const int p = j + it * AZ_POS;
    const bool mine = active && p <= dep;
    const bool leaf = p == dep;
    WaveRow<AMAX, KMAX> w;
    if (pre != nullptr && it == 0) {
      w = *pre;
    } else if (it == 0) {
      az_wave_load<AMAX, KMAX>(l, active, p, dep, sim, wp.e_prev, wp.e_mine, w, AOS ? aos_rows : nullptr);
    } else {
      const int pp = p <= dep ? p : dep;  // keeps the reads of unused positions inside the path
      az_wave_load<AMAX, KMAX>(l, active, p, dep, sim, s.path[pp - 1], s.path[pp], w, AOS ? aos_rows : nullptr);
    }
    FullRow<AMAX, KMAX>& f = w.f;
    int node = w.node;
    const int act = w.act;
    // A row loaded ahead of the evaluation: the addresses of its words were worked out there, and
    // the ones written here are the same -- left visible, they are kept in registers (or spilled)
    // across the networks instead of being worked out again.
    if (pre != nullptr) asm volatile("" : "+v"(node));
    AZ_WAVE_STAMP(3);
    // What the refreshed selection needs besides the backup's result: the evaluated node takes the
    // new prior logits, the path's nodes one more visit of the action the path took.
    GumbelPre<AMAX> gpre;
    if (FAST_GUMBEL) {
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a < A) {
          if (leaf) f.logits[a] = ev_logits[a];
          if (!leaf && a == act) f.vis[a] = f.vis[a] + 1;
        }
      // (slot sim + 1 of node_probs holds the values for this simulation's evaluation)
      const float* cached = nullptr;
      if (AOS)
        cached = aos_rows + (leaf ? sim + 1 : node) * AZ_AOS_WORDS + AZ_AOS_PROBS;
      else if (node_probs != nullptr)
        cached = node_probs + (leaf ? sim + 1 : node) * A;
      az_gumbel_pre<AMAX>(c.d, A, f.logits, f.vis, gpre, rc.table_row, cached);
      az_gumbel_pre_pin<AMAX>(A, gpre);
    }
    // the edge from this position's node to the next node of the path
    const int jstart = dep - 1 - it * AZ_POS < AZ_POS - 1 ? dep - 1 - it * AZ_POS : AZ_POS - 1;
    const int js_max = wp.dep_max - 1 - it * AZ_POS < AZ_POS - 1 ? wp.dep_max - 1 - it * AZ_POS : AZ_POS - 1;
    float r_p = -0.0f, d_p = 1.0f;
    if (active && j <= jstart) {
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a == act) {
          r_p = f.rew[a];
          d_p = f.disc[a];
        }
      if (p == dep - 1) {
        r_p = reward;
        d_p = discount;
      }
    }
    float y[KMAX], pin[KMAX];
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k) {
      y[k] = carry_x[k];
      pin[k] = jstart == AZ_POS - 1 ? r_p + d_p * carry_x[k] : carry_x[k];
    }
    for (int u = 0; u <= js_max; u += 4) {  // steps past the last needed one change nothing
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
AZ_UNROLL_K
        for (int k = 0; k < AZ_TRIP_K; ++k)
          if (k < K) {
            const float v = r_p + d_p * az_from_next_lane_keep(y[k]);
            y[k] = top ? pin[k] : v;
          }
      }
    }
    AZ_WAVE_STAMP(4);
    // the node's new value; its child's new value arrives from the next lane
    const float cnt = (float)f.nv;
    float nval_new[KMAX], child_new[KMAX];
AZ_UNROLL_K
    for (int k = 0; k < AZ_TRIP_K; ++k) {
      nval_new[k] = 0.0f;
      if (k < K) nval_new[k] = leaf ? ev[k] : (f.nval[k] * cnt + y[k]) / (cnt + 1.0f);
      child_new[k] = az_from_next_lane(nval_new[k]);
    }
    if (wp.dep_max >= AZ_POS) {  // several passes: hand the block's lowest position to the next pass
AZ_UNROLL_K
      for (int k = 0; k < AZ_TRIP_K; ++k) {
        if (top) child_new[k] = carry_child[k];
        const float y0 = __shfl(y[k], 0, AZ_POS), nv0 = __shfl(nval_new[k], 0, AZ_POS);
        if (jstart >= 0) carry_x[k] = y0;
        carry_child[k] = nv0;
      }
    }
    AZ_WAVE_STAMP(5);
    // The look-ahead words of the refreshed entry come from the selected child's entry: with few
    // actions the entries of ALL children are requested here, before the rule runs, so that the
    // read's latency is not added behind the rule.
    constexpr bool kEarlyEntries = false;  // (measured: the extra reads cost more than the round trip they hide)
    az_i32x4 child_entry[kEarlyEntries ? AMAX : 1];
    if (kEarlyEntries) {
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a < A) child_entry[a] = s.selx[mine ? f.cidx[a] + 1 : 0];
    }
    int32_t sel_new = 0;
    const WaveLeaf wl = az_wave_leaf(s, dep > 0 ? dep : 1, sim);  // (an inactive lane reads level 0)
    if (mine) {
      // expand at the evaluated node (update_tree_node, search.py:304-342), backward
      // (search.py:265-286) at the nodes of the path
      if (leaf) {
AZ_UNROLL_A
        for (int a = 0; a < AZ_TRIP_A; ++a)
          if (a < A) {
            f.logits[a] = ev_logits[a];
            if (!AOS) l.clog[node * A + a] = f.logits[a];
          }
AZ_UNROLL_K
        for (int k = 0; k < AZ_TRIP_K; ++k)
          if (k < K) {
            f.raw[k] = ev[k];
            if (!AOS) l.raw[(size_t)node * K + k] = ev[k];
          }
        l.par[node] = wl.parent;
        l.afp[node] = wl.action;
        if (FAST_GUMBEL && !AOS && node_probs != nullptr) {
          // an existing node expanded again at the depth cap takes the new evaluation's values
AZ_UNROLL_A
          for (int a = 0; a < AZ_TRIP_A; ++a)
            if (a < A) node_probs[node * A + a] = gpre.probs[a];
        }
      } else {
        if (p == dep - 1) {
          // the edge parent -> evaluated node (search.py:235-241)
          const int newn = wl.node;
AZ_UNROLL_A
          for (int a = 0; a < AZ_TRIP_A; ++a)
            if (a == act) {
              f.cidx[a] = newn;
              f.rew[a] = reward;
              f.disc[a] = discount;
            }
          if (!AOS) {
            l.cidx[node * A + act] = newn;
            l.crew[node * A + act] = reward;
            l.cdisc[node * A + act] = discount;
          }
        }
        int32_t newvis = 0;
AZ_UNROLL_A
        for (int a = 0; a < AZ_TRIP_A; ++a)
          if (a == act) {
AZ_UNROLL_K
            for (int k = 0; k < AZ_TRIP_K; ++k)
              if (k < K) f.cval[a][k] = child_new[k];
            if (!FAST_GUMBEL) f.vis[a] = f.vis[a] + 1;  // (counted ahead of the backup otherwise)
            newvis = f.vis[a];
          }
        if (!AOS) {
AZ_UNROLL_K
          for (int k = 0; k < AZ_TRIP_K; ++k)
            if (k < K) l.cval[((size_t)node * A + act) * K + k] = child_new[k];
          l.cvis[node * A + act] = newvis;
        }
      }
AZ_UNROLL_K
      for (int k = 0; k < AZ_TRIP_K; ++k)
        if (k < K) {
          f.nval[k] = nval_new[k];
          if (!AOS) l.nval[(size_t)node * K + k] = nval_new[k];
        }
      f.nv = f.nv + 1;
      if (!AOS) l.nv[node] = f.nv;
      if constexpr (AOS) az_aos21_store(aos_rows + node * AZ_AOS_WORDS, f, gpre.probs[0], gpre.probs[1]);
      AZ_WAVE_STAMP(6);
      if (!FAST_GUMBEL) {
        // refresh the cached selection of this node
        NodeRow<AMAX> r;
        az_node_row<AMAX, KMAX>(f, A, K, r);
        bool ambiguous;
        const int best = az_select_cached<AMAX, NOISY>(c, rc, A, p == 0, r, ambiguous);
        int child = AZ_UNVISITED;
AZ_UNROLL_A
        for (int a = 0; a < AZ_TRIP_A; ++a)
          if (a == best) child = f.cidx[a];
        sel_new = az_pack_sel(best, child, ambiguous);
      }
    }
    if (FAST_GUMBEL) {
      // refresh the cached selection: every thread runs the rule (no branch around it), the
      // threads without a node on a row nobody uses
      NodeRow<AMAX> r;
      az_node_row<AMAX, KMAX>(f, A, K, r);
      const int best = az_gumbel_post<AMAX>(c, A, gpre, f.logits, f.vis, r.q, r.rawmean, mine && p == 0, rc, s.root_glg);
      int child = AZ_UNVISITED;
AZ_UNROLL_A
      for (int a = 0; a < AZ_TRIP_A; ++a)
        if (a == best) child = f.cidx[a];
      sel_new = mine ? az_pack_sel(best, child) : 0;
    }
    // The other words of the entry: the selections of the next three nodes down the chain.  A
    // child on the path was refreshed just now by the next lane, whose own words two and three
    // settle one round after its first (three rounds of one-lane shifts); any other child's entry
    // did not change in this simulation (a node changes only together with its parent) and is
    // read back from LDS -- the entry of code 0, for an unvisited child, reads as 0.
    const int child_code = sel_new & 0xffff;
    const int next_code = leaf ? 0 : (p == dep - 1 ? wl.node + 1 : w.e_mine_code);
    az_i32x4 entry = {sel_new, 0, 0, 0};
    {
      const bool on_path = child_code != 0 && child_code == next_code;
      az_i32x4 stored;
      if (kEarlyEntries) {
        // (the edge to the evaluated node was unvisited when the entries were requested: its
        // child is on the path, and the stored entry is not used then)
        stored = az_i32x4{0, 0, 0, 0};
        const int best_a = (sel_new >> 16) & 0xff;
AZ_UNROLL_A
        for (int a = 0; a < AZ_TRIP_A; ++a)
          if (a == best_a) stored = child_entry[a];
      } else {
        stored = s.selx[child_code];
      }
      const bool several = wp.dep_max >= AZ_POS;
      int32_t n0 = __builtin_amdgcn_update_dpp(0, entry.x, 0x130, 0xf, 0xf, false);
      if (several && top) n0 = carry_sel[0];
      entry.y = on_path ? n0 : stored.x;
      int32_t n1 = __builtin_amdgcn_update_dpp(0, entry.y, 0x130, 0xf, 0xf, false);
      if (several && top) n1 = carry_sel[1];
      entry.z = on_path ? n1 : stored.y;
      int32_t n2 = __builtin_amdgcn_update_dpp(0, entry.z, 0x130, 0xf, 0xf, false);
      if (several && top) n2 = carry_sel[2];
      entry.w = on_path ? n2 : stored.z;
      if (several) {
        carry_sel[0] = __shfl(entry.x, 0, AZ_POS);
        carry_sel[1] = __shfl(entry.y, 0, AZ_POS);
        carry_sel[2] = __shfl(entry.z, 0, AZ_POS);
      }
    }
    if (mine) s.selx[node + 1] = entry;
  }
  az_wave_sync();
}

// The search descriptor with the fields a compile-time rule fixes replaced by constants, so that
// the selection code folds to that rule alone.  RULE 0: everything at run time; 1: the Gumbel
// rules with qtransform_completed_by_mix_value at its defaults (policies.py:237-250);
// 2: a PUCT rule (muzero or sampled) with qtransform_by_parent_and_siblings (policies.py:30-45).
template <int RULE>
AZ_DEV SearchCtx az_rule_ctx(const SearchCtx& c) {
  SearchCtx cc = c;
  if (RULE == 1) {
    cc.d.variant = AZ_VARIANT_GUMBEL;
    cc.d.qtransform = AZ_QT_COMPLETED_MIX;
    cc.d.use_mixed_value = 1;
    cc.d.rescale_values = 1;
  } else if (RULE == 2) {
    cc.d.qtransform = AZ_QT_PARENT_SIBLINGS;
  }
  return cc;
}
// The rule a descriptor can run under (host side).
inline int az_rule_of(const az_search_desc* d) {
  if (d->variant == AZ_VARIANT_GUMBEL)
    return d->qtransform == AZ_QT_COMPLETED_MIX && d->use_mixed_value && d->rescale_values ? 1 : 0;
  return d->qtransform == AZ_QT_PARENT_SIBLINGS ? 2 : 0;
}

#endif  // AZ_WAVE_H_
