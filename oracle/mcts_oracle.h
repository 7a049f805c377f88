/* mcts_oracle.h -- interface of the CPU checker (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C, one-lane-at-a-time restatement of the batched MCTS of
 * Matt00n/alphazero (mctx_dist/_src/{search,action_selection,qtransforms,
 * seq_halving,policies,tree}.py, alphazero.py:287-403 and the three gymnax
 * classic-control environments).  Only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py may load this library; the product never
 * calls it.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors and its JAX
 * runtime is not installable here, so this restatement is pinned only by the
 * known-answer cases derived by hand from the reference source
 * (tests/test_oracle_kats.py) and by a second, independent NumPy restatement
 * (oracle/mctx_numpy.py).
 */
#ifndef MCTS_ORACLE_H_
#define MCTS_ORACLE_H_
#include <stdint.h>

#define OZ_MAX_LAYERS 8
#define OZ_MAX_A 64
#define OZ_MAX_K 64

enum { OZ_VARIANT_MUZERO = 0, OZ_VARIANT_SAMPLED = 1, OZ_VARIANT_GUMBEL = 2, OZ_VARIANT_STOCHASTIC = 3 };
enum { OZ_QT_PARENT_SIBLINGS = 0, OZ_QT_COMPLETED_MIX = 1, OZ_QT_MIN_MAX = 2 };
enum { OZ_ENV_NONE = 0, OZ_ENV_CARTPOLE = 1, OZ_ENV_MOUNTAINCAR = 2, OZ_ENV_ACROBOT = 3,
       OZ_ENV_ACROBOT_NOTERM = 4 /* acrobot_no_term.py: Acrobot with done = 0 */ };
enum { OZ_NOISE_THREEFRY = 0, OZ_NOISE_INJECTED = 1, OZ_NOISE_ZERO = 2 };

typedef struct {
  int32_t B, S, A, K;
  int32_t variant;
  int32_t max_depth;
  int32_t qtransform;
  float pb_c_init, pb_c_base;
  float q_epsilon;
  float value_scale, maxvisit_init;
  int32_t rescale_values, use_mixed_value;
  float min_value, max_value;
  int32_t max_num_considered;
  int32_t noise_mode;  /* PUCT tie-break noise source */
  int32_t noise_depth; /* Dn of an injected noise tensor [B,S,Dn,A] */
  uint32_t key[2];     /* the rng_key handed to search() */
  int32_t num_decision_actions; /* OZ_VARIANT_STOCHASTIC: of the A = AD + C columns the first AD are
                                 * actions, the rest chance outcomes (policies.py:584-589) */
} oz_cfg;

/* Tree arrays in the reference's layout (mctx_dist/_src/tree.py:60-73). */
typedef struct {
  int32_t* node_visits;           /* [B,N]     */
  float* raw_values;              /* [B,N,K]   */
  float* node_values;             /* [B,N,K]   */
  int32_t* parents;               /* [B,N]     */
  int32_t* action_from_parent;    /* [B,N]     */
  int32_t* children_index;        /* [B,N,A]   */
  float* children_prior_logits;   /* [B,N,A]   */
  int32_t* children_visits;       /* [B,N,A]   */
  float* children_rewards;        /* [B,N,A]   */
  float* children_discounts;      /* [B,N,A]   */
  float* children_values;         /* [B,N,A,K] */
  float* emb_state;               /* [B,N,4] native env state, or NULL */
  int32_t* emb_time;              /* [B,N], or NULL */
} oz_tree;

/* Dense network in flax layout: kernel[in,out] row-major, bias[out]. */
typedef struct {
  int32_t n_layers;
  int32_t sizes[OZ_MAX_LAYERS + 1];
  const float* w[OZ_MAX_LAYERS];
  const float* b[OZ_MAX_LAYERS];
} oz_mlp;

typedef struct {
  int32_t env_id;
  float gamma;             /* Config.n_step_gamma, alphazero.py:329 */
  oz_mlp policy, value;
  const float* norm_mean;  /* [obs] */
  const float* norm_std;   /* [obs] */
} oz_model;

/* Evaluations injected per (lane, simulation) instead of a model. */
typedef struct {
  const float* prior_logits; /* [B,S,A] */
  const float* value;        /* [B,S,K] */
  const float* reward;       /* [B,S]   */
  const float* discount;     /* [B,S]   */
} oz_injected;

/* Stochastic MuZero: what the two node functions return per (lane, simulation)
 * (mctx_dist/_src/base.py:103-139); the parent's kind picks one of them. */
typedef struct {
  const float* chance_logits;    /* [B,S,C] DecisionRecurrentFnOutput */
  const float* afterstate_value; /* [B,S,K] */
  const float* action_logits;    /* [B,S,AD] ChanceRecurrentFnOutput */
  const float* value;            /* [B,S,K]  */
  const float* reward;           /* [B,S]    */
  const float* discount;         /* [B,S]    */
} oz_injected_stochastic;

#endif
