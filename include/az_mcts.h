/* az_mcts.h -- C ABI of libaz_mcts.so, the MI355X (gfx950) implementation of
 * the batched MCTS self-play hot path of Matt00n/alphazero.
 *
 * Every entry point takes plain pointers and sizes.  Unless a parameter says
 * "host", pointers are DEVICE pointers owned by the caller (e.g. the storage
 * of torch tensors); `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Calls enqueue work on `stream` and return without
 * synchronising.  Return value: 0 on success, a negative AZ_ERR_* code
 * otherwise, with a message available from az_last_error() (thread-local).
 * No C++ exception crosses the boundary.
 *
 * The comment on each declaration cites the reference function it replaces
 * (paths relative to the reference checkout).
 */
#ifndef AZ_MCTS_H_
#define AZ_MCTS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_MAX_LAYERS 8
#define AZ_MAX_ACTIONS 64
#define AZ_MAX_ATOMS 64
#define AZ_MAX_WIDTH 1024 /* widest dense layer the fused networks accept */

enum { AZ_OK = 0, AZ_ERR_INVALID = -1, AZ_ERR_UNSUPPORTED = -2, AZ_ERR_HIP = -3 };
enum { AZ_VARIANT_MUZERO = 0, AZ_VARIANT_SAMPLED = 1, AZ_VARIANT_GUMBEL = 2, AZ_VARIANT_STOCHASTIC = 3 };
enum { AZ_QT_PARENT_SIBLINGS = 0, AZ_QT_COMPLETED_MIX = 1, AZ_QT_MIN_MAX = 2 };
/* AZ_ENV_ACROBOT_NOTERM: 'Acrobot-v2' (gymnax registration.py:36-37, acrobot_no_term.py): the
 * dynamics, reward and observation of Acrobot-v1 with `done` always 0 (is_terminal, :151-157);
 * episodes end by truncation only. */
enum { AZ_ENV_CARTPOLE = 1, AZ_ENV_MOUNTAINCAR = 2, AZ_ENV_ACROBOT = 3, AZ_ENV_ACROBOT_NOTERM = 4 };
enum { AZ_NOISE_THREEFRY = 0, AZ_NOISE_INJECTED = 1, AZ_NOISE_ZERO = 2 };

/* Static description of one search() call (mctx_dist/_src/search.py:31-43 plus
 * the keyword arguments the policies bind into the selection functions,
 * mctx_dist/_src/policies.py:93-100, 205-212, 306-315). */
typedef struct az_search_desc {
  int32_t B, S, A, K;          /* lanes, num_simulations, num_actions, value atoms */
  int32_t variant;             /* AZ_VARIANT_*: which pair of selection functions   */
  int32_t max_depth;           /* search.py:83-84: num_simulations when "None"      */
  int32_t qtransform;          /* AZ_QT_*                                           */
  float pb_c_init, pb_c_base;  /* action_selection.py:54-55                         */
  float q_epsilon;             /* qtransforms.py:58, 98                             */
  float value_scale, maxvisit_init;        /* qtransforms.py:94-95                  */
  int32_t rescale_values, use_mixed_value; /* qtransforms.py:96-97                  */
  float min_value, max_value;  /* qtransforms.py:28-29                              */
  int32_t max_num_considered;  /* policies.py:248                                   */
  int32_t noise_mode;          /* AZ_NOISE_*: source of the 1e-7 tie-break noise    */
  int32_t noise_depth;         /* Dn of an injected noise tensor [B,S,Dn,A]         */
  uint32_t key[2];             /* rng_key handed to search()                        */
  int32_t kernel_form;         /* 0: automatic; 1: the one-thread-per-lane kernels  */
  int32_t num_decision_actions; /* AZ_VARIANT_STOCHASTIC (policies.py:346-483): of the A columns the
                                * first num_decision_actions are the actions of decision nodes and the
                                * rest the chance outcomes of chance nodes (policies.py:584-589)       */
} az_search_desc;

/* The search tree, in exactly the reference's array layout
 * (mctx_dist/_src/tree.py:60-73; row-major, int32 / float32).  N = S + 1.
 * emb_state / emb_time hold the native environment state per node (the
 * reference stores its whole env State pytree in Tree.embeddings); they may be
 * NULL for searches whose embeddings live with the caller. */
typedef struct az_tree {
  int32_t* node_visits;         /* [B,N]     */
  float* raw_values;            /* [B,N,K]   */
  float* node_values;           /* [B,N,K]   */
  int32_t* parents;             /* [B,N]     */
  int32_t* action_from_parent;  /* [B,N]     */
  int32_t* children_index;      /* [B,N,A]   */
  float* children_prior_logits; /* [B,N,A]   */
  int32_t* children_visits;     /* [B,N,A]   */
  float* children_rewards;      /* [B,N,A]   */
  float* children_discounts;    /* [B,N,A]   */
  float* children_values;       /* [B,N,A,K] */
  float* emb_state;             /* [B,N,4]   */
  int32_t* emb_time;            /* [B,N]     */
} az_tree;

/* Per-search inputs shared by the search entry points. */
typedef struct az_search_inputs {
  const float* invalid_actions; /* [B,A] 1 = invalid, or NULL (search.py:85-86); [B,num_decision_actions]
                                 * for AZ_VARIANT_STOCHASTIC                                */
  const float* root_gumbel;     /* [B,A] GumbelMuZeroExtraData.root_gumbel, gumbel variant */
  const int32_t* table;         /* [(m+1),S] from az_considered_table, gumbel variant      */
  const float* noise;           /* [B,S,Dn,A] uniform draws when noise_mode = INJECTED
                                 * ([B,S,Dn,num_decision_actions] for AZ_VARIANT_STOCHASTIC) */
  const uint8_t* is_decision;   /* [B,N] 1 = decision node, 0 = chance node (the is_decision_node leaf of
                                 * StochasticRecurrentState, base.py:141-161, kept by the caller next to
                                 * its embeddings); AZ_VARIANT_STOCHASTIC only                          */
  const uint32_t* sim_keys;     /* [S,2] the output of az_sim_keys copied to the device, or NULL: the
                                 * cooperative kernels then derive the keys themselves, one thread
                                 * of every workgroup walking the sequential chain (read with
                                 * noise_mode = THREEFRY only)                                */
} az_search_inputs;

/* Evaluations injected per (lane, simulation) in place of a recurrent_fn. */
typedef struct az_injected {
  const float* prior_logits; /* [B,S,A] */
  const float* value;        /* [B,S,K] */
  const float* reward;       /* [B,S]   */
  const float* discount;     /* [B,S]   */
} az_injected;

/* A dense network as the caller holds it: flax layout kernel[in,out] row-major
 * and bias[out] per layer (networks/networks.py:25-48), device pointers. */
typedef struct az_mlp_params {
  int32_t n_layers;
  int32_t sizes[AZ_MAX_LAYERS + 1]; /* sizes[0] = input width */
  const float* w[AZ_MAX_LAYERS];
  const float* b[AZ_MAX_LAYERS];
} az_mlp_params;

/* The in-tree model of alphazero.py:317-339: environment + normaliser + the
 * policy and value networks, with the weights re-laid-out for the matrix
 * cores by az_model_pack. */
typedef struct az_model {
  int32_t env_id;   /* AZ_ENV_* */
  float gamma;      /* Config.n_step_gamma (alphazero.py:329) */
  int32_t obs_dim, num_actions, num_atoms;
  int32_t policy_layers, value_layers;
  int32_t policy_sizes[AZ_MAX_LAYERS + 1];
  int32_t value_sizes[AZ_MAX_LAYERS + 1];
  int32_t hidden_bf16;    /* 0: float32 throughout (the reference's arithmetic); 1: `packed` comes from
                           * az_model_pack_bf16 and the layers between the first and the last one of
                           * each network run on the bfloat16 matrix instructions */
  const float* packed;    /* device buffer filled by az_model_pack / az_model_pack_bf16 */
  const float* norm_mean; /* [obs_dim] */
  const float* norm_std;  /* [obs_dim] */
} az_model;

const char* az_last_error(void);
/* Number of GPUs visible / the gfx target the library was built for. */
int az_device_count(void);
const char* az_build_arch(void);

/* mctx_dist/_src/seq_halving.py:73-88 get_table_of_considered_visits: fills the HOST
 * array table[(max_m+1) * S]. */
int az_considered_table(int32_t max_m, int32_t S, int32_t* table_host);

/* mctx_dist/_src/search.py:90 `rng, simulate_key, expand_key = split(rng, 3)` for all S
 * simulations of a search with key `key`: fills the HOST array keys_host[S][2] with every
 * simulation's simulate key.  The chain is sequential in the simulation index, which suits one
 * CPU core far better than one GPU thread; az_search_inputs.sim_keys takes a device copy. */
int az_sim_keys(const uint32_t key[2], int32_t S, uint32_t* keys_host);

/* mctx_dist/_src/search.py:345-387 instantiate_tree_from_root.  root_state [B,4] /
 * root_time [B] may be NULL when the tree carries no native embedding. */
int az_tree_init(const az_search_desc* d, const az_tree* t, const float* root_logits,
                 const float* root_value, const float* root_state, const int32_t* root_time,
                 void* stream);

/* mctx_dist/_src/search.py:88-112 (the whole fori_loop: simulate :127-187, expand
 * :190-244, backward :247-292) as ONE persistent launch, with the recurrent_fn
 * outputs read from `inj`.  Like search() of the reference, which instantiates its tree
 * itself (:345-387), this call starts from a fresh tree: `t` must be as az_tree_init leaves it
 * (some kernels read only its root row and rebuild the rest).  sum_depth (device int64, may be NULL) accumulates the
 * depth of every expanded node, the quantity the bandwidth model is priced on. */
int az_search_injected(const az_search_desc* d, const az_tree* t, const az_search_inputs* in,
                       const az_injected* inj, int64_t* sum_depth, void* stream);

/* mctx_dist/_src/search.py:93-99: simulate for simulation `sim`; writes the parent
 * node, the action to expand and the index of the node to (re)evaluate, each [B]. */
/* AZ_VARIANT_STOCHASTIC (policies.py:676-708) runs on this entry point only: decision nodes
 * select with the muzero rule over their first num_decision_actions columns, chance nodes take
 * argmax(softmax(chance logits) / (visits + 1)) over the rest. */
int az_simulate(const az_search_desc* d, const az_tree* t, const az_search_inputs* in,
                int32_t sim, int32_t* parent, int32_t* action, int32_t* next,
                int64_t* sum_depth, void* stream);

/* mctx_dist/_src/search.py:100-103: expand (tree writes of :231-244, :304-342; the
 * caller scatters its own embeddings) followed by backward (:247-292), given the
 * outputs of the caller's recurrent_fn. */
int az_expand_backward(const az_search_desc* d, const az_tree* t, const int32_t* parent,
                       const int32_t* action, const int32_t* next, const float* prior_logits,
                       const float* value, const float* reward, const float* discount,
                       void* stream);

/* Size in floats of the packed-weight buffer of a model with these networks: the re-laid-out
 * weights and biases followed by a fixed-size block for the networks' work schedule. */
int64_t az_model_packed_floats(const az_mlp_params* policy, const az_mlp_params* value);
/* Re-lays the flax-layout weights out for the matrix cores into `packed` and leaves the networks'
 * work schedule (which wave multiplies which tiles in which order: a function of the layer shapes
 * only) behind them, so that the kernels copy it instead of working it out at every launch.  The
 * buffer may be copied or moved afterwards (the schedule's addresses are adjusted by the distance);
 * a kernel that finds no schedule for its model's shapes there builds its own. */
int az_model_pack(const az_mlp_params* policy, const az_mlp_params* value, float* packed,
                  void* stream);

/* No counterpart in the reference (which computes in float32 throughout): the same buffer with
 * the weights of the hidden layers -- every layer but the first and the last of each network --
 * rounded to bfloat16 (nearest even) and laid out for v_mfma_f32_16x16x32_bf16; their inputs are
 * rounded to bfloat16 as they are read, products are accumulated in float32, biases and
 * activation functions stay float32, and so do the first layer (the observation) and the output
 * layer.  Those layers' input widths must be multiples of 32.  A model packed this way sets
 * az_model.hidden_bf16; its results follow the float32 ones to about three decimal digits and
 * are NOT bit-comparable with the CPU checker. */
int64_t az_model_packed_floats_bf16(const az_mlp_params* policy, const az_mlp_params* value);
int az_model_pack_bf16(const az_mlp_params* policy, const az_mlp_params* value, float* packed,
                       void* stream);

/* alphazero.py:208-216 forward_networks (observation normalisation of
 * running_statistics.py:183-199, policy and value MLPs of networks/networks.py:153-193)
 * on obs [B,obs_dim]; writes logits [B,A] and value [B,K]. */
int az_forward(const az_model* m, int32_t B, const float* obs, float* logits, float* value,
               void* stream);

/* alphazero.py:317-339 recurrent_fn on the native state: environment step
 * (gymnax classic_control step_env through gymnax/wrappers/brax.py:57-70), networks,
 * discount = gamma * where(truncation, 1, 1 - done).  state [B,4] and time [B] are
 * updated in place. */
int az_recurrent_step(const az_model* m, int32_t B, float* state, int32_t* time,
                      const int32_t* action, float* reward, float* discount, float* logits,
                      float* value, void* stream);

/* The whole search of search.py:88-112 with the recurrent_fn of alphazero.py:317-339
 * fused into the persistent launch (environment step + both MLPs on the matrix
 * cores).  The tree must carry emb_state / emb_time. */
int az_search_fused(const az_search_desc* d, const az_tree* t, const az_search_inputs* in,
                    const az_model* m, int64_t* sum_depth, void* stream);

/* mctx_dist/_src/policies.py:80-90 (+ :486-516): root prior logits of muzero_policy
 * from the network logits and a Dirichlet sample [B,A]. */
int az_root_muzero(int32_t B, int32_t A, const float* prior_logits, const float* dirichlet,
                   float dirichlet_fraction, const float* invalid, float* out_logits,
                   void* stream);
/* mctx_dist/_src/policies.py:184-200, 534-546: root priors of sampled_muzero_policy.
 * counts_in [B,A] (injected draw counts) or NULL to draw num_samples categorical
 * samples from `key` (host, 2 words).  exponent = 1 - 1/sampling temperature. */
int az_root_sampled(int32_t B, int32_t A, const float* prior_logits, const float* dirichlet,
                    float dirichlet_fraction, const float* invalid, int32_t num_samples,
                    float temperature, float exponent, const float* counts_in,
                    const uint32_t* key_host, float* out_probs, void* stream);
/* mctx_dist/_src/policies.py:291-297: masked root logits and the scaled Gumbel
 * sample of gumbel_muzero_policy (gumbel_in [B,A] injected, or NULL to draw from
 * `key`). */
int az_root_gumbel(int32_t B, int32_t A, const float* prior_logits, const float* invalid,
                   const float* gumbel_in, const uint32_t* key_host, float gumbel_scale,
                   float* out_logits, float* out_gumbel, void* stream);
/* Dirichlet(alpha) samples [B,A] from `key` (stands in for jax.random.dirichlet at
 * mctx_dist/_src/policies.py:511-514; same distribution, different stream). */
int az_dirichlet(int32_t B, int32_t A, float alpha, const uint32_t* key_host, float* out,
                 void* stream);

/* mctx_dist/_src/policies.py:113-122 / 225-234 with tree.py:102-120: action_weights =
 * root visit probabilities, action = categorical(logits / temperature) drawn as
 * argmax(gumbel + logits) with gumbel_in [B,A] or from `key`. */
int az_epilogue_visits(const az_search_desc* d, const az_tree* t, float temperature,
                       const float* gumbel_in, const uint32_t* key_host, int32_t* action,
                       float* action_weights, void* stream);
/* mctx_dist/_src/policies.py:321-343: action and action_weights of gumbel_muzero_policy. */
int az_epilogue_gumbel(const az_search_desc* d, const az_tree* t, const float* root_logits,
                       const float* root_gumbel, const float* invalid, int32_t* action,
                       float* action_weights, void* stream);
/* mctx_dist/_src/tree.py:102-120 Tree.summary(): visit_counts/visit_probs [B,A],
 * value [B,K], qvalues [B,A,K]. */
int az_summary(const az_search_desc* d, const az_tree* t, float* visit_counts,
               float* visit_probs, float* value, float* qvalues, void* stream);
/* alphazero.py:367-379: greedy action over searched actions (or the sampled one),
 * search value and prior log-probability of the chosen action. */
int az_actor_post(const az_search_desc* d, const az_tree* t, const float* action_weights,
                  const int32_t* sampled_action, int32_t deterministic,
                  const float* prior_logits, int32_t* action, float* search_value,
                  float* prior_log_prob, void* stream);

/* jax.random.split(key, n) in the partitionable counter layout: out [n,2] (used as
 * alphazero.py does to derive one reset key per environment). */
int az_split_keys(const uint32_t* key_host, int32_t n, uint32_t* out, void* stream);
/* jax.random.bits(key, (n,), uint32) in the partitionable counter layout, zero-extended to
 * int64 out[n]: the sort keys of one round of jax.random.permutation (alphazero.py:1079 shuffles
 * the sampled batch with it; the rounds' key splits and stable sorts are the caller's). */
int az_random_sort_keys(const uint32_t* key_host, int32_t n, int64_t* out, void* stream);
/* reset of the wrapped environment (gymnax/wrappers/brax.py:41-55 on top of the
 * reset_env of the classic_control envs): keys [B,2] -> state [B,4], time [B],
 * obs [B,obs_dim], and rng_out [B,2] = split(key)[0], the key the wrapper stores in
 * info['_rng'] (may be NULL). */
int az_env_reset(int32_t env_id, int32_t B, const uint32_t* keys, float* state, int32_t* time,
                 float* obs, uint32_t* rng_out, void* stream);
/* alphazero.py:383 env.step = AutoResetWrapper(VmapWrapper(EpisodeWrapper(
 * GymnaxToBraxWrapper))).step (envs/brax_wrappers.py:81-97, 109-134;
 * gymnax/wrappers/brax.py:57-70).  All arrays are per-lane and updated in place;
 * rng is [B,2]. */
int az_real_env_step(int32_t env_id, int32_t B, int32_t episode_length, float* state,
                     int32_t* time, float* obs, float* real_obs, float* reward, float* done,
                     float* truncation, float* steps, uint32_t* rng, const int32_t* action,
                     void* stream);

/* Device evaluation of the numerics specification (exp, log, sin, cos, swish, sqrt,
 * reciprocal, gumbel-from-bits: fn = 0..7; fn = 8: swish in the packed form the dense
 * layers' epilogue uses; fn = 9, 10: sine and cosine in the select-only form the Acrobot step
 * uses) and of the threefry block, used to check
 * that host and device produce the same bits.  No counterpart in the reference. */
int az_math_apply(int32_t fn, int32_t n, const float* x, float* y, void* stream);
int az_threefry(int32_t n, const uint32_t* key, const uint32_t* ctr, uint32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AZ_MCTS_H_ */
