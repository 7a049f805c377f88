"""The training step of the reference's driver, one process per GPU.

Mirrors alphazero.py: `actor_step` (:287-403), `generate_unroll` (:406-430), `training_step`
(:1092-1183), `prefill_replay_buffer` (:1224-1296) and the evaluator's episode loop
(:433-508) for the built-in environments.  Self-play (search, environment, networks in
the tree) runs in the HIP kernels; the learner side is `learner.py`; replay is
`replay_buffers.py`.  'Breakout-MinAtar' (is_atari_env, alphazero.py:878) takes the reference's
other branch: the shared convolutional torso (alphazero.py:246-267, 1304-1308) evaluated by
library calls, the search on the per-simulation kernels with the environment's HIP step as the
model, flattened [400] observations in the replay rows.  Data parallelism follows alphazero.py:893-896, 935-943: every rank
owns `num_envs // world` lanes, its share of the replay rows and of the sampled batch;
the only collectives are the gradient mean per minibatch and the observation-statistics
sums per training step (torch.distributed: RCCL on GPUs).
"""
import dataclasses
import time

import torch

from . import envs as envs_lib
from . import learner as L
from . import networks, parallel, policies, prng, replay_buffers
from .base import RootFnOutput
from .qtransforms import qtransform_by_parent_and_siblings, qtransform_completed_by_mix_value


@dataclasses.dataclass
class Config:
    """The hyper-parameters of alphazero.py:37-100 that the training step reads, with the
    reference's values as defaults (tests and benchmarks pass their own, smaller ones).
    `policy` and `num_samples` have no Config entry in the reference: its actor_step calls
    sampled_muzero_policy (alphazero.py:354) without num_samples, i.e. with the policy's
    default of 100 (mctx_dist/_src/policies.py:131)."""
    env_id: str = "Acrobot-v1"
    seed: int = 20
    num_envs: int = 16
    episode_length: int = 500
    unroll_length: int = 128
    num_simulations: int = 30
    policy: str = "sampled"            # "sampled" (alphazero.py:354), "muzero", "gumbel"
    num_samples: int = 100
    max_num_considered_actions: int = 16
    policy_hidden: tuple = (32, 32, 32, 32)
    value_hidden: tuple = (256, 256, 256, 256, 256)
    atari_dense_layer_sizes: tuple = (256,)     # alphazero.py:100 (Breakout-MinAtar only)
    num_atoms: int = 1
    qr_kappa: float = 1.0
    use_gae: bool = True
    gae_lambda: float = 0.95
    n_step_n: int = 5
    n_step_gamma: float = 0.99
    max_replay_size: int = 8192
    min_replay_size: int = 8192
    replay_buffer_batch_size: int = 128
    num_minibatches: int = 128
    update_epochs: int = 1
    per_alpha: float = 0.0
    per_importance_sampling: bool = True
    per_importance_sampling_beta: float = 1.0
    learning_rate: float = 1e-3
    anneal_lr: bool = True
    total_timesteps: int = 1_000_000
    max_grad_norm: float = 0.5
    vf_cost: float = 0.5
    l2_coef: float = 1e-4
    normalize_observations: bool = True
    num_eval_envs: int = 64            # alphazero.py:56
    deterministic_eval: bool = True
    # not in the reference: self-play evaluates the networks' hidden layers on the bfloat16 matrix
    # instructions (networks.FusedModel(hidden_bf16=True)); the learner stays float32
    selfplay_hidden_bf16: bool = False
    # not in the reference: with one process the minibatch update (loss, gradient, clip + Adam)
    # is captured once in a HIP graph and replayed (learner.GraphedMinibatchStep)
    graph_sgd: bool = True


class Trainer:
    def __init__(self, cfg, device=None):
        self.cfg = cfg
        self.rank, self.world, local = parallel.trainer_ranks()
        self.device = torch.device("cuda", local) if device is None else torch.device(device)
        if self.device.type == "cuda":
            # the native launches go to the CURRENT device's stream: make it this rank's
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            torch.cuda.set_device(self.device)
        shard = parallel.shard_lanes(cfg.num_envs, self.rank, self.world, key=cfg.seed)
        self.lanes = shard.lanes
        self.key = shard.key
        self.env = envs_lib.make(cfg.env_id, episode_length=cfg.episode_length)
        self.atari = cfg.env_id == "Breakout-MinAtar"
        # identical initial parameters on every rank (alphazero.py:1305-1326 replicates them)
        g = torch.Generator(device=self.device).manual_seed(cfg.seed)
        if self.atari:
            from . import minatar
            obs, act = minatar.OBS, minatar.ACTIONS
            self.stats = L.RunningStatistics(obs, device=self.device)
            # the learner's functions take the networks object in the policy slot (learner.forward_networks)
            self.policy_layers = minatar.AtariNetworks(act, cfg.num_atoms, tuple(cfg.atari_dense_layer_sizes),
                                                       generator=g, device=self.device)
            self.value_layers = None
            self.model = None
            self.recurrent_fn = minatar.recurrent_fn(self.env, self.policy_layers, cfg.n_step_gamma,
                                                     normalizer=self._normalizer)
        else:
            obs, act = envs_lib.observation_size(cfg.env_id), envs_lib.action_size(cfg.env_id)
            self.policy_layers = networks.init_mlp((obs,) + tuple(cfg.policy_hidden) + (act,), g, self.device)
            self.value_layers = networks.init_mlp((obs,) + tuple(cfg.value_hidden) + (cfg.num_atoms,), g,
                                                  self.device)
            self.stats = L.RunningStatistics(obs, device=self.device)
            self.model = networks.FusedModel(cfg.env_id, self.policy_layers, self.value_layers,
                                             self.stats.mean, self.stats.std, gamma=cfg.n_step_gamma,
                                             hidden_bf16=cfg.selfplay_hidden_bf16)
            self.recurrent_fn = networks.FusedRecurrentFn(self.model)
        self.optimizer = L.ClipAdam(L.flat_leaves(self.policy_layers, self.value_layers),
                                    cfg.learning_rate, cfg.max_grad_norm, cfg.anneal_lr,
                                    cfg.total_timesteps)
        self.layout = L.row_layout(obs, act, cfg.num_atoms)
        sample_bs = cfg.replay_buffer_batch_size * cfg.num_minibatches // self.world
        cap = cfg.max_replay_size // self.world
        if cfg.per_alpha > -1:                                        # alphazero.py:933-943
            self.replay = replay_buffers.PrioritizedSamplingQueue(
                cap, self.layout["_width"], sample_bs, cfg.per_importance_sampling_beta)
        else:
            self.replay = replay_buffers.UniformSamplingQueue(cap, self.layout["_width"], sample_bs)
        self.key, kb, ke = prng.split(self.key, 3)
        self.buffer_state = self.replay.init(kb, device=self.device)
        self.env_state = self.env.reset(ke, batch_size=self.lanes, device=self.device)
        self.env_steps = 0
        self.trace = None   # a list here records every actor step's raw inputs and outputs
        self._perm_gen = torch.Generator(device=self.device).manual_seed(cfg.seed + 1)
        self._graphed = None

    # -------------------------------------------------------------- self-play
    def _normalizer(self):
        if self.cfg.normalize_observations:
            return self.stats.mean, self.stats.std
        return None, None

    def _refresh_model(self):
        if self.model is not None:   # (the torso's networks are read in place)
            self.model.update(self.policy_layers, self.value_layers, *self._normalizer())

    def _forward(self, obs):
        """alphazero.py:208-216 without gradients: (prior logits, value) of the current networks."""
        if self.model is not None:
            return self.model.forward(obs)
        with torch.no_grad():
            return self.policy_layers.forward(obs, *self._normalizer())

    def actor_step(self, key, deterministic=False, state=None):
        """alphazero.py:287-403: search on the states `state` (default: the training
        environments' current ones), act, step the environment."""
        cfg, st = self.cfg, self.env_state if state is None else state
        _, _, key = prng.split(key, 3)   # key, logits_rng, search_rng (:298); the search takes the third
        logits, value = self._forward(st.obs)
        root = RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
        if cfg.policy == "gumbel":
            out = policies.gumbel_muzero_policy(
                (), key, root, self.recurrent_fn, cfg.num_simulations,
                max_num_considered_actions=cfg.max_num_considered_actions,
                qtransform=qtransform_completed_by_mix_value)
        elif cfg.policy == "muzero":
            out = policies.muzero_policy((), key, root, self.recurrent_fn, cfg.num_simulations,
                                         qtransform=qtransform_by_parent_and_siblings)
        else:
            out = policies.sampled_muzero_policy((), key, root, self.recurrent_fn,
                                                 cfg.num_simulations, num_samples=cfg.num_samples)
        action, search_value, log_prob = policies.actor_post(
            out.search_tree, out.action_weights, out.action, logits, deterministic=deterministic)
        def rows_of(x):   # the replay rows hold flattened observations (a no-op for vectors)
            return x.reshape(x.shape[0], -1)

        obs, real_obs = rows_of(st.obs).clone(), rows_of(st.real_obs).clone()
        nst = self.env.step(st, action)
        if self.trace is not None:
            self.trace.append(dict(obs=obs, real_obs=real_obs, value=value, action=action,
                                   action_weights=out.action_weights, search_value=search_value,
                                   prior_log_prob=log_prob, reward=nst.reward, done=nst.done,
                                   truncation=nst.truncation, next_real_obs=rows_of(nst.real_obs),
                                   steps=nst.steps))
        ones = torch.ones_like(nst.reward)
        tr = dict(observation=obs, real_obs=real_obs, action=action.float(), reward=nst.reward,
                  discount=1 - nst.done, next_observation=rows_of(nst.real_obs),
                  target_policy_probs=out.action_weights, search_value=search_value,
                  value_prefix_target=torch.zeros_like(nst.reward),
                  bootstrap_observation=torch.zeros_like(obs),
                  bootstrap_value=value,            # overloaded with the prior value (:396)
                  bootstrap_discount=ones, prior_log_prob=log_prob, raw_action=action.float(),
                  truncation=nst.truncation, priority=ones, weight=ones)
        return nst, tr

    def generate_unroll(self, key, length):
        """alphazero.py:406-430: dict of [T, B, ...] tensors."""
        steps = []
        for _ in range(length):
            sub, key = prng.split(key, 2)    # current_key, next_key (:422)
            self.env_state, tr = self.actor_step(sub)
            steps.append(tr)
        return {k: torch.stack([s[k] for s in steps], 0) for k in steps[0]}

    def _targets(self, data, final_search_value):
        cfg = self.cfg
        trunc = data["truncation"]
        kw = dict(rewards=data["reward"], discounts=data["discount"] * (1 - trunc),
                  termination_discount=data["discount"], observations=data["next_observation"],
                  values=torch.cat([data["search_value"], final_search_value.unsqueeze(0)], 0))
        if cfg.use_gae:
            return L.compute_gae(**kw, lambda_=cfg.gae_lambda, discount=cfg.n_step_gamma)
        return L.n_step_bootstrapped_targets(**kw, n=cfg.n_step_n, gamma=cfg.n_step_gamma)

    def collect(self, k_unroll, k_boot, length=None):
        """alphazero.py:1104-1151: unroll of `length` actor steps (default Config.unroll_length),
        bootstrap search, targets, priorities, insert."""
        cfg = self.cfg
        length = cfg.unroll_length if length is None else int(length)
        data = self.generate_unroll(k_unroll, length)
        saved = self.env_state
        _, boot = self.actor_step(k_boot)       # the extra search only provides the bootstrap value
        self.env_state = saved
        prefix, b_obs, b_val, b_disc = self._targets(data, boot["search_value"])
        targets = prefix.unsqueeze(-1) + b_disc.unsqueeze(-1) * b_val
        data["priority"] = L.priorities(targets, data["bootstrap_value"], cfg.per_alpha)
        data.update(value_prefix_target=prefix, bootstrap_observation=b_obs, bootstrap_value=b_val,
                    bootstrap_discount=b_disc)
        flat = {k: v.reshape((-1,) + tuple(v.shape[2:])) for k, v in data.items()}
        rows = L.pack_rows(self.layout, flat)
        self.replay.insert(self.buffer_state, rows)
        self.env_steps += length * self.lanes * self.world
        return flat

    # -------------------------------------------------------------- training
    def training_step(self):
        """alphazero.py:1092-1183."""
        cfg = self.cfg
        # key_sgd, key_generate_unroll, key_bootstrap, reanalyze_key, new_key (:1097)
        k_sgd, k_unroll, k_boot, _, self.key = prng.split(self.key, 5)
        flat = self.collect(k_unroll, k_boot)
        return self.learn(flat, k_sgd)

    def learn(self, flat, key_sgd=None):
        """alphazero.py:1153-1183: statistics, sample, SGD, priority refresh.  key_sgd: the carry
        key of the SGD steps (:1167), which orders the minibatches (jax.random.permutation under
        each step's key_perm, :1076-1079); without it the order comes from a torch generator."""
        cfg = self.cfg
        if cfg.normalize_observations:
            self.stats.update(flat["observation"])
        mean, std = self.stats.mean, self.stats.std
        self.buffer_state, rows = self.replay.sample(self.buffer_state)
        batch = L.unpack_rows(self.layout, rows)
        metrics = []
        loss_kw = dict(qr_kappa=cfg.qr_kappa, vf_cost=cfg.vf_cost, l2_coef=cfg.l2_coef,
                       per_importance_sampling=cfg.per_importance_sampling)
        # (the torso's convolutions stay outside graph capture: eager minibatches for Breakout)
        if cfg.graph_sgd and self.world == 1 and self.device.type == "cuda" and self._graphed is None \
                and not self.atari:
            self._graphed = L.GraphedMinibatchStep(
                self.policy_layers, self.value_layers, self.optimizer, self.layout,
                rows.shape[0] // cfg.num_minibatches, cfg.num_atoms, self.device, **loss_kw)
        for _ in range(cfg.update_epochs):
            metrics += L.sgd_step(self.policy_layers, self.value_layers, mean, std, batch,
                                  self.optimizer, cfg.num_atoms, cfg.num_minibatches,
                                  generator=self._perm_gen, rows=rows, layout=self.layout,
                                  graphed=self._graphed, key=key_sgd,
                                  qr_kappa=cfg.qr_kappa, vf_cost=cfg.vf_cost,
                                  l2_coef=cfg.l2_coef,
                                  per_importance_sampling=cfg.per_importance_sampling)
            if key_sgd is not None:
                key_sgd = prng.split(key_sgd, 3)[0]   # the carry of sgd_step (:1076, :1089)
        # refresh the priorities of the sampled rows with the PRE-update networks (:1177-1180)
        targets = batch["value_prefix_target"].unsqueeze(-1) + \
            batch["bootstrap_discount"].unsqueeze(-1) * batch["bootstrap_value"]
        _, values = self._forward(batch["observation"].contiguous())
        self.replay.set_priorities(self.buffer_state, L.priorities(targets, values, cfg.per_alpha))
        self._refresh_model()
        out = {k: torch.stack([m[k] for m in metrics]).mean() for k in metrics[0]}
        out["buffer_current_size"] = self.replay.size(self.buffer_state)
        return out

    def prefill(self):
        """alphazero.py:1224-1296 as the driver calls it (:845, :1427-1431): ONE unroll of
        ceil(min_replay_size / num_envs) actor steps with one 3-way key split and one bootstrap
        search, inserted without learning; then the observation statistics take the prefilled
        observations (summed over the ranks) and self-play switches to them.  The driver splits
        its key in two and drops the key the prefill hands back (:1428-1431).  Returns the
        number of actor steps."""
        cfg = self.cfg
        steps = -(-cfg.min_replay_size // (self.lanes * self.world))     # np.ceil (:845)
        if steps <= 0:
            return 0
        prefill_key, self.key = prng.split(self.key, 2)
        k_unroll, k_boot, _ = prng.split(prefill_key, 3)                 # :1229
        flat = self.collect(k_unroll, k_boot, length=steps)
        if cfg.normalize_observations:
            self.stats.update(flat["observation"])
            self._refresh_model()
        return steps

    def evaluate(self, key=None, num_eval_envs=None, episode_length=None):
        """alphazero.py:433-508 (Evaluator.run_evaluation) with the accumulation of EvalWrapper
        (envs/brax_wrappers.py:153-197): `num_eval_envs` fresh environments run one unroll of
        `episode_length` actor steps (greedy actions when Config.deterministic_eval); every lane
        adds up the rewards of its FIRST episode and records that episode's length.  Runs on
        this rank's device only, like the reference's evaluator (process 0, one device)."""
        cfg = self.cfg
        n = cfg.num_eval_envs if num_eval_envs is None else int(num_eval_envs)
        length = cfg.episode_length if episode_length is None else int(episode_length)
        if key is None:
            self.key, key = prng.split(self.key, 2)
        t0 = time.perf_counter()
        st = self.env.reset(key, batch_size=n, device=self.device)      # split(key, num_eval_envs) (:460)
        active = torch.ones_like(st.reward)
        episode_reward = torch.zeros_like(st.reward)
        episode_steps = torch.zeros_like(st.reward)
        for _ in range(length):
            sub, key = prng.split(key, 2)                                # generate_unroll (:422)
            st, _ = self.actor_step(sub, deterministic=cfg.deterministic_eval, state=st)
            episode_steps = torch.where(active != 0, st.steps, episode_steps)
            episode_reward = episode_reward + st.reward * active
            active = active * (1 - st.done)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - t0
        return {"eval/episode_reward": float(episode_reward.mean()),
                "eval/episode_reward_std": float(episode_reward.std(unbiased=False)),   # np.std
                "eval/avg_episode_length": float(episode_steps.mean()),
                "eval/sps": length * n / dt,                             # :473, :500
                "episode_reward": episode_reward, "episode_steps": episode_steps,
                "active_episodes": active}

    def train(self, num_training_steps, log=None):
        """The driver's loop (alphazero.py:1425-1470) without its logging: prefill once while the
        replay queue holds fewer than min_replay_size rows over all ranks, then training steps."""
        if self.cfg.min_replay_size > 0 and \
                int(parallel.sum_over_ranks(int(self.replay.size(self.buffer_state)), device=self.device)) \
                < self.cfg.min_replay_size:
            self.prefill()
        t0 = time.perf_counter()
        steps0 = self.env_steps
        m = {}
        for _ in range(num_training_steps):
            m = self.training_step()
        torch.cuda.synchronize(self.device)
        dt = parallel.max_over_ranks(time.perf_counter() - t0, device=self.device)
        m = {k: float(v) for k, v in m.items()}
        m["training/sps"] = (self.env_steps - steps0) / dt          # alphazero.py:1210-1212
        if log:
            log(m)
        return m
