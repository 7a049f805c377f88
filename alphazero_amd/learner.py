"""Learner side of the training step in PyTorch (ROCm): targets, losses, optimiser.

Mirrors alphazero.py: `n_step_bootstrapped_targets` (:544-602), `compute_gae` (:605-663),
`quantile_regression_loss` (:667-706), `mse_value_loss` (:709-720), `compute_muzero_loss`
(:723-807), the priority formula (:1135, :1179), the optimiser
`optax.chain(clip_by_global_norm, adam(linear_schedule))` (:968-979), the minibatch / SGD
loops (:1060-1089) and `running_statistics.update` (running_statistics.py:75-180).
Tensors keep the reference's shapes ([T, B, ...] for sequences, [N, ...] for batches) and
live on whatever device the inputs live on; gradients of several ranks are averaged with
one all-reduce of the flat gradient per minibatch (the reference's `pmean`, :1021).
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import prng


# ---------------------------------------------------------------- targets
def n_step_bootstrapped_targets(rewards, discounts, termination_discount, observations, values,
                                n=5, gamma=1.0):
    """alphazero.py:544-602.  rewards/discounts/termination_discount [T,B], observations
    [T,B,...], values [T+1,B,K] -> (value_prefix_targets [T,B], bootstrap_observations,
    bootstrap_values [T,B,K], bootstrap_discounts [T,B])."""
    values = values[1:]
    T = rewards.shape[0]
    pad = min(n - 1, T)

    def shift(x):
        tail = x[-1:].expand((pad,) + tuple(x.shape[1:]))
        return torch.cat([x[n - 1:], tail], 0)

    boot_obs, boot_val = shift(observations), shift(values)
    tail = (n - 1,) + tuple(rewards.shape[1:])
    rewards_p = torch.cat([rewards, rewards.new_zeros(tail)], 0)
    discounts_p = torch.cat([discounts, discounts.new_ones(tail)], 0) * gamma
    prefix = rewards_p[n - 1:n - 1 + T]
    boot_disc = (torch.cat([termination_discount, discounts.new_ones(tail)], 0) * gamma)[n - 1:n - 1 + T]
    for i in range(n - 2, -1, -1):
        d_ = discounts_p[i:i + T]
        prefix = rewards_p[i:i + T] + d_ * prefix
        boot_disc = boot_disc * d_
    return prefix, boot_obs, boot_val, boot_disc


def compute_gae(rewards, discounts, termination_discount, observations, values, lambda_=1.0,
                discount=0.99):
    """alphazero.py:605-663: returns (0, 0, vs [T,B,K], 1) in the n-step function's slots."""
    termination = (1 - termination_discount).unsqueeze(-1)
    truncation = (1 - discounts).unsqueeze(-1) * (1 - termination)
    mask = 1 - truncation
    deltas = (rewards.unsqueeze(-1) + discount * (1 - termination) * values[1:] - values[:-1]) * mask
    acc = torch.zeros_like(values[0])
    out = []
    for t in range(rewards.shape[0] - 1, -1, -1):
        acc = deltas[t] + discount * (1 - termination[t]) * mask[t] * lambda_ * acc
        out.append(acc)
    vs = torch.stack(out[::-1], 0) + values[:-1]
    return torch.zeros_like(rewards), torch.zeros_like(observations), vs, torch.ones_like(discounts)


def priorities(targets, values, per_alpha):
    """alphazero.py:1135 / :1179."""
    return ((targets - values).abs().mean(-1) + 1e-10) ** per_alpha


# ---------------------------------------------------------------- losses
def quantile_regression_loss(values, targets, kappa, num_atoms):
    """alphazero.py:667-706: values, targets [N,K] -> [N]."""
    err = targets.unsqueeze(-2) - values.unsqueeze(-1)
    a = err.abs()
    huber = (a <= kappa).to(err.dtype) * 0.5 * err ** 2 + (a > kappa).to(err.dtype) * kappa * (a - 0.5 * kappa)
    tau = ((torch.arange(num_atoms, dtype=values.dtype, device=values.device) + 0.5) / num_atoms)
    w = (tau.view(1, -1, 1) - (err < 0).to(err.dtype)).abs()
    return (w * huber).mean(2).sum(1)


def mse_value_loss(values, targets):
    """alphazero.py:709-720."""
    e = targets.detach() - values
    return (e * e * 0.5).squeeze()


def mlp_apply(layers, x):
    """networks/networks.py:25-48: Dense + swish on all but the last layer."""
    for i, (w, b) in enumerate(layers):
        x = x @ w + b
        if i + 1 < len(layers):
            x = x * torch.sigmoid(x)
    return x


def forward_networks(policy_layers, value_layers, mean, std, obs):
    """alphazero.py:208-216 (differentiable twin of FusedModel.forward).  With value_layers None,
    `policy_layers` is a networks object with forward(obs, mean, std) and leaves() of its own
    (the shared convolutional torso with its two heads, alphazero.py:246-267): the normaliser is
    applied once, in front of the torso (networks/networks.py:145-147)."""
    if value_layers is None:
        return policy_layers.forward(obs, mean, std)
    x = (obs - mean) / std
    return mlp_apply(policy_layers, x), mlp_apply(value_layers, x)


# columns of a flat MCTSTransition row (envs/transition.py:18-34 in ravel order; extras is
# {'policy_extras': {'prior_log_prob', 'raw_action'}, 'state_extras': {'truncation'}})
def row_layout(obs_dim, num_actions, num_atoms):
    fields = [("observation", obs_dim), ("real_obs", obs_dim), ("action", 1), ("reward", 1),
              ("discount", 1), ("next_observation", obs_dim), ("target_policy_probs", num_actions),
              ("search_value", num_atoms), ("value_prefix_target", 1),
              ("bootstrap_observation", obs_dim), ("bootstrap_value", num_atoms),
              ("bootstrap_discount", 1), ("prior_log_prob", 1), ("raw_action", 1),
              ("truncation", 1), ("priority", 1), ("weight", 1)]
    out, off = {}, 0
    for name, w in fields:
        out[name] = (off, w)
        off += w
    out["_width"] = off
    return out


def pack_rows(layout, fields):
    """dict of [N, w] / [N] tensors -> flat rows [N, D]."""
    cols = []
    for name, spec in layout.items():
        if name == "_width":
            continue
        w = spec[1]
        x = fields[name].to(torch.float32)
        cols.append(x.reshape(x.shape[0], w))
    return torch.cat(cols, 1).contiguous()


def unpack_rows(layout, rows):
    out = {}
    for name, spec in layout.items():
        if name == "_width":
            continue
        off, w = spec
        x = rows[:, off:off + w]
        out[name] = x[:, 0] if w == 1 and name not in ("search_value", "bootstrap_value",
                                                      "target_policy_probs") else x
    return out


def compute_muzero_loss(policy_layers, value_layers, mean, std, data, num_atoms, qr_kappa=1.0,
                        vf_cost=0.5, l2_coef=1e-4, per_importance_sampling=True, l2_outside_graph=False):
    """alphazero.py:723-807 on a dict of batch fields; returns (total_loss, metrics).
    l2_outside_graph: the L2 penalty enters the reported losses only, computed without autograd
    in three launches; the caller adds its gradient, l2_coef * parameter, itself
    (minibatch_step) -- the differentiable form costs about a hundred launches of a few
    microseconds per minibatch for the 22 parameter leaves."""
    logits, baseline = forward_networks(policy_layers, value_layers, mean, std, data["observation"])
    lsm = F.log_softmax(logits, -1)
    policy_loss = -(data["target_policy_probs"].detach() * lsm).sum(-1).mean()
    vs = (data["value_prefix_target"].unsqueeze(-1) +
          data["bootstrap_discount"].unsqueeze(-1) * data["bootstrap_value"]).detach()
    if num_atoms > 1:
        v_losses = quantile_regression_loss(baseline, vs, qr_kappa, num_atoms)
    else:
        v_losses = mse_value_loss(baseline, vs)
    if per_importance_sampling:
        v_losses = v_losses * data["weight"]
    v_loss = vf_cost * v_losses.mean()
    if l2_outside_graph:
        with torch.no_grad():
            norms = torch.stack(torch._foreach_norm(flat_leaves(policy_layers, value_layers)))
            l2 = l2_coef * 0.5 * (norms * norms).sum()
    else:
        l2 = l2_coef * 0.5 * sum((p ** 2).sum() for p in flat_leaves(policy_layers, value_layers))
    total = policy_loss + v_loss + l2
    with torch.no_grad():
        a = data["raw_action"].long()
        log_ratio = lsm.gather(1, a.unsqueeze(1))[:, 0] - data["prior_log_prob"]
        approx_kl = ((log_ratio.exp() - 1) - log_ratio).mean()
        entropy = -(lsm.exp() * lsm).sum(-1).mean()
    return total, dict(total_loss=total.detach(), policy_loss=policy_loss.detach(),
                       value_loss=v_loss.detach(), l2_penalty=l2.detach(), entropy=entropy,
                       approx_kl=approx_kl)


# ---------------------------------------------------------------- optimiser
class ClipAdam:
    """optax.chain(clip_by_global_norm(max_norm), adam(lr)) with optax's linear_schedule
    (alphazero.py:968-979).  `params` is a flat list of leaf tensors updated in place."""

    def __init__(self, params, learning_rate=1e-3, max_grad_norm=0.5, anneal_lr=True,
                 total_timesteps=1, end_fraction=0.01, b1=0.9, b2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr0, self.max_norm, self.anneal = learning_rate, max_grad_norm, anneal_lr
        self.steps, self.end = total_timesteps, learning_rate * end_fraction
        self.b1, self.b2, self.eps = b1, b2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.count = 0

    def learning_rate(self):
        if not self.anneal:
            return self.lr0
        frac = 1 - min(self.count, self.steps) / self.steps
        return (self.lr0 - self.end) * frac + self.end

    def device_scalars(self, count):
        """(learning rate, 1 - b1^t, 1 - b2^t) as float32 device scalars from `count`, a float64
        device scalar holding the number of updates so far: the same double-precision expressions
        as learning_rate() / step(), evaluated on the device so that a captured update needs
        nothing from the host."""
        t = count + 1
        c1, c2 = 1 - self.b1 ** t, 1 - self.b2 ** t
        if self.anneal:
            frac = 1 - torch.clamp(count, max=float(self.steps)) / self.steps
            lr = (self.lr0 - self.end) * frac + self.end
        else:
            lr = torch.full_like(count, self.lr0)
        return lr.float(), c1.float(), c2.float()

    @torch.no_grad()
    def step(self, grads, scalars=None):
        """`grads`: one tensor per parameter leaf (views of one flat buffer are fine).  The update
        runs as multi-tensor operations -- a dozen launches for all leaves instead of eight
        per leaf; element by element the arithmetic is that of optax's clip + adam.
        scalars: device_scalars() of the update count (captured updates); the host's otherwise."""
        grads = list(grads)
        gnorm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))  # optax.global_norm
        scale = torch.where(gnorm < self.max_norm, torch.ones_like(gnorm), self.max_norm / gnorm)
        if scalars is None:
            lr = self.learning_rate()
            t = self.count + 1
            c1, c2 = 1 - self.b1 ** t, 1 - self.b2 ** t
            self.count += 1
        else:
            lr, c1, c2 = scalars
        g = torch._foreach_mul(grads, scale)
        torch._foreach_mul_(self.m, self.b1)
        torch._foreach_add_(self.m, g, alpha=1 - self.b1)
        torch._foreach_mul_(self.v, self.b2)
        torch._foreach_addcmul_(self.v, g, g, value=1 - self.b2)
        upd = torch._foreach_div(self.m, c1)
        torch._foreach_mul_(upd, lr)
        den = torch._foreach_div(self.v, c2)
        torch._foreach_sqrt_(den)
        torch._foreach_add_(den, self.eps)
        torch._foreach_div_(upd, den)
        torch._foreach_sub_(self.params, upd)
        return gnorm


def flat_leaves(policy_layers, value_layers):
    if value_layers is None:    # a networks object (see forward_networks)
        return list(policy_layers.leaves())
    return [t for w, b in list(policy_layers) + list(value_layers) for t in (w, b)]


def allreduce_mean_(tensors):
    """The reference's `jax.lax.pmean(grad)` (alphazero.py:1021): one all-reduce of the
    flattened gradient; identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    return tensors


def minibatch_step(policy_layers, value_layers, mean, std, data, optimizer, num_atoms, **loss_kw):
    """alphazero.py:1060-1070: loss, gradient (averaged over ranks), optimiser update."""
    leaves = flat_leaves(policy_layers, value_layers)
    for p in leaves:
        p.requires_grad_(True)
    total, metrics = compute_muzero_loss(policy_layers, value_layers, mean, std, data, num_atoms,
                                         l2_outside_graph=True, **loss_kw)
    grads = list(torch.autograd.grad(total, leaves))
    for p in leaves:
        p.requires_grad_(False)
    # d/dp of l2_coef * 0.5 * p^2 (the same two-term sum per element as autograd's accumulation)
    torch._foreach_add_(grads, leaves, alpha=loss_kw.get("l2_coef", 1e-4))
    allreduce_mean_(grads)
    optimizer.step(grads)
    return metrics


class GraphedMinibatchStep:
    """minibatch_step for one process (no gradient all-reduce), captured once in a HIP graph and
    replayed: the step is some three hundred launches of a few microseconds each, and replaying
    them costs the host one call.  The minibatch's packed rows and the normaliser are copied into
    fixed buffers before a replay; the optimiser's update count lives on the device (ClipAdam.
    device_scalars), so nothing is read from the host inside the graph."""

    METRICS = ("total_loss", "policy_loss", "value_loss", "l2_penalty", "entropy", "approx_kl")

    def __init__(self, policy_layers, value_layers, optimizer, layout, n_rows, num_atoms, device, **loss_kw):
        self.policy_layers, self.value_layers, self.optimizer = policy_layers, value_layers, optimizer
        self.layout, self.num_atoms, self.loss_kw = layout, num_atoms, loss_kw
        obs_dim = layout["observation"][1]
        self.rows = torch.zeros((n_rows, layout["_width"]), device=device)
        self.mean = torch.zeros(obs_dim, device=device)
        self.std = torch.ones(obs_dim, device=device)
        self.count = torch.full((), float(optimizer.count), dtype=torch.float64, device=device)
        self.out = None
        self.graph = None

    def _body(self):
        leaves = flat_leaves(self.policy_layers, self.value_layers)
        for p in leaves:
            p.requires_grad_(True)
        total, metrics = compute_muzero_loss(self.policy_layers, self.value_layers, self.mean, self.std,
                                             unpack_rows(self.layout, self.rows), self.num_atoms,
                                             l2_outside_graph=True, **self.loss_kw)
        grads = list(torch.autograd.grad(total, leaves))
        for p in leaves:
            p.requires_grad_(False)
        with torch.no_grad():
            torch._foreach_add_(grads, leaves, alpha=self.loss_kw.get("l2_coef", 1e-4))
            self.optimizer.step(grads, scalars=self.optimizer.device_scalars(self.count))
            self.count += 1
            return torch.stack([metrics[k] for k in self.METRICS])

    def _capture(self):
        # (the state a warm-up run would change is put back before the capture)
        saved = [t.clone() for t in self.optimizer.params + self.optimizer.m + self.optimizer.v] + [self.count.clone()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        for t, s in zip(self.optimizer.params + self.optimizer.m + self.optimizer.v + [self.count], saved):
            t.copy_(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()
        for t, s in zip(self.optimizer.params + self.optimizer.m + self.optimizer.v + [self.count], saved):
            t.copy_(s)

    def __call__(self, rows, mean, std):
        self.rows.copy_(rows)
        self.mean.copy_(mean)
        self.std.copy_(std)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        self.optimizer.count += 1
        out = self.out.clone()
        return {k: out[i] for i, k in enumerate(self.METRICS)}


def permutation(key, n, device):
    """prng.permutation(key, n) as a tensor on `device`.  On a GPU the rounds' sort keys come from
    az_random_sort_keys and the stable sorts run there too: nothing crosses the host."""
    device = torch.device(device)
    if device.type != "cuda":
        return torch.from_numpy(prng.permutation(key, n)).to(device)
    import ctypes as C
    import math
    from . import _native as nat
    x = torch.arange(n, device=device)
    bits = torch.empty((n,), dtype=torch.int64, device=device)
    key = prng.as_key(key)
    for _ in range(int(math.ceil(3 * math.log(max(1, n)) / math.log(2 ** 32 - 1)))):
        key, subkey = prng.split(key, 2)
        nat.check(nat.lib().az_random_sort_keys((C.c_uint32 * 2)(*subkey), n, nat.ptr(bits), nat.stream_ptr()))
        x = x[torch.sort(bits, stable=True).indices]
    return x


def sgd_step(policy_layers, value_layers, mean, std, data, optimizer, num_atoms, num_minibatches,
             generator=None, rows=None, layout=None, graphed=None, key=None, **loss_kw):
    """alphazero.py:1074-1089: one permutation of the sampled batch, split into
    `num_minibatches`, one optimiser update per minibatch.
    key (optional): this step's carry key; the permutation is then jax.random.permutation's under
    key_perm (prng.permutation), otherwise it comes from torch.randperm(generator).
    rows / layout (optional): the packed rows `data` was unpacked from (unpack_rows); they are
    permuted once and the minibatches' fields are views of the result, instead of one gather
    per field and minibatch.  graphed (optional, needs rows): a GraphedMinibatchStep that runs the
    minibatches in place of minibatch_step."""
    n = next(iter(data.values())).shape[0]
    dev = next(iter(data.values())).device
    if key is not None:
        # key, key_perm, key_grad = split(key, 3) (:1076); key_grad reaches a loss that draws nothing
        _, key_perm, _ = prng.split(key, 3)
        perm = permutation(key_perm, n, dev)
    else:
        perm = torch.randperm(n, generator=generator, device=dev)
    metrics = []
    if rows is not None:
        shuffled = rows[perm].reshape(num_minibatches, -1, rows.shape[1])
    for i, chunk in enumerate(perm.reshape(num_minibatches, -1)):
        if graphed is not None:
            metrics.append(graphed(shuffled[i], mean, std))
            continue
        mb = unpack_rows(layout, shuffled[i]) if rows is not None else {k: v[chunk] for k, v in data.items()}
        metrics.append(minibatch_step(policy_layers, value_layers, mean, std, mb, optimizer,
                                      num_atoms, **loss_kw))
    return metrics


# ---------------------------------------------------------------- observation statistics
class RunningStatistics:
    """running_statistics.py:31-57, 75-180 for a flat observation vector.  Over several
    ranks the mean update and the variance update are summed (the reference's two `psum`s;
    the second depends on the first, so they stay two small all-reduces)."""

    def __init__(self, obs_dim, device="cpu", dtype=torch.float32):
        self.count = 0
        self.mean = torch.zeros(obs_dim, device=device, dtype=dtype)
        self.summed_variance = torch.zeros(obs_dim, device=device, dtype=dtype)
        self.std = torch.ones(obs_dim, device=device, dtype=dtype)

    @torch.no_grad()
    def update(self, batch, std_min_value=1e-6, std_max_value=1e6):
        batch = batch.reshape(-1, batch.shape[-1]).to(self.mean.dtype)
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        count = self.count + batch.shape[0] * world   # psum of equal per-rank increments
        diff_old = batch - self.mean
        mean_update = diff_old.sum(0) / count
        if world > 1:
            dist.all_reduce(mean_update, op=dist.ReduceOp.SUM)
        mean = self.mean + mean_update
        var_update = (diff_old * (batch - mean)).sum(0)
        if world > 1:
            dist.all_reduce(var_update, op=dist.ReduceOp.SUM)
        self.summed_variance = self.summed_variance + var_update
        self.mean, self.count = mean, count
        self.std = torch.sqrt(self.summed_variance.clamp_min(0) / count).clamp(std_min_value, std_max_value)
        return self
