"""Batched MCTS (mirrors mctx_dist/_src/search.py) on HIP kernels.

`search()` keeps the reference signature.  Depending on the `recurrent_fn` it
drives one of three device paths, all writing the same Tree arrays:

* `networks.FusedRecurrentFn`  -> az_search_fused: ONE persistent launch for all
  simulations, environment step and both MLPs fused into the loop;
* `InjectedRecurrentFn`        -> az_search_injected: ONE persistent launch, the
  recurrent_fn outputs are read from per-(lane, simulation) arrays;
* any other callable           -> per simulation: az_simulate, the Python callback
  on the gathered embeddings, az_expand_backward.
"""
import ctypes as C
import dataclasses

import torch

from . import _native as nat
from . import action_selection
from . import base
from . import prng
from . import seq_halving
from . import tree as tree_lib
from .networks import FusedRecurrentFn

Tree = tree_lib.Tree


@dataclasses.dataclass(frozen=True)
class InjectedRecurrentFn:
    """Pre-computed recurrent_fn outputs indexed by (lane, simulation):
    prior_logits [B,S,A], value [B,S(,K)], reward [B,S], discount [B,S]."""
    prior_logits: torch.Tensor
    value: torch.Tensor
    reward: torch.Tensor
    discount: torch.Tensor


# ---- minimal pytree helpers for caller-owned embeddings
def tree_map(fn, *trees):
    t0 = trees[0]
    if t0 is None:
        return None
    if torch.is_tensor(t0):
        return fn(*trees)
    if isinstance(t0, dict):
        return {k: tree_map(fn, *[t[k] for t in trees]) for k in t0}
    if isinstance(t0, tuple) and hasattr(t0, "_fields"):
        return type(t0)(*[tree_map(fn, *xs) for xs in zip(*trees)])
    if isinstance(t0, (list, tuple)):
        return type(t0)(tree_map(fn, *xs) for xs in zip(*trees))
    if dataclasses.is_dataclass(t0):
        return dataclasses.replace(t0, **{
            f.name: tree_map(fn, *[getattr(t, f.name) for t in trees])
            for f in dataclasses.fields(t0)})
    return t0  # non-array leaves (python scalars, strings) are carried through


def _f32(x, name):
    if not torch.is_tensor(x):
        raise TypeError(f"{name} must be a torch tensor")
    if not x.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {x.device})")
    return x.detach().to(torch.float32).contiguous()


def _key_words(key):
    return (C.c_uint32 * 2)(*key)


def _timed(stats, launch):
    """Runs `launch`; when stats['record_events'] is set, brackets it with HIP events on the
    launch stream and appends the (start, end) pair to stats['events']."""
    if stats is not None and stats.get("record_events"):
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        launch()
        end.record()
        stats.setdefault("events", []).append((start, end))
    else:
        launch()


class _KeyRing:
    """Pinned host slots for the per-simulation keys of consecutive searches: the keys go to the
    device with an asynchronous copy (a copy from pageable memory would make the host wait for
    everything queued on the stream, once per search), and a slot is written again only after the
    copy that last read it has finished."""
    SLOTS = 16

    def __init__(self, S):
        self.host = torch.empty((self.SLOTS, S, 2), dtype=torch.int32).pin_memory()
        self.done = [None] * self.SLOTS
        self.next = 0

    def upload(self, lib, key, S, dev):
        i = self.next
        self.next = (i + 1) % self.SLOTS
        if self.done[i] is not None:
            self.done[i].synchronize()
        slot = self.host[i]
        nat.check(lib.az_sim_keys(_key_words(key), S, C.c_void_p(slot.data_ptr())))
        out = torch.empty((S, 2), dtype=torch.int32, device=dev)
        out.copy_(slot, non_blocking=True)
        self.done[i] = torch.cuda.Event()
        self.done[i].record()
        return out


_key_rings = {}


def build_desc(*, B, S, A, K, root_action_selection_fn, interior_action_selection_fn,
               max_depth, rng_key, tiebreak_noise):
    d = nat.SearchDesc()
    d.B, d.S, d.A, d.K = B, S, A, K
    d.max_depth = S if max_depth is None else int(max_depth)
    action_selection.resolve(root_action_selection_fn, interior_action_selection_fn, d, S)
    key = prng.as_key(rng_key)
    d.key[0], d.key[1] = key
    if d.variant == nat.VARIANT_GUMBEL:
        d.noise_mode, d.noise_depth = nat.NOISE_ZERO, 0
    elif tiebreak_noise is None:
        d.noise_mode, d.noise_depth = nat.NOISE_THREEFRY, 0
    elif isinstance(tiebreak_noise, str) and tiebreak_noise == "zero":
        d.noise_mode, d.noise_depth = nat.NOISE_ZERO, 0
    else:
        # (the noise of a stochastic search is drawn for the decision nodes' actions only)
        An = d.num_decision_actions if d.variant == nat.VARIANT_STOCHASTIC else A
        if tiebreak_noise.dim() != 4 or tiebreak_noise.shape[0] != B or \
                tiebreak_noise.shape[1] != S or tiebreak_noise.shape[3] != An:
            raise ValueError("tiebreak_noise must have shape [B, num_simulations, depth, A]")
        d.noise_mode, d.noise_depth = nat.NOISE_INJECTED, tiebreak_noise.shape[2]
    return d


# Every search needs a device int64 that starts at zero (the kernels add the depths of the expanded
# nodes to it).  A fill kernel for it is ~3.5 us of the GPU's time per search -- as much as the
# root's selection kernel -- so the zeros come out of a pool filled 4096 at a time, one fresh slot
# per search (a slot is never handed out twice).
_zero_pools = {}


def _zero_slot(dev):
    pool = _zero_pools.get(dev)
    if pool is None or pool[1] >= pool[0].shape[0]:
        pool = _zero_pools[dev] = [torch.zeros((4096,), dtype=torch.int64, device=dev), 0]
    slot = pool[0][pool[1]:pool[1] + 1]
    pool[1] += 1
    return slot


def search(params, rng_key, *, root, recurrent_fn, root_action_selection_fn,
           interior_action_selection_fn, num_simulations, max_depth=None, invalid_actions=None,
           extra_data=None, loop_fn=None, tiebreak_noise=None, stats=None):
    """mctx_dist/_src/search.py:31-114.

    Extra keyword arguments (no counterpart in the reference): `tiebreak_noise`
    injects the uniform(0,1) draws of the 1e-7 tie-break noise as a tensor
    [B, num_simulations, depth, A] (or "zero"); `stats`, a dict, receives
    `sum_depth`, a device int64 with the summed depth of all expanded nodes; on input
    `stats['kernel_form'] = 'literal'` selects the one-thread-per-lane kernels (used to
    cross-check the two kernel forms), `stats['sim_keys'] = 'device'` leaves the
    per-simulation keys to the kernels (cross-check of the host-derived ones) and
    `stats['record_events']` brackets the search launch with HIP events.
    `loop_fn` is accepted for signature compatibility and ignored.
    """
    del loop_fn
    lib = nat.lib()
    logits = _f32(root.prior_logits, "root.prior_logits")
    if logits.dim() != 2:
        raise AssertionError("root.prior_logits must have rank 2")  # search.py:351
    B, A = logits.shape
    value = _f32(root.value, "root.value")
    scalar_values = value.dim() == 1
    if scalar_values:
        value2 = value.unsqueeze(-1)
    else:
        value2 = value
    if value2.shape[0] != B or value2.dim() != 2:
        raise AssertionError(f"root.value has shape {tuple(root.value.shape)}, expected "
                             f"[{B}] or [{B}, num_atoms]")  # search.py:354
    K = value2.shape[1]
    S = int(num_simulations)
    dev = logits.device
    noise = None
    if torch.is_tensor(tiebreak_noise):
        noise = _f32(tiebreak_noise, "tiebreak_noise")
    d = build_desc(B=B, S=S, A=A, K=K, root_action_selection_fn=root_action_selection_fn,
                   interior_action_selection_fn=interior_action_selection_fn,
                   max_depth=max_depth, rng_key=rng_key,
                   tiebreak_noise=noise if noise is not None else tiebreak_noise)
    if stats is not None and stats.get("kernel_form") == "literal":
        d.kernel_form = 1  # cross-checks: the one-thread-per-lane kernels
    inv = None if invalid_actions is None else _f32(invalid_actions, "invalid_actions")
    stochastic = d.variant == nat.VARIANT_STOCHASTIC
    if inv is not None and inv.shape != (B, d.num_decision_actions if stochastic else A):
        raise ValueError("invalid_actions must have shape [B, num_actions]")
    if stochastic and isinstance(recurrent_fn, (FusedRecurrentFn, InjectedRecurrentFn)):
        raise NotImplementedError("the stochastic search runs around caller-side node functions")

    inputs = nat.SearchInputs()
    keep = [logits, value2, inv, noise]
    inputs.invalid_actions = nat.ptr(inv)
    inputs.noise = nat.ptr(noise)
    if d.variant == nat.VARIANT_GUMBEL:
        if extra_data is None or not hasattr(extra_data, "root_gumbel"):
            raise ValueError("gumbel action selection needs extra_data.root_gumbel")
        gumbel = _f32(extra_data.root_gumbel, "extra_data.root_gumbel")
        table = seq_halving.device_table(d.max_num_considered, S, dev)
        inputs.root_gumbel = nat.ptr(gumbel)
        inputs.table = nat.ptr(table)
        keep += [gumbel, table]

    if d.noise_mode == nat.NOISE_THREEFRY and not (stats is not None and stats.get("sim_keys") == "device"):
        # the per-simulation keys (search.py:90) form a sequential chain: one CPU core walks it in
        # microseconds, one GPU thread per workgroup would take tens of them
        ring = _key_rings.get((dev, S))
        if ring is None:
            ring = _key_rings[(dev, S)] = _KeyRing(S)
        sim_keys = ring.upload(lib, (int(d.key[0]), int(d.key[1])), S, dev)
        inputs.sim_keys = nat.ptr(sim_keys)
        keep.append(sim_keys)

    arrays = tree_lib.allocate(B, S, A, K, dev, scalar_values=scalar_values)
    tp = nat.TreePtrs()
    for name in tree_lib.ARRAY_FIELDS:
        setattr(tp, name, nat.ptr(arrays[name]))
    sum_depth = _zero_slot(dev)
    stream = nat.stream_ptr()
    N = S + 1

    if isinstance(recurrent_fn, FusedRecurrentFn):
        emb = root.embedding
        if hasattr(emb, "embedding"):
            emb = emb.embedding
        root_state = _f32(emb["state"], "root.embedding['state']")
        root_time = emb["time"].to(torch.int32).contiguous()
        if root_state.shape != (B, 4) or root_time.shape != (B,):
            raise ValueError("fused search needs embedding {'state': [B,4], 'time': [B]}")
        emb_state = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
        emb_time = torch.empty((B, N), dtype=torch.int32, device=dev)
        tp.emb_state, tp.emb_time = nat.ptr(emb_state), nat.ptr(emb_time)
        nat.check(lib.az_tree_init(d, tp, nat.ptr(logits), nat.ptr(value2), nat.ptr(root_state),
                                   nat.ptr(root_time), stream))
        _timed(stats, lambda: nat.check(lib.az_search_fused(
            d, tp, inputs, recurrent_fn.model.native, nat.ptr(sum_depth), stream)))
        embeddings = {"state": emb_state, "time": emb_time}
    elif isinstance(recurrent_fn, InjectedRecurrentFn):
        inj = nat.Injected()
        il = _f32(recurrent_fn.prior_logits, "injected prior_logits")
        iv = _f32(recurrent_fn.value, "injected value").reshape(B, S, K)
        ir = _f32(recurrent_fn.reward, "injected reward")
        idc = _f32(recurrent_fn.discount, "injected discount")
        if il.shape != (B, S, A) or ir.shape != (B, S) or idc.shape != (B, S):
            raise ValueError("injected evaluations must be [B,S,A], [B,S(,K)], [B,S], [B,S]")
        inj.prior_logits, inj.value = nat.ptr(il), nat.ptr(iv)
        inj.reward, inj.discount = nat.ptr(ir), nat.ptr(idc)
        keep += [il, iv, ir, idc]
        nat.check(lib.az_tree_init(d, tp, nat.ptr(logits), nat.ptr(value2), None, None, stream))
        _timed(stats, lambda: nat.check(lib.az_search_injected(
            d, tp, inputs, inj, nat.ptr(sum_depth), stream)))
        embeddings = tree_map(lambda x: x.unsqueeze(1).expand(B, N, *x.shape[1:]), root.embedding)
    else:
        nat.check(lib.az_tree_init(d, tp, nat.ptr(logits), nat.ptr(value2), None, None, stream))

        def _alloc(x):
            buf = torch.zeros((B, N) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            buf[:, 0] = x
            return buf
        embeddings = tree_map(_alloc, root.embedding)
        if stochastic:
            # the kind of every node, [B, N] bytes, which the selection kernel reads in place
            kind = embeddings.is_decision_node
            if kind.dtype != torch.bool or kind.shape != (B, N):
                raise ValueError("StochasticRecurrentState.is_decision_node must be a [B] bool tensor")
            inputs.is_decision = nat.ptr(kind)
        parent = torch.empty((B,), dtype=torch.int32, device=dev)
        action = torch.empty((B,), dtype=torch.int32, device=dev)
        nxt = torch.empty((B,), dtype=torch.int32, device=dev)
        lanes = torch.arange(B, device=dev)
        rng = prng.as_key(rng_key)
        for sim in range(S):
            rng_next, _, expand_key = prng.split(rng, 3)  # search.py:90
            nat.check(lib.az_simulate(d, tp, inputs, sim, nat.ptr(parent), nat.ptr(action),
                                      nat.ptr(nxt), nat.ptr(sum_depth), stream))
            pidx = parent.long()
            emb = tree_map(lambda x: x[lanes, pidx], embeddings)  # search.py:222-223
            # sampled_muzero_policy wraps the recurrent_fn and hands it the first half of
            # split(expand_key) (policies.py:557); the other policies pass expand_key itself
            rf_key = prng.split(expand_key, 2)[0] if d.variant == nat.VARIANT_SAMPLED else expand_key
            step, new_emb = recurrent_fn(params, rf_key, action, emb)
            sl = _f32(step.prior_logits, "recurrent_fn prior_logits")
            sv = _f32(step.value, "recurrent_fn value").reshape(B, -1)
            sr = _f32(step.reward, "recurrent_fn reward")
            sd = _f32(step.discount, "recurrent_fn discount")
            if sl.shape != (B, A) or sr.shape != (B,) or sd.shape != (B,) or sv.shape != (B, K):
                raise AssertionError("recurrent_fn output shapes do not match "
                                     "[B,A], [B], [B], [B(,K)]")  # search.py:227-230
            nidx = nxt.long()

            def _scatter(buf, x):
                buf[lanes, nidx] = x.to(buf.dtype)
                return buf
            embeddings = tree_map(_scatter, embeddings, new_emb)
            nat.check(lib.az_expand_backward(d, tp, nat.ptr(parent), nat.ptr(action),
                                             nat.ptr(nxt), nat.ptr(sl), nat.ptr(sv), nat.ptr(sr),
                                             nat.ptr(sd), stream))
            rng = rng_next

    if stats is not None:
        stats["sum_depth"] = sum_depth
    root_invalid = inv if inv is not None else torch.zeros_like(logits)
    return Tree(embeddings=embeddings, root_invalid_actions=root_invalid, extra_data=extra_data,
                **arrays)
