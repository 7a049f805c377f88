"""The bfloat16 hidden-layer mode of the networks (az_model_pack_bf16, FusedModel(hidden_bf16=True)).

Not the reference's arithmetic (float32 throughout) and not comparable bit for bit with the CPU
checker: the forward pass is checked against a NumPy evaluation that rounds the same weights and
the same layer inputs to bfloat16 (tolerance of the float32 summation order), against the
float32 model (bfloat16's three digits), and the fused search against the per-simulation
callback path of the same model, which must build the identical tree (same kernels' numerics)."""
import numpy as np
import pytest

from oracle import runners

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    import gpu_util
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gpu_util


def _bf16(x):
    """float32 -> nearest-even bfloat16, returned as float32."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def _swish(x):
    return x / (1.0 + np.exp(-x))


def _mlp_ref(layers, x, hidden_bf16):
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        if hidden_bf16 and 1 <= i <= n - 2:
            x = _bf16(x).astype(np.float64) @ _bf16(w).astype(np.float64) + b
        else:
            x = x.astype(np.float64) @ w.astype(np.float64) + b
        x = x.astype(np.float32)
        if i < n - 1:
            x = _swish(x.astype(np.float64)).astype(np.float32)
    return x


def _layers(rng, sizes):
    return [((rng.uniform(-1, 1, (a, b)) * np.sqrt(3.0 / a)).astype(np.float32),
             rng.uniform(-0.2, 0.2, b).astype(np.float32)) for a, b in zip(sizes[:-1], sizes[1:])]


@pytest.mark.parametrize("env_id,policy_hidden,value_hidden,K", [
    ("CartPole-v1", (32, 32, 32, 32), (256, 256, 256, 256, 256), 1),   # the reference's default networks
    ("Acrobot-v1", (32, 32), (64, 128, 64), 1),
    ("MountainCar-v0", (64, 32, 96), (512, 32), 3),                    # a 512-wide input: 16 k-blocks of 32
])
def test_forward_with_bf16_hidden_layers(hip, env_id, policy_hidden, value_hidden, K):
    import alphazero_amd as mctx
    rng = np.random.default_rng(5)
    obs_dim, A = mctx.envs.observation_size(env_id), mctx.envs.action_size(env_id)
    pol = _layers(rng, (obs_dim,) + policy_hidden + (A,))
    val = _layers(rng, (obs_dim,) + value_hidden + (K,))
    mean = rng.uniform(-0.1, 0.1, obs_dim).astype(np.float32)
    std = rng.uniform(0.5, 1.5, obs_dim).astype(np.float32)
    dev = hip.dev
    mk = lambda bf: mctx.FusedModel(env_id, [(dev(w), dev(b)) for w, b in pol], [(dev(w), dev(b)) for w, b in val],
                                    dev(mean), dev(std), hidden_bf16=bf)
    obs = rng.uniform(-2, 2, (53, obs_dim)).astype(np.float32)
    x = ((obs - mean) / std).astype(np.float32)
    lg, v = (t.cpu().numpy() for t in mk(True).forward(dev(obs)))
    for got, ref in ((lg, _mlp_ref(pol, x, True)), (v, _mlp_ref(val, x, True))):
        # nearly every output to the float32 summation order; an activation that sits on a
        # bfloat16 rounding boundary may round the other way on the device (its float32 value
        # differs in the last bit) and move an output by the weight times 2^-8 of it
        close = np.abs(got - ref) <= 2e-4 * np.abs(ref) + 2e-5
        assert close.mean() >= 0.9, close.mean()
        np.testing.assert_allclose(got, ref, rtol=5e-3, atol=2e-3)
    # against float32: bfloat16 keeps 8 bits of the inputs of every hidden layer
    lg32, v32 = (t.cpu().numpy() for t in mk(False).forward(dev(obs)))
    assert np.abs(lg - lg32).max() < 0.05 * max(1.0, np.abs(lg32).max())
    assert np.abs(v - v32).max() < 0.05 * max(1.0, np.abs(v32).max())
    assert not np.array_equal(lg, lg32)                      # (the mode is really a different arithmetic)


def test_bf16_hidden_layers_need_widths_of_32(hip):
    import alphazero_amd as mctx
    rng = np.random.default_rng(1)
    pol = _layers(rng, (4, 48, 48, 2))
    val = _layers(rng, (4, 64, 64, 1))
    dev = hip.dev
    with pytest.raises((NotImplementedError, RuntimeError, ValueError)):
        mctx.FusedModel("CartPole-v1", [(dev(w), dev(b)) for w, b in pol], [(dev(w), dev(b)) for w, b in val],
                        hidden_bf16=True)


@pytest.mark.parametrize("env_id,variant", [("CartPole-v1", "gumbel"), ("Acrobot-v1", "gumbel"),
                                            ("CartPole-v1", "muzero")])
def test_fused_search_with_bf16_hidden_layers_equals_its_callback_path(hip, env_id, variant, monkeypatch):
    """One model, two routes: the fused persistent search and the per-simulation kernels around
    az_recurrent_step evaluate the networks with the same code, so the trees must be identical;
    and the tree satisfies the search's structural invariants."""
    case = runners.synthetic_case(variant, B=40, S=24, K=1, seed=23, env_id=env_id, random_bias=True)
    import alphazero_amd as mctx
    real = mctx.FusedModel
    monkeypatch.setattr(mctx, "FusedModel", lambda *a, **k: real(*a, hidden_bf16=True, **k))
    a = hip.run_policy_hip(case, "persistent")
    b = hip.run_policy_hip(case, "callback")
    hip.assert_same(a, b)
    assert np.array_equal(a["emb_state"], b["emb_state"])
    S = case["S"]
    assert (a["node_visits"][:, 0] == S + 1).all() and (a["children_visits"][:, 0].sum(-1) == S).all()
    par = a["parents"][:, 1:]
    assert (par >= 0).all() and (par < np.arange(1, S + 1)[None]).all()
    monkeypatch.setattr(mctx, "FusedModel", real)
    f32 = hip.run_policy_hip(case, "persistent")
    assert not np.array_equal(f32["node_values"], a["node_values"])
    assert (f32["action"] == a["action"]).mean() > 0.7
