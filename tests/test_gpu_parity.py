"""GPU parity: the HIP path (through the C ABI) against the plain-C CPU checker.

Bar: selected actions, visit counts and node/parent/child indices bit-exact; on
top of that the float arrays are required to be bit-identical too, because both
sides evaluate the same numerics specification (az_math.h) -- a stricter bar
than the 1e-5 relative tolerance the task allows for Q-values and
action_weights.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import runners

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import gpu_util
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gpu_util


# ---------------------------------------------------------------- numerics spec
@pytest.mark.parametrize("fn,lo,hi", [("exp", -110.0, 90.0), ("log", 0.0, 1e30), ("sin", -40.0, 40.0),
                                      ("cos", -40.0, 40.0), ("swish", -100.0, 100.0),
                                      ("sqrt", 0.0, 1e30), ("rcp", -1e3, 1e3),
                                      ("swish_packed", -100.0, 100.0), ("swish_packed", -8.0, 8.0),
                                      ("sin_select", -40.0, 40.0), ("cos_select", -40.0, 40.0),
                                      ("sin_select", -3e9, 3e9), ("cos_select", -3e9, 3e9)])
def test_numerics_spec_host_device_bits(hip, oracle, fn, lo, hi):
    """Every elementary function gives the same bits on the host and on gfx950.
    swish_packed is swish the way the dense layers' epilogue issues it (packed pairs, a
    shorter form for |x| <= 64): checked against the same host function.  sin_select /
    cos_select: the select-only form the Acrobot step evaluates, on the device, against the
    early-return form on the host."""
    from alphazero_amd import _native as nat
    rng = np.random.default_rng(0)
    x = rng.uniform(lo, hi, 1 << 20).astype(np.float32)
    if fn in ("log", "sqrt"):
        x = np.exp(rng.uniform(-100, 69, 1 << 20)).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 1e-45, 3.4e38]
    if fn == "swish_packed":
        # the switch-over points, tiny and subnormal inputs, and magnitudes spread over all exponents
        x[8:16] = [64.0, -64.0, np.nextafter(np.float32(64), np.float32(65)),
                   -np.nextafter(np.float32(64), np.float32(65)), 1e-38, -1e-38, 1e-42, -88.8]
        n = 1 << 16
        x[16:16 + n] = (np.exp(rng.uniform(-100, 4.2, n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    xd = hip.dev(x)
    yd = torch.empty_like(xd)
    code = 8 if fn == "swish_packed" else oracle._MATH_FN[fn]
    nat.check(nat.lib().az_math_apply(code, x.size, nat.ptr(xd), nat.ptr(yd), nat.stream_ptr()))
    y = yd.cpu().numpy()
    ref = oracle.math_apply({"swish_packed": "swish", "sin_select": "sin", "cos_select": "cos"}.get(fn, fn), x)
    same = (y.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(y) & np.isnan(ref))
    assert same.all(), (fn, x[~same][:5], y[~same][:5], ref[~same][:5])


def test_gumbel_from_bits_host_device(hip, oracle):
    from alphazero_amd import _native as nat
    bits = np.random.default_rng(1).integers(0, 2 ** 32, 1 << 18, dtype=np.uint64).astype(np.uint32)
    bits[:3] = [0, 0xffffffff, 1 << 9]
    xd = hip.dev(bits.view(np.float32))
    yd = torch.empty_like(xd)
    nat.check(nat.lib().az_math_apply(7, bits.size, nat.ptr(xd), nat.ptr(yd), nat.stream_ptr()))
    ref = oracle.math_apply("gumbel_from_bits", bits.view(np.float32))
    assert np.array_equal(yd.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_threefry_device(hip, oracle):
    from alphazero_amd import _native as nat
    from alphazero_amd import prng
    key = np.array([[0, 0], [0xffffffff, 0xffffffff], [0x13198a2e, 0x03707344]], np.uint32)
    ctr = np.array([[0, 0], [0xffffffff, 0xffffffff], [0x243f6a88, 0x85a308d3]], np.uint32)
    kd, cd = hip.dev(key.view(np.int32)), hip.dev(ctr.view(np.int32))
    od = torch.empty_like(kd)
    nat.check(nat.lib().az_threefry(3, nat.ptr(kd), nat.ptr(cd), nat.ptr(od), nat.stream_ptr()))
    out = od.cpu().numpy().view(np.uint32)
    expect = np.array([[0x6b200159, 0x99ba4efe], [0x1cb996fc, 0xbb002be7],
                       [0xc4923a9c, 0x483df7a0]], np.uint32)
    assert np.array_equal(out, expect)
    assert np.array_equal(out, oracle.threefry(key, ctr))
    # host key derivation agrees with the device / C implementation
    for k in ((0, 0), (123, 456)):
        assert [tuple(int(v) for v in r) for r in oracle.key_split(np.array(k, np.uint32), 5)] == \
            prng.split(k, 5)


# ---------------------------------------------------------------- tree kernels
INJECTED_CASES = [
    dict(variant="gumbel", B=64, S=32, A=2, K=1),
    dict(variant="gumbel", B=37, S=16, A=4, K=1),
    # value spreads below the rescaling epsilon (the two-action rule's division path)
    dict(variant="gumbel", B=64, S=32, A=2, K=1, qt_kwargs=dict(epsilon=0.3)),
    dict(variant="gumbel", B=16, S=24, A=6, K=3, with_invalid=True),
    dict(variant="gumbel", B=16, S=24, A=3, K=1, max_depth=3),
    dict(variant="gumbel", B=16, S=16, A=4, K=1, qt_kwargs=dict(use_mixed_value=False)),
    dict(variant="gumbel", B=8, S=16, A=12, K=1, qt_kwargs=dict(rescale_values=False, value_scale=0.5)),
    dict(variant="gumbel", B=8, S=20, A=40, K=1),
    dict(variant="muzero", B=64, S=32, A=2, K=1),
    dict(variant="muzero", B=16, S=24, A=3, K=2, with_invalid=True),
    dict(variant="muzero", B=16, S=24, A=3, K=1, max_depth=2),
    dict(variant="muzero", B=8, S=16, A=4, K=1, qtransform="min_max",
         qt_kwargs=dict(min_value=-1.0, max_value=2.0)),
    dict(variant="muzero", B=8, S=16, A=20, K=12),
    dict(variant="sampled", B=32, S=16, A=4, K=8),
    dict(variant="sampled", B=16, S=24, A=3, K=1, with_invalid=True),
    # 16-atom quantile heads on the built-in environments' action counts (cooperative form)
    dict(variant="sampled", B=20, S=24, A=3, K=16),
    dict(variant="gumbel", B=20, S=40, A=2, K=16),
    dict(variant="muzero", B=8, S=12, A=5, K=16),    # other shapes with K > 8: literal form
    # smallest shapes: one lane, one simulation, one action; a ragged last workgroup
    dict(variant="gumbel", B=1, S=1, A=2, K=1),
    dict(variant="muzero", B=17, S=1, A=1, K=1),
    dict(variant="gumbel", B=3, S=5, A=1, K=2),
]
_ids = lambda k: "-".join(f"{a}{b}" for a, b in k.items())  # noqa: E731


def _case(kw, seed=7):
    kw = dict(kw)
    return runners.synthetic_case(kw.pop("variant"), kw.pop("B"), kw.pop("S"), kw.pop("A"),
                                  kw.pop("K"), seed=seed, **kw)


@pytest.mark.parametrize("kw", INJECTED_CASES, ids=_ids)
def test_persistent_tree_kernel_vs_oracle(hip, kw):
    case = _case(kw)
    ref = runners.run_policy_c(case)
    out = hip.run_policy_hip(case, "persistent")
    hip.assert_same(out, ref)


@pytest.mark.parametrize("kw", [INJECTED_CASES[i] for i in (0, 2, 3, 8, 12)], ids=_ids)
def test_callback_path_vs_oracle(hip, kw):
    case = _case(kw, seed=9)
    ref = runners.run_policy_c(case)
    out = hip.run_policy_hip(case, "callback")
    hip.assert_same(out, ref)
    # the caller-owned embeddings were gathered from the parent and scattered to the child
    t = out["_tree"]
    emb = t.embeddings.cpu().numpy()
    par = out["parents"]
    for b in range(case["B"]):
        for n in range(1, case["S"] + 1):
            if par[b, n] >= 0 and case.get("max_depth") is None:
                assert emb[b, n] == emb[b, par[b, n]] + 1


def test_golden_fixture(hip):
    """Committed golden vectors (generated by tests/golden/make_golden.py from the C checker)."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "mcts_golden.npz")
    g = np.load(path)
    for name in ("gumbel", "muzero", "sampled"):
        case = runners.synthetic_case(name, B=8, S=16, A=3, K=2 if name != "gumbel" else 1, seed=101)
        out = hip.run_policy_hip(case, "persistent")
        for f in hip.INT_FIELDS + hip.FLT_FIELDS + ("action", "action_weights"):
            assert np.array_equal(out[f], g[f"{name}/{f}"].reshape(out[f].shape)), (name, f)


# ---------------------------------------------------------------- model pieces
@pytest.mark.parametrize("env_id", ["CartPole-v1", "MountainCar-v0", "Acrobot-v1", "Acrobot-v2"])
def test_forward_and_recurrent_step_vs_oracle(hip, oracle, env_id):
    import alphazero_amd as mctx
    rng = np.random.default_rng(5)
    case = runners.synthetic_case("gumbel", B=50, S=1, K=3, seed=5, env_id=env_id, random_bias=True)
    model = case["model"]
    fm = hip.make_model(model)
    md = oracle.make_model(env_id, model["policy_layers"], model["value_layers"], model["mean"],
                           model["std"])
    # forward
    obs = oracle.env_reset(env_id, rng.integers(0, 2 ** 32, (50, 2), dtype=np.uint64).astype(np.uint32))["obs"]
    lg, val = fm.forward(hip.dev(obs))
    rl, rv = oracle.forward(md, obs)
    assert np.array_equal(lg.cpu().numpy(), rl) and np.array_equal(val.cpu().numpy(), rv)
    # recurrent step over a few steps, including states far past termination
    state = rng.uniform(-3, 3, (50, 4)).astype(np.float32)
    time = rng.integers(0, 10, 50).astype(np.int32)
    if env_id == "MountainCar-v0":
        state[:, 0] = rng.uniform(-1.2, 0.6, 50)
        state[:, 1] = rng.uniform(-0.07, 0.07, 50)
        state[:, 2:] = 0
    time[0] = 4999 if env_id != "CartPole-v1" else 49999
    frf = mctx.FusedRecurrentFn(fm)
    emb = {"state": hip.dev(state), "time": hip.dev(time)}
    for _ in range(4):
        action = rng.integers(0, fm.num_actions, 50).astype(np.int32)
        out, emb = frf((), None, hip.dev(action), emb)
        o = oracle.env_step(env_id, state, time, action)
        state, time = o["state"], o["time"]
        rl, rv = oracle.forward(md, o["obs"])
        disc = np.float32(0.99) * np.where(o["truncated"] != 0, np.float32(1), np.float32(1) - o["terminated"])
        assert np.array_equal(emb["state"].cpu().numpy(), state)
        assert np.array_equal(emb["time"].cpu().numpy(), time)
        assert np.array_equal(out.reward.cpu().numpy(), o["reward"])
        assert np.array_equal(out.discount.cpu().numpy(), disc.astype(np.float32))
        assert np.array_equal(out.prior_logits.cpu().numpy(), rl)
        assert np.array_equal(out.value.cpu().numpy(), rv)


@pytest.mark.parametrize("policy_hidden,value_hidden,K", [
    ((33, 1024, 64), (32, 48, 16, 48), 1),     # the policy's items take more rounds than the value's:
    ((48, 600, 128, 64), (100, 200), 1),       # a wave other than wave 0 runs the most units
    ((1, 48, 600, 17), (17, 16, 16, 32), 64),
    ((), (), 1),                                # one layer each: the first unit is also the last
    # output heads that run on the vector ALU (at most 8 output columns in the last phase):
    ((64,), (64,), 1),                          # both heads in the last phase, 4 k-blocks each
    ((128, 48), (32, 128), 1),                  # an odd number of k-blocks (3) and 8 of them
    ((256,), (16,), 4),                         # 3 actions + 4 atoms = 7 columns; 16 k-blocks and a single one
    ((32, 32), (256,), 1),                      # only the policy's head is in the last phase
    ((100,), (100,), 8),                        # 11 columns: stays on the matrix pipe
])
def test_forward_network_shapes_vs_oracle(hip, oracle, policy_hidden, value_hidden, K):
    """Forward pass of unusual network shapes (found by tests/fuzz_forward.py) against the checker."""
    rng = np.random.default_rng(9)
    env_id = "MountainCar-v0"
    case = runners.synthetic_case("gumbel", B=37, S=1, K=K, seed=9, env_id=env_id, random_bias=True,
                                  policy_hidden=policy_hidden, value_hidden=value_hidden)
    model = case["model"]
    fm = hip.make_model(model)
    md = oracle.make_model(env_id, model["policy_layers"], model["value_layers"], model["mean"],
                           model["std"])
    obs = rng.uniform(-2, 2, (37, 2)).astype(np.float32)
    lg, val = fm.forward(hip.dev(obs))
    rl, rv = oracle.forward(md, obs)
    assert np.array_equal(lg.cpu().numpy(), rl) and np.array_equal(val.cpu().numpy(), rv)


@pytest.mark.parametrize("policy_hidden,value_hidden,K", [((32, 32, 32), (256, 256, 256, 256, 256), 1),
                                                          ((48, 600, 128, 64), (100, 200), 1),
                                                          ((64,), (64,), 1)])
def test_schedule_table_sources_agree(hip, oracle, policy_hidden, value_hidden, K):
    """The networks' schedule table comes from behind the packed weights (az_model_pack builds it):
    as packed, from a copy of the buffer at another address (the pointers inside are moved by the
    distance), and -- header wiped, as for a buffer that az_model_pack did not fill -- built by the
    kernel itself.  All three give the checker's bits, through the forward kernel and one model step."""
    import alphazero_amd as mctx
    rng = np.random.default_rng(4)
    env_id = "CartPole-v1"
    case = runners.synthetic_case("gumbel", B=21, S=1, K=K, seed=4, env_id=env_id, random_bias=True,
                                  policy_hidden=policy_hidden, value_hidden=value_hidden)
    model = case["model"]
    fm = hip.make_model(model)
    md = oracle.make_model(env_id, model["policy_layers"], model["value_layers"], model["mean"], model["std"])
    obs = rng.uniform(-2, 2, (21, 4)).astype(np.float32)
    rl, rv = oracle.forward(md, obs)
    weights = fm.packed.numel() - (20 + 8 * 16 * 36)  # az_mlp.h: header + [8 waves][16 units][36 words]
    header = fm.packed[weights:weights + 2].view(torch.int32).cpu().numpy()
    assert header[0] != 0 and 1 <= header[1] <= 16
    st = mctx.envs.make(env_id).reset(0, batch_size=21)
    act = hip.dev(rng.integers(0, 2, 21).astype(np.int32))
    rf = mctx.FusedRecurrentFn(fm)
    ref_step = None
    for mode in ("packed", "moved", "wiped"):
        if mode == "moved":
            pad = torch.empty(4099, dtype=torch.float32, device=fm.packed.device)  # (keeps the copy off the old address)
            fm.packed = fm.packed.clone()
            del pad
        elif mode == "wiped":
            fm.packed[weights:weights + 20] = 0
        fm.native.packed = fm.packed.data_ptr()
        lg, val = fm.forward(hip.dev(obs))
        assert np.array_equal(lg.cpu().numpy(), rl) and np.array_equal(val.cpu().numpy(), rv), mode
        out, emb = rf((), None, act, st.embedding)
        step = [x.cpu().numpy() for x in (out.prior_logits, out.value, out.reward, emb["state"])]
        if ref_step is None:
            ref_step = step
        assert all(np.array_equal(a, b) for a, b in zip(step, ref_step)), mode


FUSED_CASES = [
    ("CartPole-v1", "gumbel", 40, 16, 1, {}),
    ("CartPole-v1", "gumbel", 20, 256, 1, {}),   # the S of the largest baseline configuration
    ("CartPole-v1", "muzero", 16, 12, 1, {}),
    ("CartPole-v1", "muzero", 64, 32, 1, {}),    # the smallest baseline configuration (64 lanes x 32 simulations, PUCT)
    ("CartPole-v1", "gumbel", 16, 12, 1, dict(qt_kwargs=dict(use_mixed_value=False))),
    ("MountainCar-v0", "muzero", 16, 12, 1, {}),
    ("Acrobot-v1", "gumbel", 16, 12, 1, {}),
    ("Acrobot-v1", "sampled", 16, 12, 8, {}),
    ("Acrobot-v2", "gumbel", 16, 24, 1, {}),        # AcrobotNoTerm: never terminated, truncation only
    ("Acrobot-v2", "muzero", 16, 12, 1, {}),
    ("CartPole-v1", "sampled", 20, 40, 16, {}),     # 16-atom value head
    ("MountainCar-v0", "gumbel", 16, 24, 16, {}),
    ("CartPole-v1", "gumbel", 16, 20, 1, dict(max_depth=3)),
    ("CartPole-v1", "gumbel", 24, 24, 1, dict(qt_kwargs=dict(epsilon=0.3))),
]


@pytest.mark.parametrize("env_id,variant,B,S,K,extra", FUSED_CASES,
                         ids=[f"{c[0]}-{c[1]}-B{c[2]}-S{c[3]}-K{c[4]}" + ("-x" if c[5] else "")
                              for c in FUSED_CASES])
def test_fused_search_vs_oracle(hip, env_id, variant, B, S, K, extra):
    case = runners.synthetic_case(variant, B=B, S=S, K=K, seed=13, env_id=env_id, random_bias=True,
                                  **extra)
    ref = runners.run_policy_c(case)
    out = hip.run_policy_hip(case, "persistent")
    hip.assert_same(out, ref)
    assert np.array_equal(out["emb_state"], ref["emb_state"])
    assert np.array_equal(out["emb_time"], ref["emb_time"])


def test_fused_equals_callback_path(hip):
    """The fused persistent kernel and the per-simulation callback path build the same tree."""
    case = runners.synthetic_case("gumbel", B=24, S=10, K=1, seed=17, env_id="CartPole-v1")
    a = hip.run_policy_hip(case, "persistent")
    b = hip.run_policy_hip(case, "callback")
    hip.assert_same(a, b)


@pytest.mark.parametrize("kw", [
    dict(variant="gumbel", B=40, S=64, A=2, K=1),
    dict(variant="gumbel", B=20, S=48, A=3, K=4, with_invalid=True),
    dict(variant="gumbel", B=20, S=40, A=2, K=1, max_depth=5),
    dict(variant="gumbel", B=20, S=40, A=2, K=1, max_depth=1),
    dict(variant="gumbel", B=10, S=40, A=8, K=8),
    dict(variant="gumbel", B=24, S=256, A=2, K=1),   # paths deeper than one 32-thread group
    dict(variant="muzero", B=40, S=64, A=3, K=2),    # PUCT with per-visit tie-break noise
    dict(variant="sampled", B=24, S=48, A=4, K=1, with_invalid=True),
    dict(variant="muzero", B=16, S=40, A=8, K=1, max_depth=6),
    dict(variant="sampled", B=12, S=72, A=3, K=16),  # 16 atoms, paths deeper than one pass
], ids=_ids)
def test_literal_and_cooperative_tree_kernels_agree(hip, kw):
    """The literal one-thread-per-lane kernel and the cooperative 32-threads-per-lane kernel
    (cached selections in LDS) build bit-identical trees, and both equal the CPU checker."""
    case = _case(kw, seed=51)
    coop = hip.run_policy_hip(case, "persistent")
    literal = hip.run_policy_hip(case, "persistent", literal=True)
    hip.assert_same(coop, literal)
    hip.assert_same(coop, runners.run_policy_c(case))
    assert coop["sum_depth"] > case["B"] * case["S"] or case.get("max_depth") == 1


@pytest.mark.parametrize("variant", ["muzero", "sampled"])
def test_cooperative_puct_when_only_the_noise_decides(hip, variant):
    """All priors and evaluations equal: every selection is a tie that the 1e-7 noise of
    that very visit breaks, so the cached picks are all ambiguous and re-evaluated exactly."""
    case = runners.synthetic_case(variant, B=24, S=32, A=4, K=1, seed=57)
    case["prior_logits"][:] = 0.25 if variant == "sampled" else 0.0
    case["evals"]["prior_logits"][:] = 0.25 if variant == "sampled" else 0.0
    case["evals"]["value"][:] = 0.5
    case["value"][:] = 0.5
    ref = runners.run_policy_c(case)
    coop = hip.run_policy_hip(case, "persistent")
    hip.assert_same(coop, ref)
    literal = hip.run_policy_hip(case, "persistent", literal=True)
    hip.assert_same(coop, literal)


def test_cooperative_kernel_with_zero_noise_puct(hip, oracle):
    """PUCT selection with the tie-break noise disabled also runs in the cooperative form."""
    import alphazero_amd as mctx
    case = runners.synthetic_case("muzero", B=24, S=32, A=3, K=2, seed=53)
    case["tie_noise"] = None
    ref = runners.run_policy_c(case)
    ev = {k: hip.dev(x) for k, x in case["evals"].items()}
    rf = mctx.InjectedRecurrentFn(ev["prior_logits"], ev["value"], ev["reward"], ev["discount"])
    root = mctx.RootFnOutput(prior_logits=hip.dev(case["prior_logits"]), value=hip.dev(case["value"]),
                             embedding=None)
    out = mctx.muzero_policy((), 0, root, rf, 32, dirichlet_noise=hip.dev(case["dirichlet"]),
                             tiebreak_noise="zero", action_gumbel=hip.dev(case["action_gumbel"]))
    t = out.search_tree
    for f in hip.INT_FIELDS + hip.FLT_FIELDS:
        assert np.array_equal(getattr(t, f).cpu().numpy(), ref[f]), f
    assert np.array_equal(out.action.cpu().numpy(), ref["action"])


def test_literal_and_cooperative_fused_kernels_agree(hip):
    case = runners.synthetic_case("gumbel", B=40, S=48, K=1, seed=55, env_id="CartPole-v1",
                                  policy_hidden=(32, 32), value_hidden=(64, 64))
    coop = hip.run_policy_hip(case, "persistent")
    literal = hip.run_policy_hip(case, "persistent", literal=True)
    hip.assert_same(coop, literal)
    assert np.array_equal(coop["emb_state"], literal["emb_state"])
    ref = runners.run_policy_c(case)
    hip.assert_same(coop, ref)
    assert np.array_equal(coop["emb_state"], ref["emb_state"])


# ---------------------------------------------------------------- full size, properties
def check_invariants(out, S):
    nv, cv, ci = out["node_visits"], out["children_visits"], out["children_index"]
    par, afp = out["parents"], out["action_from_parent"]
    B, N = nv.shape
    assert (nv[:, 0] == S + 1).all()
    assert (cv[:, 0].sum(-1) == S).all()
    assert (par[:, 1:] >= 0).all() and (par[:, 1:] < np.arange(1, N)[None]).all()
    lanes = np.arange(B)[:, None].repeat(N - 1, 1)
    nodes = np.arange(1, N)[None].repeat(B, 0)
    assert (ci[lanes, par[:, 1:], afp[:, 1:]] == nodes).all()
    assert (nv[:, 1:] == 1 + cv[:, 1:].sum(-1)).all()


def test_full_size_tree_kernel_properties(hip):
    """BASELINE config sizes (B=4096, S=64): structural invariants, the A=2 sequential
    halving split (32, 32), and equality with the CPU checker."""
    case = runners.synthetic_case("gumbel", B=4096, S=64, A=2, K=1, seed=21)
    out = hip.run_policy_hip(case, "persistent")
    check_invariants(out, 64)
    assert (out["children_visits"][:, 0] == 32).all()
    ref = runners.run_policy_c(case)
    hip.assert_same(out, ref)


def test_full_size_fused_properties(hip):
    case = runners.synthetic_case("gumbel", B=4096, S=64, K=1, seed=23, env_id="CartPole-v1")
    out = hip.run_policy_hip(case, "persistent")
    check_invariants(out, 64)
    assert (out["children_visits"][:, 0] == 32).all()
    w = out["action_weights"]
    np.testing.assert_allclose(w.sum(-1), 1.0, rtol=1e-6)
    # idempotence: a second run on the same inputs gives the same bits
    out2 = hip.run_policy_hip(case, "persistent")
    hip.assert_same(out, out2)
    # lane independence: the first 64 lanes alone give the same trees
    ref = runners.run_policy_c({**case, "B": 64, **{k: case[k][:64] for k in
                                ("prior_logits", "value", "root_gumbel", "root_state", "root_time")}})
    for f in hip.INT_FIELDS + hip.FLT_FIELDS:
        assert np.array_equal(out[f][:64], ref[f].reshape(out[f][:64].shape)), f


@pytest.mark.parametrize("variant,extra", [("muzero", dict(temperature=0.5)),
                                           ("muzero", dict(temperature=2.0)),
                                           ("sampled", dict(temperature=0.25, sampling_temperature=0.5)),
                                           ("sampled", dict(sampling_temperature=3.0))])
def test_temperatures_vs_oracle(hip, variant, extra):
    """temperature (policies.py:118-119) and sampling_dist_temperature (policies.py:546) away
    from 1: action, action_weights, the root prior and the whole tree against the checker."""
    case = runners.synthetic_case(variant, B=32, S=16, A=4, K=1, seed=61)
    case.update(extra)
    ref = runners.run_policy_c(case)
    out = hip.run_policy_hip(case, "persistent")
    hip.assert_same(out, ref)


def test_largest_configuration_full_size_properties(hip):
    """The large-tree stress configuration at full size (16384 lanes x 256 simulations, fused
    CartPole Gumbel search): structural invariants on every lane, the sequential-halving split,
    and bit equality of the first 64 lanes with the CPU checker (lanes are independent)."""
    B, S = 16384, 256
    case = runners.synthetic_case("gumbel", B=B, S=S, K=1, seed=29, env_id="CartPole-v1")
    out = hip.run_policy_hip(case, "persistent")
    check_invariants(out, S)
    assert (out["children_visits"][:, 0] == S // 2).all()
    np.testing.assert_allclose(out["action_weights"].sum(-1), 1.0, rtol=1e-6)
    assert out["sum_depth"] > B * S
    ref = runners.run_policy_c({**case, "B": 64, **{k: case[k][:64] for k in
                                ("prior_logits", "value", "root_gumbel", "root_state", "root_time")}})
    for f in hip.INT_FIELDS + hip.FLT_FIELDS + ("emb_state", "emb_time", "action", "action_weights"):
        assert np.array_equal(out[f][:64], ref[f].reshape(out[f][:64].shape)), f


@pytest.mark.parametrize("env_id,variant", [("Acrobot-v1", "sampled"), ("MountainCar-v0", "gumbel"),
                                            ("Acrobot-v1", "gumbel")])
def test_three_action_environments_full_size_properties(hip, env_id, variant):
    """The 8192-lane configuration of the three-action environments (the reference trainer's
    default environment and policy among them) at full size: structural invariants on every lane
    and bit equality of the first 64 lanes -- trees, environment states, actions, weights -- with
    the CPU checker."""
    B, S = 8192, 64
    case = runners.synthetic_case(variant, B=B, S=S, K=1, seed=37, env_id=env_id, random_bias=True)
    out = hip.run_policy_hip(case, "persistent")
    check_invariants(out, S)
    np.testing.assert_allclose(out["action_weights"].sum(-1), 1.0, rtol=1e-6)
    head = {k: (v[:64] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in case.items()}
    head["B"] = 64
    ref = runners.run_policy_c(head)
    for f in hip.INT_FIELDS + hip.FLT_FIELDS + ("emb_state", "emb_time", "action", "action_weights"):
        assert np.array_equal(out[f][:64], ref[f].reshape(out[f][:64].shape)), f


# ---------------------------------------------------------------- device RNG paths
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("sim_keys", ["host", "device"])
def test_default_rng_paths_vs_oracle(hip, oracle, ties, sim_keys):
    """Without injected noise the device threefry streams equal the C checker's, with the
    per-simulation keys worked out by the host (az_sim_keys) and by the kernel itself.
    ties: equal priors and evaluations everywhere, so that the visit's own noise decides every
    selection and the keys are actually used."""
    import alphazero_amd as mctx
    from alphazero_amd import prng
    case = runners.synthetic_case("muzero", B=32, S=16, A=3, K=1, seed=31)
    if ties:
        case["prior_logits"][:] = 0.0
        case["dirichlet"][:] = 1.0 / 3.0
        case["evals"]["prior_logits"][:] = 0.0
        case["evals"]["value"][:] = 0.5
        case["value"][:] = 0.5
    ev = {k: hip.dev(x) for k, x in case["evals"].items()}
    rf = mctx.InjectedRecurrentFn(ev["prior_logits"], ev["value"], ev["reward"], ev["discount"])
    root = mctx.RootFnOutput(prior_logits=hip.dev(case["prior_logits"]), value=hip.dev(case["value"]),
                             embedding=None)
    key = (7, 11)
    stats = {"sim_keys": sim_keys}
    out = mctx.muzero_policy((), key, root, rf, 16, dirichlet_noise=hip.dev(case["dirichlet"]), stats=stats)
    torch.cuda.synchronize()
    rng_key, _, search_key = prng.split(key, 3)
    cfg = oracle.make_cfg(32, 16, 3, 1, variant="muzero", noise_mode="threefry", key=search_key)
    tree = oracle.alloc_tree(32, 16, 3, 1)
    prior = oracle.root_muzero(case["prior_logits"], case["dirichlet"], 0.25)
    oracle.tree_init(cfg, tree, prior, case["value"])
    oracle.search_injected(cfg, tree, case["evals"])
    action, weights = oracle.epilogue_visits(cfg, tree, 1.0, key=rng_key)
    t = out.search_tree
    for f in hip.INT_FIELDS + hip.FLT_FIELDS:
        assert np.array_equal(getattr(t, f).cpu().numpy(), tree[f]), f
    assert np.array_equal(out.action.cpu().numpy(), action)
    assert np.array_equal(out.action_weights.cpu().numpy(), weights)


def test_gumbel_and_sampled_default_noise_vs_oracle(hip, oracle):
    import alphazero_amd as mctx
    from alphazero_amd import prng
    case = runners.synthetic_case("gumbel", B=16, S=8, A=5, K=1, seed=33)
    ev = {k: hip.dev(x) for k, x in case["evals"].items()}
    rf = mctx.InjectedRecurrentFn(ev["prior_logits"], ev["value"], ev["reward"], ev["discount"])
    root = mctx.RootFnOutput(prior_logits=hip.dev(case["prior_logits"]), value=hip.dev(case["value"]),
                             embedding=None)
    out = mctx.gumbel_muzero_policy((), 42, root, rf, 8, gumbel_scale=0.5)
    torch.cuda.synchronize()
    _, gkey = prng.split(42, 2)
    _, g = oracle.root_gumbel(case["prior_logits"], key=gkey, gumbel_scale=0.5)
    assert np.array_equal(out.search_tree.extra_data.root_gumbel.cpu().numpy(), g)
    # sampled: categorical draw counts from the device stream
    case = runners.synthetic_case("sampled", B=16, S=8, A=5, K=1, seed=35, num_samples=16)
    root = mctx.RootFnOutput(prior_logits=hip.dev(case["prior_logits"]), value=hip.dev(case["value"]),
                             embedding=None)
    ev = {k: hip.dev(x) for k, x in case["evals"].items()}
    rf = mctx.InjectedRecurrentFn(ev["prior_logits"], ev["value"], ev["reward"], ev["discount"])
    out = mctx.sampled_muzero_policy((), 5, root, rf, 8, num_samples=16,
                                     dirichlet_noise=hip.dev(case["dirichlet"]), tiebreak_noise="zero")
    torch.cuda.synchronize()
    _, skey, _, _ = prng.split(5, 4)
    probs, counts = oracle.root_sampled(case["prior_logits"], case["dirichlet"], 0.25, None, 16, 1.0,
                                        key=skey)
    assert (counts.sum(-1) == 16).all()
    assert np.array_equal(out.search_tree.children_prior_logits[:, 0].cpu().numpy(), probs)


def test_dirichlet_statistics(hip):
    from alphazero_amd import _native as nat
    B, A = 20000, 4
    out = torch.empty((B, A), device="cuda")
    key = (C.c_uint32 * 2)(3, 9)
    nat.check(nat.lib().az_dirichlet(B, A, 0.3, key, nat.ptr(out), nat.stream_ptr()))
    x = out.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(x.sum(-1), 1.0, rtol=1e-5)
    assert (x >= 0).all()
    np.testing.assert_allclose(x.mean(0), 0.25, atol=0.01)
    # Var = a0(1-a0)/(A*alpha+1) with a0 = 1/A
    np.testing.assert_allclose(x.var(0), 0.25 * 0.75 / (A * 0.3 + 1), rtol=0.08)


# ---------------------------------------------------------------- environment + actor glue
@pytest.mark.parametrize("env_id", ["CartPole-v1", "MountainCar-v0", "Acrobot-v1", "Acrobot-v2"])
def test_real_env_step_vs_oracle(hip, oracle, env_id):
    import alphazero_amd as mctx
    env = mctx.envs.make(env_id, episode_length=7)
    st = env.reset(3, batch_size=33)
    keys = oracle.key_split(np.array([0, 3], np.uint32), 33)
    r = oracle.env_reset(env_id, keys)
    assert np.array_equal(st.state.cpu().numpy(), r["state"])
    assert np.array_equal(st.obs.cpu().numpy(), r["obs"])
    ost = dict(state=r["state"], time=r["time"], obs=r["obs"], real_obs=r["obs"].copy(),
               reward=np.zeros(33, np.float32), done=np.zeros(33, np.float32),
               truncation=np.zeros(33, np.float32), steps=np.zeros(33, np.float32),
               rng=np.stack([oracle.key_split(k, 1)[0] for k in keys]).astype(np.uint32))
    assert np.array_equal(st.rng.cpu().numpy().view(np.uint32), ost["rng"])
    rng = np.random.default_rng(2)
    for step in range(20):
        action = rng.integers(0, env.action_size, 33).astype(np.int32)
        st = env.step(st, hip.dev(action))
        ost = oracle.real_env_step(env_id, 7, ost, action)
        for f in ("state", "time", "obs", "real_obs", "reward", "done", "truncation", "steps"):
            assert np.array_equal(getattr(st, f).cpu().numpy(), ost[f]), (step, f)
        assert np.array_equal(st.rng.cpu().numpy().view(np.uint32), ost["rng"])
    assert ost["done"].sum() >= 0 and (ost["steps"] <= 7).all()


def test_summary_and_actor_post_vs_oracle(hip, oracle):
    import alphazero_amd as mctx
    case = runners.synthetic_case("muzero", B=16, S=12, A=4, K=3, seed=41)
    ref = runners.run_policy_c(case)
    out = hip.run_policy_hip(case, "persistent")
    cfg = runners.make_cfg(case)
    tree = {f: ref[f] for f in runners.TREE_FIELDS}
    s_ref = oracle.summary(cfg, tree)
    s = out["_tree"].summary()
    for f in ("visit_counts", "visit_probs", "value", "qvalues"):
        assert np.array_equal(getattr(s, f).cpu().numpy(), s_ref[f]), f
    q = out["_tree"].qvalues(0).cpu().numpy()
    assert np.array_equal(q, s_ref["qvalues"])
    for det in (False, True):
        a, sv, lp = mctx.actor_post(out["_tree"], out["_out"].action_weights, out["_out"].action,
                                    hip.dev(case["prior_logits"]), deterministic=det)
        ra, rsv, rlp = oracle.actor_post(cfg, tree, ref["action_weights"], ref["action"], det,
                                         case["prior_logits"])
        assert np.array_equal(a.cpu().numpy(), ra)
        assert np.array_equal(sv.cpu().numpy(), rsv)
        assert np.array_equal(lp.cpu().numpy(), rlp)


def test_scalar_value_layout(hip):
    """Root value given as [B] (the layout of mctx/_src) -> value arrays without the atom axis."""
    import alphazero_amd as mctx
    case = runners.synthetic_case("gumbel", B=8, S=8, A=3, K=1, seed=43)
    ref = runners.run_policy_c(case)
    ev = {k: hip.dev(x) for k, x in case["evals"].items()}
    rf = mctx.InjectedRecurrentFn(ev["prior_logits"], ev["value"][..., 0], ev["reward"], ev["discount"])
    root = mctx.RootFnOutput(prior_logits=hip.dev(case["prior_logits"]),
                             value=hip.dev(case["value"][:, 0]), embedding=None)
    out = mctx.gumbel_muzero_policy((), 0, root, rf, 8, root_gumbel=hip.dev(case["root_gumbel"]))
    t = out.search_tree
    assert t.node_values.shape == (8, 9) and t.children_values.shape == (8, 9, 3)
    assert np.array_equal(t.node_values.cpu().numpy(), ref["node_values"][..., 0])
    assert np.array_equal(t.children_values.cpu().numpy(), ref["children_values"][..., 0])
    s = t.summary()
    assert s.value.shape == (8,) and s.qvalues.shape == (8, 3)


# ---------------------------------------------------------------- second, independent checker
NUMPY_CASES = [
    dict(variant="gumbel", B=32, S=16, A=4, K=1),
    dict(variant="gumbel", B=64, S=48, A=2, K=1),           # the headline shape's rule instantiation
    dict(variant="gumbel", B=16, S=24, A=6, K=3, with_invalid=True),
    dict(variant="gumbel", B=16, S=24, A=3, K=1, max_depth=3),
    dict(variant="muzero", B=32, S=16, A=4, K=1),
    dict(variant="muzero", B=16, S=24, A=3, K=2, with_invalid=True),
    dict(variant="sampled", B=32, S=16, A=4, K=8),
    dict(variant="sampled", B=16, S=24, A=3, K=1, with_invalid=True),
]


def _assert_close_to_numpy(out, ref, max_tie_lanes, rtol=1e-5, atol=1e-6):
    """The NumPy checker shares no code with the library (NumPy's libm, BLAS products), so
    selection scores may differ in the last bit and pick another action on a floating-point
    near-tie: such lanes must stay rare, all others have identical integer arrays and float
    arrays within the north star's 1e-5 relative."""
    B = out["node_visits"].shape[0]
    same = np.ones(B, bool)
    for f in hip_int_fields:
        same &= (out[f].reshape(B, -1) == ref[f].reshape(B, -1)).all(1)
    assert int((~same).sum()) <= max_tie_lanes, f"{int((~same).sum())} of {B} lanes differ"
    for f in hip_flt_fields + ("action_weights",):
        np.testing.assert_allclose(out[f][same], ref[f].reshape(out[f].shape)[same], rtol=rtol, atol=atol,
                                   err_msg=f)
    assert np.array_equal(out["action"][same], ref["action"][same])


hip_int_fields = ("node_visits", "parents", "action_from_parent", "children_index", "children_visits")
hip_flt_fields = ("raw_values", "node_values", "children_prior_logits", "children_rewards",
                  "children_discounts", "children_values")


@pytest.mark.parametrize("kw", NUMPY_CASES, ids=_ids)
def test_tree_kernel_vs_numpy_checker(hip, kw):
    """The HIP search against oracle/mctx_numpy.py directly: unlike the plain-C checker it does
    not include the library's numerics header, so exp / log / softmax come from NumPy."""
    case = _case(kw, seed=23)
    ref = runners.run_policy_numpy(case)
    out = hip.run_policy_hip(case, "persistent")
    _assert_close_to_numpy(out, ref, max_tie_lanes=max(1, case["B"] // 16))


@pytest.mark.parametrize("env_id,variant", [("CartPole-v1", "gumbel"), ("MountainCar-v0", "muzero"),
                                            ("Acrobot-v1", "sampled")])
def test_fused_search_vs_numpy_checker(hip, env_id, variant):
    """Environment step, networks (NumPy matrix products, libm sin / cos / exp) and tree of the
    fused kernel against the NumPy checker; float32 summation order differs, hence 2e-4."""
    case = runners.synthetic_case(variant, B=8, S=12, K=1, seed=3, env_id=env_id,
                                  policy_hidden=(32, 32), value_hidden=(64, 64, 64), random_bias=True)
    ref = runners.run_policy_numpy(case)
    out = hip.run_policy_hip(case, "persistent")
    _assert_close_to_numpy(out, ref, max_tie_lanes=1, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out["emb_state"], ref["emb_state"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(out["emb_time"], ref["emb_time"])


# ---------------------------------------------------------------- stochastic MuZero
@pytest.mark.parametrize("kw", [dict(B=32, S=16, A=3, C=2), dict(B=16, S=24, A=2, C=4, with_invalid=True),
                                dict(B=16, S=24, A=3, C=3, max_depth=4), dict(B=70, S=40, A=4, C=5)],
                         ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_stochastic_muzero_policy_vs_oracle(hip, kw):
    """stochastic_muzero_policy (policies.py:346-483) around Python node functions against the C
    checker: the decision columns of the tree, the kind of every node, action and weights."""
    import alphazero_amd as mctx
    case = runners.synthetic_stochastic_case(seed=13, **kw)
    ref = runners.run_stochastic_c(case)
    B, S = case["B"], case["S"]
    ev = {k: hip.dev(v) for k, v in case["evals"].items()}
    calls = {"decision": -1, "chance": 0}   # (the policy's first decision call builds the dummy afterstate)

    def decision_fn(params, rng, action, state):
        s = max(calls["decision"], 0)
        calls["decision"] += 1
        return mctx.DecisionRecurrentFnOutput(chance_logits=ev["chance_logits"][:, s],
                                              afterstate_value=ev["afterstate_value"][:, s, 0]), state + 1

    def chance_fn(params, rng, outcome, afterstate):
        s = calls["chance"]
        calls["chance"] += 1
        return mctx.ChanceRecurrentFnOutput(action_logits=ev["action_logits"][:, s], value=ev["value"][:, s, 0],
                                            reward=ev["reward"][:, s], discount=ev["discount"][:, s]), afterstate + 10

    root = mctx.RootFnOutput(prior_logits=hip.dev(case["prior_logits"]), value=hip.dev(case["value"][:, 0]),
                             embedding=torch.zeros((B,), dtype=torch.int32, device="cuda"))
    inv = None if case["invalid"] is None else hip.dev(case["invalid"])
    out = mctx.stochastic_muzero_policy((), 0, root, decision_fn, chance_fn, S, invalid_actions=inv,
                                        max_depth=case["max_depth"], dirichlet_noise=hip.dev(case["dirichlet"]),
                                        tiebreak_noise=hip.dev(case["tie_noise"]),
                                        action_gumbel=hip.dev(case["action_gumbel"]))
    torch.cuda.synchronize()
    assert calls == {"decision": S, "chance": S}
    t = out.search_tree
    assert t.children_index.shape == (B, S + 1, case["A"]) and t.scalar_values
    for f in hip.INT_FIELDS + hip.FLT_FIELDS:
        got = getattr(t, f).cpu().numpy()
        assert np.array_equal(got, ref[f].reshape(got.shape)), f
    assert np.array_equal(t.embeddings.is_decision_node.cpu().numpy().astype(np.uint8), ref["is_decision"])
    assert np.array_equal(out.action.cpu().numpy(), ref["action"])
    assert np.array_equal(out.action_weights.cpu().numpy(), ref["action_weights"])
    # embeddings travel parent -> child: a decision node's state is its parent's afterstate + 10, a
    # chance node's afterstate its parent's state + 1 (11 per pair of levels from the root's 0)
    kind, expanded = ref["is_decision"], ref["node_visits"] > 0
    assert (t.embeddings.state_embedding.cpu().numpy()[(kind == 1) & expanded] % 11 == 0).all()
    assert (t.embeddings.afterstate_embedding.cpu().numpy()[(kind == 0) & expanded] % 11 == 1).all()


def test_stochastic_muzero_policy_default_noise_decides_ties(hip):
    """Equal priors and evaluations everywhere: the decision nodes' picks hang on the 1e-7 noise
    of the device generator alone (uniform(key, [A]) per level, the key split at every level,
    chance levels included)."""
    import alphazero_amd as mctx
    from alphazero_amd import prng
    case = runners.synthetic_stochastic_case(B=24, S=20, A=3, C=2, seed=17)
    case["prior_logits"][:] = 0.0
    case["dirichlet"][:] = 1.0 / 3.0
    case["value"][:] = 0.5
    for k in ("action_logits", "chance_logits"):
        case["evals"][k][:] = 0.0
    for k in ("value", "afterstate_value"):
        case["evals"][k][:] = 0.5
    case["evals"]["reward"][:] = 0.0
    case["tie_noise"] = None
    key = (3, 9)
    _, _, case["search_key"] = prng.split(key, 3)
    ref = runners.run_stochastic_c(case)
    B, S = case["B"], case["S"]
    ev = {k: hip.dev(v) for k, v in case["evals"].items()}
    calls = {"decision": -1, "chance": 0}

    def decision_fn(params, rng, action, state):
        s = max(calls["decision"], 0)
        calls["decision"] += 1
        return mctx.DecisionRecurrentFnOutput(chance_logits=ev["chance_logits"][:, s],
                                              afterstate_value=ev["afterstate_value"][:, s, 0]), state

    def chance_fn(params, rng, outcome, afterstate):
        s = calls["chance"]
        calls["chance"] += 1
        return mctx.ChanceRecurrentFnOutput(action_logits=ev["action_logits"][:, s], value=ev["value"][:, s, 0],
                                            reward=ev["reward"][:, s], discount=ev["discount"][:, s]), afterstate

    root = mctx.RootFnOutput(prior_logits=hip.dev(case["prior_logits"]), value=hip.dev(case["value"][:, 0]),
                             embedding=torch.zeros((B,), dtype=torch.int32, device="cuda"))
    out = mctx.stochastic_muzero_policy((), key, root, decision_fn, chance_fn, S,
                                        dirichlet_noise=hip.dev(case["dirichlet"]),
                                        action_gumbel=hip.dev(case["action_gumbel"]))
    t = out.search_tree
    for f in hip.INT_FIELDS + hip.FLT_FIELDS:
        got = getattr(t, f).cpu().numpy()
        assert np.array_equal(got, ref[f].reshape(got.shape)), f
    # the noise did decide: not every lane walked the same way
    assert len({tuple(r) for r in ref["children_visits"][:, 0].tolist()}) > 1


def test_stochastic_search_is_refused_by_the_one_launch_entry_points(hip):
    import ctypes as C
    from alphazero_amd import _native as nat
    d = nat.SearchDesc()
    d.B, d.S, d.A, d.K, d.variant, d.max_depth, d.num_decision_actions = 4, 4, 5, 1, nat.VARIANT_STOCHASTIC, 4, 5
    t = nat.TreePtrs()
    dummy = torch.zeros(4096, device="cuda")
    for name in nat.TREE_ARRAYS:
        setattr(t, name, nat.ptr(dummy))
    assert nat.lib().az_simulate(d, t, nat.SearchInputs(), 0, nat.ptr(dummy), nat.ptr(dummy), nat.ptr(dummy),
                                 None, nat.stream_ptr()) == -1       # num_decision_actions must be < A
    assert b"num_decision_actions" in nat.lib().az_last_error()
