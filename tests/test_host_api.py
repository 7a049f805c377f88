"""CPU tests of the host layer: the C-ABI library loads and exports every symbol
include/az_mcts.h declares, the host-only entry points work, and the Python
mirror of the policy API resolves selection functions / qtransforms the way the
reference's call sites use them.  No GPU compute is launched here.
"""
import ctypes as C
import functools
import os
import re

import numpy as np
import pytest

import alphazero_amd as mctx
from alphazero_amd import _native as nat
from alphazero_amd import action_selection, prng, qtransforms
from oracle import mctx_numpy as mn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    declared = set()
    for h in ("az_mcts.h", "az_replay.h", "az_minatar.h"):
        header = open(os.path.join(ROOT, "include", h)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)   # comments mention other names
        declared |= set(re.findall(r"\b(az_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found"
    lib = nat.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libaz_mcts.so does not export {name}"
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    assert lib.az_build_arch() == b"gfx950"


def test_struct_layouts_match_header():
    # sizes computed by hand from include/az_mcts.h
    assert C.sizeof(nat.SearchDesc) == 23 * 4
    assert C.sizeof(nat.TreePtrs) == 13 * 8
    assert C.sizeof(nat.SearchInputs) == 6 * 8
    assert C.sizeof(nat.Injected) == 4 * 8
    assert C.sizeof(nat.MlpParams) == 4 + 9 * 4 + 8 * 8 * 2
    assert C.sizeof(nat.Model) == 7 * 4 + 2 * 9 * 4 + 4 + 3 * 8  # hidden_bf16 sits right before the pointers
    # include/az_replay.h
    assert C.sizeof(nat.Replay) == 6 * 8 + 2 * 4
    assert C.sizeof(nat.ReplayState) == 3 * 4


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", "/nonexistent/libaz_mcts.so")
    with pytest.raises(nat.NativeLibraryError):
        nat.lib()


def test_error_channel():
    lib = nat.lib()
    rc = lib.az_considered_table(-1, 4, None)
    assert rc == -1
    assert b"table" in lib.az_last_error()
    with pytest.raises(ValueError):
        nat.check(rc)


@pytest.mark.parametrize("m,n", [(16, 64), (16, 256), (2, 64), (3, 64), (1, 5), (0, 3), (8, 3)])
def test_considered_table_matches_checker(m, n):
    assert mctx.get_table_of_considered_visits(m, n) == mn.get_table_of_considered_visits(m, n)
    assert mctx.get_sequence_of_considered_visits(m, n) == mn.get_sequence_of_considered_visits(m, n)


def test_host_threefry_known_answers():
    assert prng.threefry2x32((0, 0), (0, 0)) == (0x6b200159, 0x99ba4efe)
    assert prng.threefry2x32((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff)) == (0x1cb996fc, 0xbb002be7)
    assert prng.threefry2x32((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3)) == (0xc4923a9c, 0x483df7a0)
    assert prng.as_key(5) == (0, 5) and prng.as_key((1 << 32) + 7) == (1, 7)
    assert prng.as_key(np.array([3, 4], np.uint32)) == (3, 4)
    assert len(prng.split(0, 3)) == 3 and prng.split(0, 3)[1] == prng.threefry2x32((0, 0), (0, 1))


def test_selection_resolution_like_the_reference_call_sites():
    # mctx_dist/_src/policies.py:93-100
    interior = functools.partial(action_selection.muzero_action_selection, pb_c_base=19652,
                                 pb_c_init=1.25, qtransform=mctx.qtransform_by_parent_and_siblings)
    root = functools.partial(interior, depth=0)
    d = action_selection.resolve(root, interior, nat.SearchDesc(), 8)
    assert d.variant == nat.VARIANT_MUZERO and d.qtransform == nat.QT_PARENT_SIBLINGS
    assert d.pb_c_base == 19652.0 and abs(d.q_epsilon - 1e-8) < 1e-15
    # alphazero.py:342-352: gumbel with partial(qtransform_completed_by_mix_value, use_mixed_value=False)
    q = functools.partial(mctx.qtransform_completed_by_mix_value, use_mixed_value=False)
    d = action_selection.resolve(
        functools.partial(action_selection.gumbel_muzero_root_action_selection, num_simulations=8,
                          max_num_considered_actions=16, qtransform=q),
        functools.partial(action_selection.gumbel_muzero_interior_action_selection, qtransform=q),
        nat.SearchDesc(), 8)
    assert d.variant == nat.VARIANT_GUMBEL and d.qtransform == nat.QT_COMPLETED_MIX
    assert d.use_mixed_value == 0 and d.rescale_values == 1 and d.max_num_considered == 16
    assert abs(d.value_scale - 0.1) < 1e-7 and d.maxvisit_init == 50.0
    # sampled pair
    s = functools.partial(action_selection.sampled_muzero_action_selection, pb_c_init=2.0)
    d = action_selection.resolve(functools.partial(s, depth=0), s, nat.SearchDesc(), 8)
    assert d.variant == nat.VARIANT_SAMPLED and d.pb_c_init == 2.0


def test_unsupported_configurations_are_rejected():
    with pytest.raises(NotImplementedError):
        qtransforms.apply_to_desc(lambda tree, node: None, nat.SearchDesc())
    with pytest.raises(TypeError):
        qtransforms.apply_to_desc(functools.partial(mctx.qtransform_by_min_max), nat.SearchDesc())
    with pytest.raises(TypeError):
        qtransforms.apply_to_desc(
            functools.partial(mctx.qtransform_by_parent_and_siblings, bogus=1), nat.SearchDesc())
    with pytest.raises(NotImplementedError):
        action_selection.resolve(action_selection.muzero_action_selection,
                                 action_selection.gumbel_muzero_interior_action_selection,
                                 nat.SearchDesc(), 8)
    with pytest.raises(NotImplementedError):
        mctx.envs.env_code("Breakout-MinAtar")
    with pytest.raises(NotImplementedError):
        mctx.qtransform_by_parent_and_siblings(None, 0)


def test_policy_argument_checks_without_gpu():
    import torch
    root = mctx.RootFnOutput(prior_logits=torch.zeros(2, 3), value=torch.zeros(2), embedding=None)
    with pytest.raises(AssertionError):  # chex.assert_type(..., float), policies.py:508
        mctx.muzero_policy((), 0, root, None, 4, dirichlet_alpha=1)
    assert root.replace(value=torch.ones(2)).value.sum() == 2
    assert mctx.envs.observation_size("Acrobot-v1") == 6 and mctx.envs.action_size("CartPole-v1") == 2


def test_api_surface_matches_mctx_dist():
    """Names the reference's mctx_dist/__init__.py exports for this path."""
    for name in ["GumbelMuZeroExtraData", "gumbel_muzero_interior_action_selection",
                 "gumbel_muzero_root_action_selection", "muzero_action_selection",
                 "PolicyOutput", "RecurrentFnOutput", "RootFnOutput", "gumbel_muzero_policy",
                 "muzero_policy", "sampled_muzero_policy", "qtransform_by_min_max",
                 "qtransform_by_parent_and_siblings", "qtransform_completed_by_mix_value",
                 "search", "SearchSummary", "Tree"]:
        assert hasattr(mctx, name), name
    import inspect
    sig = inspect.signature(mctx.gumbel_muzero_policy)
    assert list(sig.parameters)[:8] == ["params", "rng_key", "root", "recurrent_fn",
                                        "num_simulations", "invalid_actions", "max_depth", "loop_fn"]
    assert sig.parameters["max_num_considered_actions"].default == 16
    sig = inspect.signature(mctx.sampled_muzero_policy)
    assert list(sig.parameters)[:9] == ["params", "rng_key", "root", "recurrent_fn",
                                        "num_simulations", "num_samples", "invalid_actions",
                                        "max_depth", "loop_fn"]
    assert sig.parameters["num_samples"].default == 100


def test_golden_fixture_matches_checker():
    """The committed golden vectors are reproduced by the plain-C checker."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    from oracle import runners
    g = np.load(os.path.join(ROOT, "tests", "golden", "mcts_golden.npz"))
    for name, case in make_golden.cases():
        res = runners.run_policy_c(case)
        for f in make_golden.FIELDS:
            assert np.array_equal(res[f], g[f"{name}/{f}"]), (name, f)
        assert res["sum_depth"] == int(g[f"{name}/sum_depth"])


def test_sim_keys_follow_the_search_loop_chain():
    """az_sim_keys against the chain of search.py:90 written with the host threefry:
    rng, simulate_key, expand_key = split(rng, 3) per simulation."""
    import ctypes as C
    import numpy as np
    lib = nat.lib()
    S, key = 37, (0x12345678, 0x9abcdef0)
    out = np.zeros((S, 2), np.uint32)
    nat.check(lib.az_sim_keys((C.c_uint32 * 2)(*key), S, out.ctypes.data_as(C.c_void_p)))
    rng = key
    for i in range(S):
        rng, simulate_key, _ = prng.split(rng, 3)
        assert tuple(int(x) for x in out[i]) == simulate_key, i
    assert lib.az_sim_keys(None, S, out.ctypes.data_as(C.c_void_p)) == -1
