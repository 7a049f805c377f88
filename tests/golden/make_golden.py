"""Generates tests/golden/mcts_golden.npz from the plain-C CPU checker.

The reference itself cannot be run in the build environment (its JAX runtime is
not installable), so these vectors freeze the CPU checker's outputs -- which the
hand-derived known-answer tests pin -- rather than outputs of the reference.
Run from the repository root:  python tests/golden/make_golden.py

The `cartpole_gumbel` case runs networks, so its float arrays follow the numerics specification's
summation order (alphazero_amd/csrc/az_math.h); it was regenerated when small output heads became
four interleaved chains (last-bit changes in four float arrays, no discrete output moved).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import runners  # noqa: E402

FIELDS = runners.TREE_FIELDS + ("action", "action_weights")


def cases():
    for name in ("gumbel", "muzero", "sampled"):
        yield name, runners.synthetic_case(name, B=8, S=16, A=3, K=2 if name != "gumbel" else 1,
                                           seed=101)
    yield "cartpole_gumbel", runners.synthetic_case(
        "gumbel", B=4, S=8, K=1, seed=103, env_id="CartPole-v1", policy_hidden=(8, 8),
        value_hidden=(16, 16))


def main():
    out = {}
    for name, case in cases():
        res = runners.run_policy_c(case)
        for f in FIELDS:
            out[f"{name}/{f}"] = res[f]
        out[f"{name}/sum_depth"] = np.int64(res["sum_depth"])
    path = os.path.join(os.path.dirname(__file__), "mcts_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
