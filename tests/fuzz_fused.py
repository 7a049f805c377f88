"""Random network shapes / variants / sizes through the fused search against the plain-C checker
(differential run, not collected by pytest).  usage: python tests/fuzz_fused.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import runners
import gpu_util

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
WIDTHS = [1, 5, 16, 17, 32, 33, 48, 64, 100, 128, 200, 256, 300]
bad = rejected = 0
for i in range(n_cases):
    env_id = ["CartPole-v1", "MountainCar-v0", "Acrobot-v1", "Acrobot-v2"][rng.integers(4)]
    variant = ["gumbel", "muzero", "sampled"][rng.integers(3)]
    ph = tuple(int(rng.choice(WIDTHS)) for _ in range(rng.integers(0, 5)))
    vh = tuple(int(rng.choice(WIDTHS)) for _ in range(rng.integers(0, 6)))
    K = int(rng.choice([1, 1, 2, 8]))
    B, S = int(rng.integers(1, 40)), int(rng.choice([1, 3, 8, 20, 40]))
    kw = {}
    if rng.random() < 0.3:
        kw["max_depth"] = int(rng.integers(1, 5))
    tag = f"{env_id} {variant} policy{ph} value{vh} K={K} B={B} S={S} {kw}"
    print(i, tag, "...", flush=True)
    case = runners.synthetic_case(variant, B=B, S=S, K=K, seed=int(rng.integers(1 << 30)), env_id=env_id,
                                  random_bias=True, policy_hidden=ph, value_hidden=vh, **kw)
    ref = runners.run_policy_c(case)
    try:
        out = gpu_util.run_policy_hip(case, "persistent")
    except (ValueError, NotImplementedError) as e:
        rejected += 1
        print("   rejected:", str(e)[:90], flush=True)
        continue
    try:
        gpu_util.assert_same(out, ref)
        assert np.array_equal(out["emb_state"], ref["emb_state"]) and np.array_equal(out["emb_time"], ref["emb_time"])
        print("   ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("   MISMATCH", str(e)[:200], flush=True)
print("cases", n_cases, "rejected", rejected, "mismatches", bad)
sys.exit(1 if bad else 0)
