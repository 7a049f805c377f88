"""Random shapes of the tree-only search through the C ABI against the plain-C checker
(differential run; prints the first mismatch).  usage: python tests/fuzz_tree.py [n_cases] [seed]
(not collected by pytest; it lives here because it uses the checker)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import runners
import gpu_util

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for i in range(n_cases):
    variant = ["gumbel", "muzero", "sampled"][rng.integers(3)]
    kw = dict(B=int(rng.integers(1, 70)), S=int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 300])),
              A=int(rng.integers(1, 9)), K=int(rng.choice([1, 1, 2, 8, 16])))
    if rng.random() < 0.3:
        kw["max_depth"] = int(rng.integers(1, 6))
    if rng.random() < 0.3:
        kw["with_invalid"] = True
    if kw["S"] >= 100:
        kw["B"] = min(kw["B"], 20)
    case = runners.synthetic_case(variant, seed=int(rng.integers(1 << 30)), **kw)
    ref = runners.run_policy_c(case)
    out = gpu_util.run_policy_hip(case, "persistent")
    try:
        gpu_util.assert_same(out, ref)
        ok = True
    except AssertionError as e:
        ok, bad = False, bad + 1
        print("MISMATCH", variant, kw, str(e)[:200], flush=True)
    if i % 10 == 0 or not ok:
        print(i, variant, kw, "ok" if ok else "FAIL", flush=True)
print("cases", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
