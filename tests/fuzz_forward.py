"""Random network shapes through the forward kernel against the plain-C checker (differential
run, not collected by pytest).  usage: python tests/fuzz_forward.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import runners, c_oracle as oracle
import gpu_util

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
WIDTHS = [1, 5, 16, 17, 31, 32, 33, 48, 64, 100, 128, 200, 256, 257, 300, 512, 600, 1024]
bad = rejected = 0
for i in range(n_cases):
    env_id = ["CartPole-v1", "MountainCar-v0", "Acrobot-v1", "Acrobot-v2"][rng.integers(4)]
    ph = tuple(int(rng.choice(WIDTHS)) for _ in range(rng.integers(0, 7)))
    vh = tuple(int(rng.choice(WIDTHS)) for _ in range(rng.integers(0, 7)))
    K = int(rng.choice([1, 3, 8, 20, 64]))
    B = int(rng.integers(1, 60))
    tag = f"{env_id} policy{ph} value{vh} K={K} B={B}"
    case = runners.synthetic_case("gumbel", B=B, S=1, K=K, seed=int(rng.integers(1 << 30)), env_id=env_id,
                                  random_bias=True, policy_hidden=ph, value_hidden=vh)
    model = case["model"]
    obs = rng.uniform(-2, 2, (B, model["policy_layers"][0][0].shape[0])).astype(np.float32)
    try:
        fm = gpu_util.make_model(model)
        lg, val = fm.forward(gpu_util.dev(obs))
    except (ValueError, NotImplementedError) as e:   # shapes the library refuses, with its reason
        rejected += 1
        print(i, tag, "rejected:", str(e)[:90], flush=True)
        continue
    md = oracle.make_model(env_id, model["policy_layers"], model["value_layers"], model["mean"], model["std"])
    rl, rv = oracle.forward(md, obs)
    ok = np.array_equal(lg.cpu().numpy(), rl) and np.array_equal(val.cpu().numpy(), rv)
    bad += not ok
    print(i, tag, "ok" if ok else "MISMATCH", flush=True)
print("cases", n_cases, "rejected", rejected, "mismatches", bad)
sys.exit(1 if bad else 0)
