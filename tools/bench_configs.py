"""Fused-search timings of the baseline configurations beyond the headline one (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphazero_amd as mctx
from bench import make_networks
dev = torch.device("cuda")
CASES = [("gumbel", "CartPole-v1", 4096, 64), ("muzero", "CartPole-v1", 64, 32),
         ("muzero", "CartPole-v1", 4096, 64), ("sampled", "CartPole-v1", 4096, 64),
         ("gumbel", "Acrobot-v1", 8192, 64), ("gumbel", "MountainCar-v0", 8192, 64),
         ("gumbel", "CartPole-v1", 16384, 256)]
if os.environ.get("ONLY"):  # e.g. ONLY=Acrobot
    CASES = [c for c in CASES if os.environ["ONLY"] in c[1]]
for variant, env_id, B, S in CASES:
    pol, val = make_networks(env_id, 1, dev)
    model = mctx.FusedModel(env_id, pol, val)
    env = mctx.envs.make(env_id)
    st = env.reset(0, batch_size=B)
    rf = mctx.FusedRecurrentFn(model)
    logits, value = model.forward(st.obs)
    root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
    ts = []
    for i in range(6):  # two warm-up searches (cold caches, first-launch set-up), then the minimum of four
        stats = {"record_events": True}
        if variant == "gumbel":
            mctx.gumbel_muzero_policy((), i, root, rf, S, stats=stats)
        elif variant == "muzero":
            mctx.muzero_policy((), i, root, rf, S, stats=stats)
        else:
            mctx.sampled_muzero_policy((), i, root, rf, S, num_samples=16, stats=stats)
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(stats["events"][0][0].elapsed_time(stats["events"][0][1]))
    print(f"{variant:8s} {env_id:15s} B={B:6d} S={S:4d}: {min(ts):8.3f} ms  {B*S/min(ts)/1e3:7.1f} M sims/s  "
          f"mean depth {stats['sum_depth'].item()/(B*S):.1f}", flush=True)
