"""Kernel time of the fused search under the PUCT policies, by tie-break noise mode (diagnostic)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import alphazero_amd as mctx
from bench import make_networks
dev = torch.device("cuda")
B, S = 4096, 64
pol, val = make_networks("CartPole-v1", 1, dev)
model = mctx.FusedModel("CartPole-v1", pol, val)
env = mctx.envs.make("CartPole-v1")
st = env.reset(0, batch_size=B)
rf = mctx.FusedRecurrentFn(model)
logits, value = model.forward(st.obs)
root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
def run(name, **kw):
    ts = []
    for i in range(6):
        stats = {"record_events": True}
        out = mctx.sampled_muzero_policy((), i, root, rf, S, num_samples=100, stats=stats, **kw)
        torch.cuda.synchronize()
        ts.append(stats["events"][0][0].elapsed_time(stats["events"][0][1]))
    print(name, "kernel ms", np.round(ts[1:], 3), "mean depth", stats["sum_depth"].item() / (B * S))
run("threefry")
run("zero", tiebreak_noise="zero")
run("injected zeros", tiebreak_noise=torch.zeros((B, S, 8, 2), device=dev))
def run2(name, **kw):
    ts = []
    for i in range(6):
        stats = {"record_events": True}
        out = mctx.muzero_policy((), i, root, rf, S, stats=stats, **kw)
        torch.cuda.synchronize()
        ts.append(stats["events"][0][0].elapsed_time(stats["events"][0][1]))
    print(name, "kernel ms", np.round(ts[1:], 3), "mean depth", stats["sum_depth"].item() / (B * S))
run2("muzero threefry")
run2("muzero zero", tiebreak_noise="zero")

