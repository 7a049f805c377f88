"""Micro-timings of the pieces of the fused search (diagnostic, not part of the product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import alphazero_amd as mctx
from bench import make_networks

def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

dev = torch.device("cuda")
B = int(os.environ.get("B", 4096))
pol, val = make_networks("CartPole-v1", 1, dev)
model = mctx.FusedModel("CartPole-v1", pol, val)
env = mctx.envs.make("CartPole-v1")
st = env.reset(0, batch_size=B)
print("forward  B=%d: %.1f us" % (B, timeit(lambda: model.forward(st.obs))))
rf = mctx.FusedRecurrentFn(model)
act = torch.zeros(B, dtype=torch.int32, device=dev)
print("recurrent_step: %.1f us" % timeit(lambda: rf((), None, act, st.embedding)))
for S in (1, 2, 4, 8, 16, 32, 64):
    logits, value = model.forward(st.obs)
    root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
    def run():
        stats = {"record_events": True}
        mctx.gumbel_muzero_policy((), 0, root, rf, S, stats=stats)
        return stats
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        stt = run(); torch.cuda.synchronize()
        ts.append(stt["events"][0][0].elapsed_time(stt["events"][0][1]) * 1e3)
    print("fused search S=%d: %.1f us  (%.2f us/sim) mean depth %.2f" % (S, min(ts), min(ts) / S, stt["sum_depth"].item() / (B * S)))
