"""Times the fused search kernel under the PUCT rules with and without the tie-break noise stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from alphazero_amd import _native as nat
if os.environ.get("COUNT"):
    nat.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libaz_mcts_stamp.so")
import ctypes as C
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
def run(name, fn, **kw):
    ts, dep = [], 0
    for i in range(6):
        stats = {"record_events": True}
        fn((), i, root, rf, S, stats=stats, **kw)
        torch.cuda.synchronize()
        ts.append(stats["events"][0][0].elapsed_time(stats["events"][0][1]))
        dep = int(stats["sum_depth"]) / (B * S)
    extra = ""
    if os.environ.get("COUNT"):
        buf = (C.c_ulonglong * 4)()
        nat.lib().az_debug_read_flag_counts(buf, 1)
        extra = "  per call: repeated walks %.0f, exact selections %.0f, key splits %.0f, walks %.0f" % tuple(x / 6 for x in buf)
    print(f"{name:40s} {np.mean(ts[1:]):.4f} ms  mean depth {dep:.2f}{extra}", flush=True)
run("gumbel", mctx.gumbel_muzero_policy)
run("muzero threefry", mctx.muzero_policy)
run("muzero zero noise", mctx.muzero_policy, tiebreak_noise="zero")
run("sampled(100) threefry", mctx.sampled_muzero_policy, num_samples=100)
run("sampled(100) zero noise", mctx.sampled_muzero_policy, num_samples=100, tiebreak_noise="zero")
