"""Per-simulation cycle stamps of the fused search (needs `make -C alphazero_amd/csrc stamp`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import _native as nat
nat.LIB_PATH = os.environ.get("AZ_MCTS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libaz_mcts_stamp.so")
import numpy as np, torch
import alphazero_amd as mctx
from bench import make_networks
dev = torch.device("cuda")
B, S = int(os.environ.get("B", 4096)), int(os.environ.get("S", 64))
ENV = os.environ.get("ENV", "CartPole-v1")
pol, val = make_networks(ENV, 1, dev)
K = int(os.environ.get("K", 1))
if K > 1:  # a K-atom value head on the same trunk
    g = torch.Generator(device=dev); g.manual_seed(2)
    val = mctx.init_mlp((mctx.envs.observation_size(ENV), 256, 256, 256, 256, 256, K), generator=g, device=dev)
model = mctx.FusedModel(ENV, pol, val, hidden_bf16=bool(int(os.environ.get("BF16", "0"))))
env = mctx.envs.make(ENV)
st = env.reset(0, batch_size=B)
rf = mctx.FusedRecurrentFn(model)
logits, value = model.forward(st.obs)
root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
for _ in range(3):
    variant = os.environ.get("VARIANT", "gumbel")
    noise = {"tiebreak_noise": "zero"} if variant.endswith("_zero") else {}
    if variant.startswith("sampled"):
        mctx.sampled_muzero_policy((), 0, root, rf, S, num_samples=100, **noise)
    elif variant.startswith("muzero"):
        mctx.muzero_policy((), 0, root, rf, S, **noise)
    else:
        mctx.gumbel_muzero_policy((), 0, root, rf, S)
torch.cuda.synchronize()
lib = nat.lib()
buf = np.zeros((64, 16), np.int64)
lib.az_debug_read_sim_stamps.argtypes = [C.c_void_p]
lib.az_debug_read_sim_stamps(buf.ctypes.data_as(C.c_void_p))
names = ["walk", "leaf+emb", "env", "obs", "rows", "prime", "barrier", "MLP", "update"]
order = [0, 1, 6, 7, 8, 9, 2, 3, 4]
print("sim   total | " + " ".join(f"{n:>8}" for n in names))
tot_sum = np.zeros(len(names) + 1)
for sim in range(min(S, 64)):
    row = buf[sim]
    nxt = buf[sim + 1][0] if sim + 1 < min(S, 64) else row[5]
    pts = [row[i] for i in order] + [nxt]
    d = [int(pts[i + 1] - pts[i]) for i in range(len(names))]
    tot_sum += np.array([int(nxt - row[0])] + d)
    print(f"{sim:3d} {int(nxt - row[0]):7d} | " + " ".join(f"{x:8d}" for x in d))
print("mean", " ".join(f"{x:8.0f}" for x in tot_sum / min(S, 64)))
# inside the update: stamps 3..6 of the path update (pass 0 of the last pass loop) and its end
n = min(S, 64)
cols = [4, 12, 13, 14, 15, 5]
seg = np.array([[buf[i][cols[k + 1]] - buf[i][cols[k]] for k in range(5)] for i in range(n)], dtype=np.float64)
print("update: load %.0f  chain %.0f  newval %.0f  store %.0f  select+entry %.0f" % tuple(seg.mean(0)))
