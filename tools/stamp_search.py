"""Per-simulation cycle stamps of the fused search (needs `make -C alphazero_amd/csrc stamp`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import _native as nat
nat.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libaz_mcts_stamp.so")
import numpy as np, torch
import alphazero_amd as mctx
from bench import make_networks
dev = torch.device("cuda")
B, S = int(os.environ.get("B", 4096)), int(os.environ.get("S", 64))
pol, val = make_networks("CartPole-v1", 1, dev)
model = mctx.FusedModel("CartPole-v1", pol, val)
env = mctx.envs.make("CartPole-v1")
st = env.reset(0, batch_size=B)
rf = mctx.FusedRecurrentFn(model)
logits, value = model.forward(st.obs)
root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
for _ in range(3):
    mctx.gumbel_muzero_policy((), 0, root, rf, S)
torch.cuda.synchronize()
lib = nat.lib()
buf = np.zeros((64, 16), np.int64)
lib.az_debug_read_sim_stamps.argtypes = [C.c_void_p]
lib.az_debug_read_sim_stamps(buf.ctypes.data_as(C.c_void_p))
names = ["chase", "env+obs", "sync", "MLP", "hdr", "load0", "sync0", "apply0", "load1", "sync1", "apply1"]
print("sim   total | " + " ".join(f"{n:>7}" for n in names))
for sim in range(S):
    row = buf[sim]
    nxt = buf[sim + 1][0] if sim + 1 < S else 0
    seq = [row[0], row[1], row[2], row[3], row[4], row[5], row[6], row[7]]
    d = [int(seq[i + 1] - seq[i]) for i in range(7)]
    extra = []
    if row[8] > row[7]:
        extra = [int(row[8] - row[7]), int(row[9] - row[8]), int(row[10] - row[9])]
    # names: chase=1-0, env=2-1, sync=3-2, MLP=4-3, load0=5-4, sync0=6-5, apply0=7-6
    tot = int(nxt - row[0]) if nxt else -1
    print(f"{sim:3d} {tot:7d} | " + " ".join(f"{x:7d}" for x in d + extra))
