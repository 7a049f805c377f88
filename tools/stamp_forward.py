"""Per-wave cycle stamps of the forward kernel (needs tools/probes/libaz_mcts_stamp.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import _native as nat
nat.LIB_PATH = os.environ.get("AZ_MCTS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libaz_mcts_stamp.so")
import numpy as np, torch
import alphazero_amd as mctx
from bench import make_networks
dev = torch.device("cuda")
B = int(os.environ.get("B", 4096))
pol, val = make_networks("CartPole-v1", 1, dev)
model = mctx.FusedModel("CartPole-v1", pol, val, hidden_bf16=bool(int(os.environ.get("BF16", "0"))))
obs = torch.randn(B, 4, device=dev)
for _ in range(20):
    model.forward(obs)
torch.cuda.synchronize()
lib = nat.lib()
buf = np.zeros((8, 64), np.int64)
lib.az_debug_read_stamps.argtypes = [C.c_void_p]
lib.az_debug_read_stamps(buf.ctypes.data_as(C.c_void_p))
t0 = buf[:, 0].min()
edges = buf[:, 60:64].copy()
buf[:, 60:64] = 0
if edges.any():  # kernel entry -> inputs staged -> table built -> weights requested -> first stamp of the forward
    for w in range(8):
        e = edges[w]
        print(f"wave {w} entry: stage {int(e[1] - e[0])}  table+barrier {int(e[2] - e[1])}  obs+prime {int(e[3] - e[2])}  -> forward {int(buf[w, 0] - e[3])}  (entry to forward {int(buf[w, 0] - e[0])})")
for w in range(8):
    row = buf[w]
    n = int((row > 0).sum())
    rel = (row[:n] - t0)
    print(f"wave {w}: " + " ".join(str(int(x)) for x in rel))
    print("   deltas: " + " ".join(str(int(x)) for x in np.diff(rel)))
