"""Per-simulation cycle stamps of the tree-only search kernel (needs `make -C alphazero_amd/csrc stamp`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AZ_MCTS_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libaz_mcts_stamp.so"))
from alphazero_amd import _native as nat
import numpy as np, torch
import alphazero_amd as mctx
dev = torch.device("cuda")
B, S, A = int(os.environ.get("B", 4096)), int(os.environ.get("S", 64)), 2
g = torch.Generator(device=dev); g.manual_seed(2)
rf = mctx.InjectedRecurrentFn(torch.randn((B, S, A), generator=g, device=dev),
                              torch.rand((B, S, 1), generator=g, device=dev) * 2 - 1,
                              torch.ones((B, S), device=dev), torch.full((B, S), 0.99, device=dev))
root = mctx.RootFnOutput(prior_logits=torch.randn((B, A), generator=g, device=dev),
                         value=torch.rand((B, 1), generator=g, device=dev) * 2 - 1, embedding=None)
for i in range(3):
    mctx.gumbel_muzero_policy((), i, root, rf, S)
torch.cuda.synchronize()
lib = nat.lib()
buf = np.zeros((64, 8), np.int64)
lib.az_debug_read_tree_stamps.argtypes = [C.c_void_p]
lib.az_debug_read_tree_stamps(buf.ctypes.data_as(C.c_void_p))
names = ["walk", "depth", "prefetch", "rows", "chain", "values", "writes", "select"]
print("sim   total | " + " ".join(f"{n:>9}" for n in names))
for sim in range(min(S, 64)):
    r = buf[sim]
    nxt = buf[sim + 1][0] if sim + 1 < min(S, 64) else r[7]
    d = [int(r[i + 1] - r[i]) for i in range(7)] + [int(nxt - r[7])]
    print(f"{sim:3d} {int(nxt - r[0]):7d} | " + " ".join(f"{x:9d}" for x in d))
edges = np.zeros(4, np.int64)
lib.az_debug_read_tree_edges.argtypes = [C.c_void_p]
lib.az_debug_read_tree_edges(edges.ctypes.data_as(C.c_void_p))
print("kernel: prologue %d, simulations %d, epilogue %d cycles" % tuple(int(edges[i + 1] - edges[i]) for i in range(3)))
