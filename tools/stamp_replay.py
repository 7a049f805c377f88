import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from alphazero_amd import _native as nat
nat.LIB_PATH = os.path.join(os.getcwd(), "tools", "probes", "libaz_mcts_stamp.so")
import numpy as np, torch
from alphazero_amd import replay_buffers as rb
cap, D, bs = 1 << 20, 30, 16384
q = rb.PrioritizedSamplingQueue(cap, D, bs, per_importance_sampling_beta=1.0)
st = q.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
fill = torch.randn((cap, D), generator=g, device="cuda"); fill[:, -2] = torch.rand(cap, generator=g, device="cuda") + 0.1
for i in range(0, cap, 262144): q.insert(st, fill[i:i + 262144].contiguous())
for _ in range(3): q.sample(st)
torch.cuda.synchronize()
buf = np.zeros(16, np.int64)
nat.lib().az_debug_read_replay_stamps.argtypes = [C.c_void_p]
nat.lib().az_debug_read_replay_stamps(buf.ctypes.data_as(C.c_void_p))
names = ["stage+runs", "run totals (serial)", "chunk offset (serial)", "cum to LDS", "write cum+weights", "max"]
for i, n in enumerate(names): print(f"{n:26s} {int(buf[i+1]-buf[i]):8d} ticks")
