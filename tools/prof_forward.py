"""Runs the network forward kernel repeatedly (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import alphazero_amd as mctx
from bench import make_networks
dev = torch.device("cuda")
B = int(os.environ.get("B", 4096))
pol, val = make_networks("CartPole-v1", 1, dev)
model = mctx.FusedModel("CartPole-v1", pol, val)
env = mctx.envs.make("CartPole-v1")
st = env.reset(0, batch_size=B)
for _ in range(int(os.environ.get("N", 100))):
    model.forward(st.obs)
torch.cuda.synchronize()
