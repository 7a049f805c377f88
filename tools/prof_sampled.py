import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
class A: env="CartPole-v1"
actor = bench.Actor("CartPole-v1", 4096, 64, torch.device("cuda"), seed=7, variant="sampled")
for _ in range(8): actor.step()
torch.cuda.synchronize()
