"""Host time against GPU time of the learner half of a training step (diagnostic)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.trainer import Config, Trainer
cfg = Config(env_id="CartPole-v1", policy="sampled", num_envs=4096, num_simulations=64, num_samples=16,
             unroll_length=8, replay_buffer_batch_size=1024, num_minibatches=4, max_replay_size=1 << 20)
tr = Trainer(cfg)
for _ in range(2): tr.training_step()
k = (0, 7)
flat = tr.collect(k, k)
torch.cuda.synchronize()
for name, sync in (("host only (no sync)", False), ("host + GPU", True)):
    t0 = time.perf_counter()
    for _ in range(5):
        tr.learn(flat)
        if sync: torch.cuda.synchronize()
    t_host = (time.perf_counter() - t0) / 5 * 1e3
    torch.cuda.synchronize()
    print(f"learn, {name}: {t_host:.3f} ms")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record(); tr.learn(flat); e.record(); torch.cuda.synchronize()
print("learn, GPU span by events: %.3f ms" % s.elapsed_time(e))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.learn(flat)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
