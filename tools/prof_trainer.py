"""Where a training step's time goes (actor step pieces, collection, SGD) -- diagnostic."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.trainer import Config, Trainer
from alphazero_amd import prng, policies
cfg = Config(env_id="CartPole-v1", policy="sampled", num_envs=4096, num_simulations=64, num_samples=16,
             unroll_length=8, replay_buffer_batch_size=1024, num_minibatches=4, max_replay_size=1 << 20)
tr = Trainer(cfg)
for _ in range(2): tr.training_step()
def T(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
k = (0, 7)
print("actor_step            %.3f ms" % T(lambda: tr.actor_step(k)))
st = tr.env_state
print("  model.forward       %.3f ms" % T(lambda: tr.model.forward(st.obs)))
logits, value = tr.model.forward(st.obs)
from alphazero_amd.base import RootFnOutput
root = RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
print("  sampled policy      %.3f ms" % T(lambda: policies.sampled_muzero_policy((), k, root, tr.recurrent_fn, 64, num_samples=16)))
stats = {"record_events": True}
policies.sampled_muzero_policy((), k, root, tr.recurrent_fn, 64, num_samples=16, stats=stats); torch.cuda.synchronize()
print("    fused kernel only %.3f ms" % stats["events"][0][0].elapsed_time(stats["events"][0][1]))
out = policies.sampled_muzero_policy((), k, root, tr.recurrent_fn, 64, num_samples=16)
print("  actor_post          %.3f ms" % T(lambda: policies.actor_post(out.search_tree, out.action_weights, out.action, logits)))
a, sv, lp = policies.actor_post(out.search_tree, out.action_weights, out.action, logits)
print("  env.step            %.3f ms" % T(lambda: tr.env.step(st, a)))
print("generate_unroll(8)    %.3f ms" % T(lambda: tr.generate_unroll(k, 8), 3))
print("collect               %.3f ms" % T(lambda: tr.collect(k, k), 3))
print("training_step         %.3f ms" % T(lambda: tr.training_step(), 3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.collect(k, k)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
