import os, sys, torch
sys.path.insert(0, "/root/repo")
import alphazero_amd as mctx
from bench import make_networks
dev = torch.device("cuda")
for B, S in ((4096, 64), (16384, 256), (4096, 256)):
    pol, val = make_networks("CartPole-v1", 1, dev)
    model = mctx.FusedModel("CartPole-v1", pol, val)
    env = mctx.envs.make("CartPole-v1")
    st = env.reset(0, batch_size=B)
    rf = mctx.FusedRecurrentFn(model)
    logits, value = model.forward(st.obs)
    root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
    ts = []
    for i in range(4):
        stats = {"record_events": True}
        mctx.gumbel_muzero_policy((), 0, root, rf, S, stats=stats)
        torch.cuda.synchronize()
        ts.append(stats["events"][0][0].elapsed_time(stats["events"][0][1]))
    print(f"B={B} S={S}: {min(ts):.2f} ms, {B*S/min(ts)/1e3:.1f} M sims/s, mean depth {stats['sum_depth'].item()/(B*S):.1f}", flush=True)
