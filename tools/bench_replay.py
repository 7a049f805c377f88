"""Timing of the GPU replay queue at the reference's sizes (diagnostic; prints one JSON line).

cap = 2^20 rows, D = 30 floats (CartPole MCTSTransition: 4*obs + A + 2K + 10), 16384 draws
(replay_buffer_batch_size 128 * num_minibatches 128, alphazero.py:937), inserts of 4096 rows.
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd import replay_buffers as rb

cap, D, bs, n_ins = 1 << 20, 30, 16384, 4096
q = rb.PrioritizedSamplingQueue(cap, D, bs, per_importance_sampling_beta=1.0)
st = q.init(0)
g = torch.Generator(device="cuda").manual_seed(0)
fill = torch.randn((cap, D), generator=g, device="cuda")
fill[:, -2] = torch.rand(cap, generator=g, device="cuda") + 0.1
for i in range(0, cap, 262144):
    q.insert(st, fill[i:i + 262144].contiguous())
up = fill[:n_ins].contiguous()
pri = torch.rand(bs, device="cuda") + 0.1


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


t_ins = timeit(lambda: q.insert(st, up))
t_smp = timeit(lambda: q.sample(st))
t_set = timeit(lambda: q.set_priorities(st, pri))
# algorithmic HBM bytes of one prioritized sample: the priority column is read by the two
# total passes and the final pass (3 x 4 B per row), cum and weight are written (8 B), the
# normalisation reads and writes weight (8 B); plus the gathered rows in and out
alg = cap * 28 + bs * D * 4 * 2
print(json.dumps({
    "workload": f"PrioritizedSamplingQueue cap={cap} D={D} draws={bs} insert={n_ins}",
    "insert_ms": round(t_ins * 1e3, 4), "insert_rows_per_s": round(n_ins / t_ins, 1),
    "sample_ms": round(t_smp * 1e3, 4), "draws_per_s": round(bs / t_smp, 1),
    "sample_algorithmic_bytes": alg, "sample_GBps": round(alg / t_smp / 1e9, 2),
    "sample_frac_of_8TBps": round(alg / t_smp / 8e12, 4),
    "set_priorities_ms": round(t_set * 1e3, 4)}))
