"""Forward-kernel time vs network shape and batch (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import alphazero_amd as mctx

dev = torch.device("cuda")
def timeit(fn, n=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

g = torch.Generator(device=dev); g.manual_seed(0)
for B in (16, 512, 4096):
    for pol_h, val_h in [((32,)*4, (256,)*5), ((32,)*4, (256,)*1), ((32,)*4, (256,)*3), ((32,)*4, (256,)*7),
                         ((32,), (256,)*5), ((32,)*4, (128,)*5), ((32,)*4, (512,)*5), ((32,)*4, (64,)*5)]:
        pol = mctx.init_mlp((4,) + pol_h + (2,), generator=g, device=dev)
        val = mctx.init_mlp((4,) + val_h + (1,), generator=g, device=dev)
        model = mctx.FusedModel("CartPole-v1", pol, val)
        obs = torch.randn(B, 4, device=dev)
        logits = torch.empty((B, 2), device=dev); value = torch.empty((B, 1), device=dev)
        from alphazero_amd import _native as nat
        def run():
            nat.lib().az_forward(model.native, B, nat.ptr(obs), nat.ptr(logits), nat.ptr(value), nat.stream_ptr())
        print(f"B={B:5d} policy {len(pol_h)}x{pol_h[0]} value {len(val_h)}x{val_h[0]}: {timeit(run):7.1f} us")
