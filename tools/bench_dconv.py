"""Channels-last MFMA conv (mv_dconv_cl_fwd) at the discriminator training shapes: forward and data-gradient operators."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
from hifigan_modified import ops, _native as N

dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 8192


def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = []
for (P, kh, kw) in ((2, 3, 3), (11, 3, 3), (1, 1, 15)):
    for cin, cout in ((32, 64), (64, 128), (128, 256), (256, 32)):
        cases.append(("fwd", P, T // P, cin, cout, kh, kw))
    for cin, cout in ((32, 256), (256, 128), (128, 64), (64, 32)):
        cases.append(("dgrad", P, T // P, cin, cout, kh, kw))
tot = 0.0
for (kind, Hh, W, cin, cout, kh, kw) in cases:
    torch.manual_seed(0)
    x = torch.randn(B, Hh, W, cin, device="cuda").to(dt)
    w = torch.randn(cout, cin, kh, kw, device="cuda") / (cin * kh * kw) ** 0.5
    pk = ops.dconv_pack(w, dt, 0)
    sv = torch.randn(B, Hh, W, cout, device="cuda").to(dt) if kind == "dgrad" else None
    fn = lambda: ops.dconv_cl(x, pk, None, cout, kh, kw, 1, N.ACT_NONE, 0.1, sv)
    us = timeit(fn)
    fl = 2.0 * B * Hh * W * cin * cout * kh * kw
    byts = (x.numel() + B * Hh * W * cout * (2 if sv is not None else 1)) * 2
    tot += us
    print(f"{kind:5s} H={Hh:2d} W={W:5d} {cin:3d}->{cout:3d} {kh}x{kw:<2d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {byts / us / 1e3:7.0f} GB/s", flush=True)
print(f"total {tot / 1e3:.2f} ms")
