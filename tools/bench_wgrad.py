"""Discriminator weight-gradient kernel at the training shapes: time + check against torch's fp32 conv wgrad."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
from hifigan_modified import ops, _native as N

_P = lambda t: ctypes.c_void_p(t.data_ptr())
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
for P in (2, 3, 5, 7, 11):
    for cin, cout in ((32, 64), (64, 128), (128, 256)):
        cases.append((P, T // P, cin, cout, 3, 3))
for cin, cout in ((32, 64), (64, 128), (128, 256)):
    for W in (T, T // 2, T // 4):
        cases.append((1, W, cin, cout, 1, 15))

tot = 0.0
for (Hh, W, cin, cout, kh, kw) in cases:
    torch.manual_seed(0)
    x = torch.randn(B, Hh, W, cin, device="cuda").to(dt)
    g = torch.randn(B, Hh, W, cout, device="cuda").to(dt)
    gw = torch.empty(cout, cin, kh, kw, device="cuda")
    ws = torch.empty(kh * kw, cout, cin, device="cuda")
    fn = lambda: N.call("mv_dconv_wgrad_cl", _P(x), _P(g), _P(gw), None, _P(ws), B, Hh, W, cin, cout, kh, kw, 1, ops._dt(x), ops._stream())
    us = timeit(fn)
    fl = 2.0 * B * Hh * W * cin * cout * kh * kw
    # reference on a batch slice (fp32)
    nb = min(B, 2)
    xr = x[:nb].float().permute(0, 3, 1, 2).contiguous()
    gr = g[:nb].float().permute(0, 3, 1, 2).contiguous()
    ref = torch.nn.grad.conv2d_weight(xr, (cout, cin, kh, kw), gr, padding=(kh // 2, kw // 2))
    gw2 = torch.empty_like(gw)
    N.call("mv_dconv_wgrad_cl", _P(x[:nb].contiguous()), _P(g[:nb].contiguous()), _P(gw2), None, _P(ws), nb, Hh, W, cin, cout, kh, kw, 1,
           ops._dt(x), ops._stream())
    err = ((gw2 - ref).norm() / ref.norm()).item()
    tot += us
    print(f"H={Hh:2d} W={W:5d} {cin:3d}->{cout:3d} {kh}x{kw:<2d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s   rel err {err:.2e}", flush=True)
print(f"total {tot / 1e3:.2f} ms")
