"""A/B of the dconv variants at the big 3x3 discriminator shapes (one process per setting: the env switch is read once)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
from hifigan_modified import ops, _native as N
dt = torch.bfloat16
B, T = 64, 8192
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0
for P in [int(v) for v in os.environ.get("PS", "2,3,5,7,11").split(",")]:
    for kind, cin, cout in (("fwd", 128, 256), ("dgrad", 256, 128), ("fwd", 64, 128), ("dgrad", 128, 64)):
        Hh, W = P, T // P
        torch.manual_seed(0)
        x = torch.randn(B, Hh, W, cin, device="cuda").to(dt)
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        pk = ops.dconv_pack(w, dt, 0)
        sv = torch.randn(B, Hh, W, cout, device="cuda").to(dt) if kind == "dgrad" else None
        y = ops.dconv_cl(x, pk, None, cout, 3, 3, 1, N.ACT_NONE, 0.1, sv)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).float()[:2], w.to(dt).float(), padding=1).permute(0, 2, 3, 1)
        if sv is not None: ref = torch.where(sv[:2].float() >= 0, ref, ref * 0.1)
        err = ((y[:2].float() - ref).norm() / ref.norm()).item()
        us = timeit(lambda: ops.dconv_cl(x, pk, None, cout, 3, 3, 1, N.ACT_NONE, 0.1, sv))
        fl = 2.0 * B * Hh * W * cin * cout * 9
        tot += us
        print(f"{kind:5s} P={P:2d} {cin:3d}->{cout:3d} {us:8.1f} us {fl / us / 1e6:7.1f} TF  err {err:.2e}", flush=True)
print(f"total {tot/1e3:.2f} ms")
