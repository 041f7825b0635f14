"""Per-kernel timing at the C2 shapes (HIP events on the launch stream), for tuning."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
import hifigan_modified as H
from hifigan_modified import functional as Fn, ops, _native as N
from hifigan_modified.fused import generator_fused_for

dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
B, Tm = 32, 32
torch.manual_seed(0)
gen = H.ModifiedHiFiGANGenerator().cuda().to(dt).train(False)
fz = generator_fused_for(gen)
mel = torch.randn(B, 80, Tm, device="cuda").to(dt)
spk = torch.randn(B, 192, device="cuda").to(dt); emo = torch.randn(B, 384, device="cuda").to(dt)
es = torch.tensor([], dtype=dt).element_size()

def timeit(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us

with torch.no_grad():
    st = gen(mel, spk, emo, return_stages=True)
    att = gen.input_proj.kernel_attention[1]
    alpha0 = ops.odconv_attn(mel, att.weight.view(4, 80), att.bias)
    x0 = ops.nct_to_ntc(mel)
    t = timeit(lambda: fz.inp.forward_cl(x0, Fn._cache, alpha=alpha0))
    print(f"input_proj      {t:8.1f} us")
    prev = ops.nct_to_ntc(st["film"])
    for i, u in enumerate(fz.ups):
        pooled = prev.float().sum(dim=1).contiguous()
        pout = torch.empty(B, u.pool_floats(B, prev.shape[1], dt, N.ACT_LRELU), device="cuda")
        t = timeit(lambda: u.forward_cl(prev, Fn._cache, pooled_in=pooled, pooled_out=pout, act=N.ACT_LRELU))
        y = u.forward_cl(prev, Fn._cache, pooled_in=pooled, act=N.ACT_LRELU)
        byts = (prev.numel() + y.numel()) * es + u.mod.kernels.numel() * es
        print(f"ups{i} {tuple(prev.shape)}->{tuple(y.shape)}  {t:8.1f} us   {byts / t / 1e3:8.1f} GB/s (in+out+weights)")
        prev = y
    m = fz.mrfs[0]
    t = timeit(lambda: m.forward_cl(prev))
    byts = 2 * prev.numel() * es
    print(f"mrf block (3 passes) {t:8.1f} us   alg {byts / t / 1e3:8.1f} GB/s   actual(3R+1W) {2 * byts / t / 1e3:8.1f} GB/s")
    wt, bias = fz.out_weights(mel.device)
    wave = torch.empty(B, 1, prev.shape[1], device="cuda", dtype=dt)
    def outc():
        N.call("mv_conv_out_act_packed_cl", ctypes.c_void_p(prev.data_ptr()), ctypes.c_void_p(wt.data_ptr()), bias,
               ctypes.c_void_p(wave.data_ptr()), B, prev.shape[1], 64, 11, 5, N.ACT_TANH, ops._dt(prev), ops._stream())
    t = timeit(outc)
    print(f"out conv+tanh   {t:8.1f} us   {prev.numel() * es / t / 1e3:8.1f} GB/s")
    t = timeit(lambda: gen(mel, spk, emo))
    print(f"generator eager {t:8.1f} us")
    from hifigan_modified.graphs import GraphedVocoder
    gv = GraphedVocoder(gen, mel, spk, emo)
    t = timeit(gv.replay)
    print(f"generator graph {t:8.1f} us   -> {B * Tm / t * 1e6:,.0f} frames/s")
