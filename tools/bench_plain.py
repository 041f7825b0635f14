"""BASELINE configs[0] (plain HiFi-GAN V3, B=1 x 344 frames): eager and captured forward time.  python tools/bench_plain.py [fp32|fp16|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
from hifigan_modified.plain_hifigan import PlainHiFiGANGenerator
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
torch.manual_seed(0)
g = PlainHiFiGANGenerator().cuda().to(dt).train(False)
mel = torch.randn(1, 80, 344, device="cuda").to(dt)
for _ in range(3):
    g(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g(mel)
torch.cuda.synchronize()
print(f"eager  {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
rp = g.graphed(mel)
for _ in range(5):
    rp()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    rp()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 50
print(f"graph  {el * 1e3:.3f} ms -> {344 / el:,.0f} frames/s")
