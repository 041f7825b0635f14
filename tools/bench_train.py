"""Timing of one full training step (variant B: G fwd -> D step -> G step) at a given batch, eager."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
import hifigan_modified as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Tm = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[3] if len(sys.argv) > 3 else "fp32"]
torch.manual_seed(0)
voc = H.ModifiedHiFiGANVocoder()
tr = H.VocoderTrainer(voc, device=torch.device("cuda"))
mel = torch.randn(B, 80, Tm, device="cuda").to(dt)
real = torch.randn(B, 1, Tm * 256, device="cuda").clamp(-1, 1).to(dt)
spk, emo = torch.randn(B, 192, device="cuda").to(dt), torch.randn(B, 384, device="cuda").to(dt)
for i in range(2):
    out = tr.train_step(mel, real, spk, emo, return_tensors=True)
torch.cuda.synchronize()
print({k: float(v) for k, v in out.items()})
t0 = time.perf_counter()
n = 3
for i in range(n):
    tr.train_step(mel, real, spk, emo, return_tensors=True)
torch.cuda.synchronize()
dtm = (time.perf_counter() - t0) / n
print(f"B={B} T={Tm*256} {dt}: {dtm*1e3:.1f} ms/step  -> {B*Tm*256/dtm:,.0f} samples/s")
