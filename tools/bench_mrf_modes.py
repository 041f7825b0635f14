"""C2 forward (B=32 x 32 frames) in the mixed storage mode with the MRF chain's operand mode switched: parity of the FULL batch against
the oracle and HIP-graph replay time.  One process per kernel geometry (the library reads MV_MRF_W16_NW / MV_MRF_W16_NTW once):
    python tools/bench_mrf_modes.py [through] [w16: 0|1] [48k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
import hifigan_modified as H
from hifigan_modified.graphs import GraphedVocoder
from oracle import vocoder_oracle as O

through = sys.argv[1] if len(sys.argv) > 1 else "up1"
through = None if through == "none" else through
w16 = len(sys.argv) > 2 and sys.argv[2] == "1"
k48 = len(sys.argv) > 3 and sys.argv[3] == "48k"
kw, nmel, Tm = (dict(mel_channels=128, upsample_factors=[8, 8, 4, 2]), 128, 16) if k48 else ({}, 80, 32)
torch.manual_seed(0)
g0 = H.ModifiedHiFiGANGenerator(**kw)
sd = {k: v.detach().clone() for k, v in g0.state_dict().items()}
B = 32
NREF = int(os.environ.get("NREF", "32"))
torch.manual_seed(1)
mel, spk, emo = torch.randn(B, nmel, Tm), torch.randn(B, 192), torch.randn(B, 384)
torch.set_num_threads(16)
with torch.no_grad():
    ref = O.generator_forward(mel[:NREF], sd, "", spk[:NREF], emo[:NREF], upsample_factors=tuple(g0.upsample_factors))
g = H.ModifiedHiFiGANGenerator(**kw); g.load_state_dict(sd); g = g.cuda().train(False)
g.set_mixed_precision(through, mrf_weights="fp16" if w16 else None)
m, s, e = mel.cuda(), spk.cuda(), emo.cuda()
with torch.no_grad():
    w = g(m, s, e).float().cpu()
    w2 = g(m, s, e).float().cpu()
err = O.rel_l2(w[:NREF], ref)
per = [O.rel_l2(w[i:i + 1], ref[i:i + 1]) for i in range(NREF)]
gv = GraphedVocoder(g, m, s, e)
for _ in range(300): gv.replay()
torch.cuda.synchronize(); t = time.perf_counter()
n = 300
for _ in range(n): gv.replay()
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
print(f"through={through} w16={int(w16)} nw={os.environ.get('MV_MRF_W16_NW','8')} ntw={os.environ.get('MV_MRF_W16_NTW','2')} {'48k' if k48 else '22k'}: "
      f"rel-L2 {err:.3e} (per-clip max {max(per):.2e}) deterministic {torch.equal(w, w2)}  {ms:.4f} ms  {B*Tm/ms/1e3:.2f} M frames/s", flush=True)
