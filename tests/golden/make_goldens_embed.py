#!/usr/bin/env python3
"""Golden vectors of the conditioning producers: runs ONLY in the build container.

Imports the reference's `embedding_extractors.py` (ECAPA_TDNN, SE_Res2Block, SE_Module, Emotion2Vec), fills the parameters
from the shared recipe (embed_weights.py), runs eval-mode forwards on CPU in fp32 and writes inputs + expected outputs.

ECAPA_TDNN.forward does not run as written (embedding_extractors.py:84-87: cat(mean, std) has 6*hidden features, final_proj
takes 3*hidden).  The fixture is produced with `final_proj` replaced by `nn.Linear(6*hidden, embedding_dim)` - the only
change; every other line executed is the reference's own.

Usage: python tests/golden/make_goldens_embed.py
"""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch
import torch.nn as nn

from reference import embedding_extractors as R
from embed_weights import fill_state, state_checksum

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)
torch.backends.mkldnn.enabled = False


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_se():
    torch.manual_seed(1)
    for name, mod, x in [("embed_se_module_c64", R.SE_Module(64), torch.randn(2, 64, 50)),
                         ("embed_se_res2_c256_d3", R.SE_Res2Block(256, dilation=3), torch.randn(2, 256, 41))]:
        fill_state(mod, 0).train(False)
        with torch.no_grad():
            y = mod(x)
        _save(name, x=_np(x), y=_np(y), checksum=np.array(state_checksum(mod)))


def gen_ecapa():
    for name, hidden, T in [("embed_ecapa_h512_t100", 512, 100), ("embed_ecapa_h256_t37", 256, 37)]:
        m = R.ECAPA_TDNN(hidden_dim=hidden, num_speakers=16)
        m.final_proj = nn.Linear(6 * hidden, m.embedding_dim)
        fill_state(m, 0).train(False)
        torch.manual_seed(1)
        x = torch.randn(2, 80, T)
        taps = {}
        hooks = [m.se_res2_blocks[i].register_forward_hook(lambda _m, _i, o, i=i: taps.__setitem__(f"block{i}", _np(o)[:, ::8]))
                 for i in range(3)]
        hooks.append(m.attention.register_forward_hook(lambda _m, _i, o: taps.__setitem__("attention", _np(o)[:, ::16])))
        hooks.append(m.final_proj.register_forward_hook(lambda _m, i, o: taps.__setitem__("pooled", _np(i[0]))))
        with torch.no_grad():
            emb, logits = m(x)
        assert logits is None
        for h in hooks:
            h.remove()
        _save(name, x=_np(x), embedding=_np(emb), checksum=np.array(state_checksum(m)), **taps)


def gen_emotion():
    for name, hidden, T in [("embed_emotion_h512_t100", 512, 100), ("embed_emotion_h128_t37", 128, 37)]:
        m = R.Emotion2Vec(hidden_dim=hidden)
        fill_state(m, 0).train(False)
        torch.manual_seed(1)
        x = torch.randn(2, 80, T)
        taps = {}
        hooks = [m.feature_extractor.register_forward_hook(lambda _m, _i, o: taps.__setitem__("features", _np(o)[:, ::8])),
                 m.transformer.layers[0].register_forward_hook(lambda _m, _i, o: taps.__setitem__("layer0", _np(o)[:, :, ::8]))]
        with torch.no_grad():
            frame, utt, logits = m(x)
        assert logits is None
        for h in hooks:
            h.remove()
        _save(name, x=_np(x), frame=_np(frame), utterance=_np(utt), checksum=np.array(state_checksum(m)), **taps)


def gen_combined():
    m = R.EmbeddingExtractor()
    m.speaker_extractor.final_proj = nn.Linear(6 * 512, 192)
    fill_state(m, 0).train(False)
    torch.manual_seed(1)
    x = torch.randn(2, 80, 32)
    with torch.no_grad():
        spk, emo = m(x)
    _save("embed_extractor_t32", x=_np(x), speaker=_np(spk), emotion=_np(emo), checksum=np.array(state_checksum(m)))


if __name__ == "__main__":
    gen_se(); gen_ecapa(); gen_emotion(); gen_combined()
