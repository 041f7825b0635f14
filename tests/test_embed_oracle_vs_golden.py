"""Pins oracle/embed_oracle.py (conditioning producers) to golden vectors generated from the reference's
embedding_extractors.py (tests/golden/make_goldens_embed.py).  Also checks that the product modules expose the reference's
state_dict (keys + shapes: the recipe checksum only matches when they do)."""
import pytest
import torch

from conftest import load_golden
from embed_cases import CASES, build, sd_of
from oracle import embed_oracle as E
from oracle.vocoder_oracle import rel_l2

TOL = 1e-5
torch.set_num_threads(4)


def _t(g, k):
    return torch.from_numpy(g[k])


def test_se_module_and_res2_block():
    g = load_golden("embed_se_module_c64")
    sd = sd_of(build(*CASES["embed_se_module_c64"][:1], g, **CASES["embed_se_module_c64"][1]))
    assert rel_l2(E.se_module(_t(g, "x"), sd, ""), _t(g, "y")) < TOL
    g = load_golden("embed_se_res2_c256_d3")
    sd = sd_of(build("res2", g, **CASES["embed_se_res2_c256_d3"][1]))
    assert rel_l2(E.se_res2_block(_t(g, "x"), sd, "", 3), _t(g, "y")) < TOL


@pytest.mark.parametrize("name", ["embed_ecapa_h512_t100", "embed_ecapa_h256_t37"])
def test_ecapa(name):
    g = load_golden(name)
    sd = sd_of(build("ecapa", g, **CASES[name][1]))
    emb, taps = E.ecapa_tdnn(_t(g, "x"), sd, want_taps=True)
    for i in range(3):
        assert rel_l2(taps[f"block{i}"][:, ::8], _t(g, f"block{i}")) < TOL
    assert rel_l2(taps["attention"][:, ::16], _t(g, "attention")) < TOL
    assert rel_l2(taps["pooled"], _t(g, "pooled")) < TOL
    assert rel_l2(emb, _t(g, "embedding")) < TOL
    assert torch.allclose(emb.norm(dim=1), torch.ones(emb.shape[0]), atol=1e-5)


@pytest.mark.parametrize("name", ["embed_emotion_h512_t100", "embed_emotion_h128_t37"])
def test_emotion2vec(name):
    g = load_golden(name)
    sd = sd_of(build("emotion", g, **CASES[name][1]))
    frame, utt, taps = E.emotion2vec(_t(g, "x"), sd, want_taps=True)
    assert rel_l2(taps["features"][:, ::8], _t(g, "features")) < TOL
    assert rel_l2(taps["layer0"][:, :, ::8], _t(g, "layer0")) < TOL
    assert rel_l2(frame, _t(g, "frame")) < 5 * TOL
    assert rel_l2(utt, _t(g, "utterance")) < 5 * TOL


def test_combined_extractor():
    g = load_golden("embed_extractor_t32")
    sd = sd_of(build("extractor", g))
    spk, emo = E.embedding_extractor(_t(g, "x"), sd)
    assert rel_l2(spk, _t(g, "speaker")) < TOL and rel_l2(emo, _t(g, "emotion")) < 5 * TOL
    assert spk.shape == (2, 192) and emo.shape == (2, 256)
