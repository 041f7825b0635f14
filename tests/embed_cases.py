"""Shared construction of the embedding-extractor test cases: product modules with the recipe weights of
tests/golden/embed_weights.py (checked against the checksum stored in the fixture)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from embed_weights import fill_state, state_checksum      # noqa: E402


def build(kind, g=None, **kw):
    import hifigan_modified as H
    mod = {"se": H.SE_Module, "res2": H.SE_Res2Block, "ecapa": H.ECAPA_TDNN, "emotion": H.Emotion2Vec,
           "extractor": H.EmbeddingExtractor}[kind](**kw)
    fill_state(mod, 0).train(False)
    if g is not None:
        assert state_checksum(mod) == str(g["checksum"]), "state_dict keys/shapes/recipe differ from the fixture's"
    return mod


def sd_of(mod):
    return {k: v.detach().clone() for k, v in mod.state_dict().items()}


CASES = {
    "embed_se_module_c64": ("se", dict(channels=64)),
    "embed_se_res2_c256_d3": ("res2", dict(channels=256, dilation=3)),
    "embed_ecapa_h512_t100": ("ecapa", dict(hidden_dim=512, num_speakers=16)),
    "embed_ecapa_h256_t37": ("ecapa", dict(hidden_dim=256, num_speakers=16)),
    "embed_emotion_h512_t100": ("emotion", dict(hidden_dim=512)),
    "embed_emotion_h128_t37": ("emotion", dict(hidden_dim=128)),
    "embed_extractor_t32": ("extractor", dict()),
}
