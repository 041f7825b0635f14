"""Pins the CPU oracle (oracle/vocoder_oracle.py) to golden vectors generated from the reference
(tests/golden/make_goldens.py).  fp32 restatement vs fp32 reference: rel-L2 <= 1e-5;
the MPD fold index map must be bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import vocoder_oracle as O

TOL = 1e-5
torch.set_num_threads(4)


def sd_of(g, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def t(g, key, grad=False):
    x = torch.from_numpy(g[key]).clone()
    return x.requires_grad_(True) if grad else x


def check_grads(g, sd_req, out, r, inputs, tol=5e-5):
    """Back-propagate the golden cotangent through the oracle and compare every stored gradient."""
    (out * r).sum().backward()
    for name, x in inputs.items():
        key = "gx." + name
        if key in g:
            assert O.rel_l2(x.grad, torch.from_numpy(g[key])) < tol, key
    nograd = set(g["nograd"].tolist()) if "nograd" in g else set()
    n_checked = 0
    for k, p in sd_req.items():
        key = "grad." + k
        if key in g:
            assert p.grad is not None, f"{k}: oracle produced no grad"
            ref = torch.from_numpy(g[key])
            if ref.abs().max() == 0:
                assert p.grad.abs().max() < 1e-6, k
            else:
                assert O.rel_l2(p.grad, ref) < tol, (k, O.rel_l2(p.grad, ref))
            n_checked += 1
        elif k in nograd:
            assert p.grad is None, f"{k}: reference has grad None (unused parameter), oracle has a grad"
    assert n_checked > 0


def req(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}


ODCONV = [("odconv1d_c16_o8_k3_d2", dict(padding=2, dilation=2)),
          ("odconv1d_c80_o32_k7", dict(padding=3)),
          ("odconv1d_c8_o8_k5_s2", dict(padding=2, stride=2))]


@pytest.mark.parametrize("name,kw", ODCONV)
@pytest.mark.parametrize("form", ["kloop", "aggregate"])
def test_odconv1d(name, kw, form):
    g = load_golden(name)
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    alpha = O.odconv_attention(x, sd["kernel_attention.1.weight"], sd["kernel_attention.1.bias"])
    assert O.rel_l2(alpha, torch.from_numpy(g["alpha"])) < TOL
    y = O.odconv1d(x, sd, "", form=form, **kw)
    assert y.shape == g["y"].shape
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x})


ODCONVT = [("odconvT_c16_o8_k16_s8", dict(stride=8, padding=4)),
           ("odconvT_c8_o8_k4_s2", dict(stride=2, padding=1)),
           ("odconvT_c8_o8_k8_s4", dict(stride=4, padding=2)),
           ("odconvT_c8_o4_k6_s3_op1", dict(stride=3, padding=1, output_padding=1))]


@pytest.mark.parametrize("name,kw", ODCONVT)
@pytest.mark.parametrize("form", ["kloop", "aggregate"])
def test_odconv_transpose1d(name, kw, form):
    g = load_golden(name)
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    y = O.odconv_transpose1d(x, sd, "", form=form, **kw)
    assert y.shape == g["y"].shape
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x})


def test_odconv_unused_attention_params_have_no_grad():
    """odconv.py:42-62 builds spatial/in/out-channel attention nets that forward never calls."""
    g = load_golden("odconv1d_c16_o8_k3_d2")
    nograd = set(g["nograd"].tolist())
    assert {"spatial_attention.1.weight", "in_channel_attention.1.weight", "in_channel_attention.3.weight",
            "out_channel_attention.1.weight", "out_channel_attention.3.weight"} <= nograd


@pytest.mark.parametrize("name,d", [("grc_64_20_d1", 1), ("grc_64_20_d3", 3), ("grc_64_20_d5", 5),
                                    ("grc_16_16_d1_r4", 1)])
def test_grc_lora_block(name, d):
    g = load_golden(name)
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    y = O.grc_lora_block(x, sd, "", d)
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x})


@pytest.mark.parametrize("name,dil", [("mrf_64_64", (1, 3, 5)), ("mrf_32_32_g2", (1, 2))])
def test_mrf_block(name, dil):
    g = load_golden(name)
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    y = O.mrf_block(x, sd, "", dil)
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x})


@pytest.mark.parametrize("name", ["film_64_64_both", "film_64_64_spk", "film_64_64_emo", "film_64_576_both",
                                  "film_16_600_trunc"])
def test_film(name):
    g = load_golden(name)
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    spk = t(g, "x.spk", grad=True) if "x.spk" in g else None
    emo = t(g, "x.emo", grad=True) if "x.emo" in g else None
    y = O.film(x, sd, "", spk, emo)
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    ins = {"x": x}
    if spk is not None:
        ins["spk"] = spk
    (y * t(g, "r")).sum().backward()
    for nme, v in ins.items():
        assert O.rel_l2(v.grad, torch.from_numpy(g["gx." + nme])) < 5e-5
    for k, p in sd.items():
        assert O.rel_l2(p.grad, torch.from_numpy(g["grad." + k])) < 5e-5
    if emo is not None and "gx.emo" in g:
        ref = torch.from_numpy(g["gx.emo"])
        got = emo.grad if emo.grad is not None else torch.zeros_like(ref)
        assert (got - ref).abs().max() < 1e-5


def test_film_none_is_identity():
    g = load_golden("film_64_64_none")
    x = t(g, "x")
    assert torch.equal(O.film(x, {}, "", None, None), torch.from_numpy(g["y"]))


@pytest.mark.parametrize("P", [2, 3, 5, 7, 11])
@pytest.mark.parametrize("T", [1000, 8192])
def test_mpd_fold_index_bit_exact(P, T):
    g = load_golden(f"mpd_index_P{P}_T{T}")
    idx = O.mpd_fold_index(T, P)
    assert idx.dtype == np.int64 and idx.shape == g["index"].shape
    assert np.array_equal(idx, g["index"])
    # and the tensor fold reproduces it on an arange signal
    sig = torch.arange(1, T + 1, dtype=torch.float32).view(1, 1, T)
    folded = O.mpd_fold(sig, P)[0, 0].to(torch.int64).numpy() - 1
    assert np.array_equal(folded, g["index"])


def test_disc2d():
    g = load_golden("disc2d_P3")
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    y = O.disc2d(x, sd, "", 3)
    assert y.shape == g["y"].shape
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x})


def test_disc1d():
    g = load_golden("disc1d_s2")
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    y = O.disc1d(x, sd, "", 2)
    assert y.shape == g["y"].shape
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x})


def test_grouped_residual_conv1d():
    g = load_golden("grouped_residual_64_k3_d3")
    sd = req(sd_of(g))
    x = t(g, "x.x", grad=True)
    y = O.grouped_residual_conv1d(x, sd, "", dilation=3, groups=4)
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x})


def test_film2():
    g = load_golden("film2_448_64")
    sd = req(sd_of(g))
    x, spk, emo = t(g, "x.x", True), t(g, "x.spk", True), t(g, "x.emo", True)
    y = O.film2(x, sd, "", spk, emo)
    assert O.rel_l2(y, torch.from_numpy(g["y"])) < TOL
    check_grads(g, sd, y, t(g, "r"), {"x": x, "spk": spk, "emo": emo})


@pytest.mark.parametrize("form", ["kloop", "aggregate"])
def test_generator_small_every_stage(form):
    g = load_golden("generator_small")
    sd = sd_of(g)
    mel, spk, emo = t(g, "x.mel"), t(g, "x.spk"), t(g, "x.emo")
    with torch.no_grad():
        st = O.generator_forward(mel, sd, "", spk, emo, hidden_channels=64, upsample_factors=(4, 2),
                                 form=form, return_stages=True)
        nc = O.generator_forward(mel, sd, "", hidden_channels=64, upsample_factors=(4, 2), form=form)
    for k in ("input_proj", "film", "up0", "up1", "mrf0", "mrf1", "mrf2", "output_proj", "wave"):
        assert O.rel_l2(st[k], torch.from_numpy(g["stage." + k])) < 2e-5, k
    assert st["wave"].shape == (2, 1, 64)
    assert O.rel_l2(nc, torch.from_numpy(g["wave_nocond"])) < 2e-5


def test_generator_small_gradients():
    g = load_golden("generator_small")
    sd = req(sd_of(g))
    mel, spk, emo = t(g, "x.mel", True), t(g, "x.spk", True), t(g, "x.emo")
    y = O.generator_forward(mel, sd, "", spk, emo, hidden_channels=64, upsample_factors=(4, 2))
    # whole-net fp32 gradients: scalar grads (lora_scaling) are cancelling sums -> 3e-4
    check_grads(g, sd, y, t(g, "r"), {"mel": mel, "spk": spk}, tol=3e-4)
    nograd = set(g["nograd"].tolist())
    # the never-used ODConv attention nets stay grad-less in the whole generator
    assert any("spatial_attention" in k for k in nograd) and len(nograd) >= 30


def test_lsgan_and_hinge_losses():
    g = load_golden("disc_system_losses")
    outs = {key: [torch.from_numpy(g[f"out.{key}.{i}"]) for i in range(n)]
            for key, n in (("mpd_real", 5), ("mpd_fake", 5), ("msd_real", 3), ("msd_fake", 3))}
    d = O.lsgan_discriminator_losses(outs)
    for k, v in d.items():
        assert abs(float(v) - float(g["dloss." + k])) < 1e-5 * max(1.0, abs(float(g["dloss." + k]))), k
    gl = O.lsgan_generator_losses(outs, torch.from_numpy(g["mel"]), torch.from_numpy(g["gen_mel"]))
    for k, v in gl.items():
        assert abs(float(v) - float(g["gloss." + k])) < 1e-5 * max(1.0, abs(float(g["gloss." + k]))), k
    # hinge (conditioned_hifigan.py:262-265): closed form on the golden outputs
    h = O.hinge_generator_loss(outs["mpd_fake"])
    ref = sum(float(np.maximum(0.0, 1.0 - g[f"out.mpd_fake.{i}"]).mean()) for i in range(5))
    assert abs(float(h) - ref) < 1e-6
    assert g["out.mpd_real.1"].shape == (2, 1, 3, 334) and g["out.msd_real.2"].shape == (2, 1, 250)


def test_mel_filterbank_and_stft_against_torch():
    """The mel/STFT loss has nothing to pin to in the reference (placeholder) - PARITY UNPINNED.
    Self-consistency: the oracle's explicit-DFT magnitude equals torch.stft's."""
    torch.manual_seed(1)
    wave = torch.randn(2, 1, 2048, dtype=torch.float64).clamp(-1, 1)
    m = O.mel_spectrogram(wave)
    assert m.shape == (2, 80, 8)
    padn = (1024 - 256) // 2
    w = torch.nn.functional.pad(wave, (padn, padn), mode="reflect")[:, 0]
    spec = torch.stft(w, 1024, hop_length=256, win_length=1024,
                      window=torch.from_numpy(O.hann_window(1024)), center=False, return_complex=True)
    mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9)
    fb = torch.from_numpy(O.mel_filterbank())
    ref = torch.log(torch.clamp(fb @ mag, min=1e-5))
    assert O.rel_l2(m, ref) < 1e-9
    fbn = O.mel_filterbank()
    assert fbn.shape == (80, 513) and (fbn >= 0).all() and (fbn.sum(axis=1) > 0).all()
