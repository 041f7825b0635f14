#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

It imports the reference's *runnable* hot-path modules (odconv.py,
grc_lora.py, discriminators.py, generator.py{GroupedResidualConv1D,FeatureWiseLinearModulation}
and the unbound loss methods of complete_vocoder.ModifiedHiFiGANVocoder), seeds them, runs them
on CPU in fp32 and writes small ``.npz`` fixtures next to this file.  The fixtures are data
(inputs, parameters, expected outputs, expected gradients) - no reference source travels.

Seeds: parameters ``torch.manual_seed(0)``, lazy ``residual_proj`` materialisation
``torch.manual_seed(2)``, inputs/cotangents ``torch.manual_seed(1)``.

Usage:  python tests/golden/make_goldens.py          (writes tests/golden/*.npz)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
sys.path.insert(0, "/root")

import numpy as np
import torch
import torch.nn as nn

from reference.hifigan_modified import odconv as R_od
from reference.hifigan_modified import grc_lora as R_grc
from reference.hifigan_modified import discriminators as R_disc
from reference.hifigan_modified import generator as R_gen
from reference.hifigan_modified import complete_vocoder as R_cv

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)
torch.backends.mkldnn.enabled = False  # plain fp32 kernels, deterministic summation order


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays)")


def _state(mod, prefix="sd."):
    return {prefix + k: _np(v) for k, v in mod.state_dict().items()}


def _grads(mod, prefix="grad.", keep=None):
    """Parameter grads; parameters whose grad is None are listed in 'nograd'.
    ``keep``: optional predicate on the parameter name (large stacks keep a subset)."""
    out, nograd = {}, []
    for k, p in mod.named_parameters():
        if p.grad is None:
            nograd.append(k)
        elif keep is None or keep(k):
            out[prefix + k] = _np(p.grad)
    out["nograd"] = np.array(nograd, dtype=np.str_)
    return out


def module_case(name, mod, inputs, warm=False, extra=None, input_names=None, keep_grad=None):
    """Forward + gradient golden for one module call: y = mod(*inputs); (y*r).sum().backward()."""
    mod.train(False)
    if warm:  # materialise lazily created parameters (grc_lora.py:62-66) under seed 2
        torch.manual_seed(2)
        with torch.no_grad():
            mod(*[i if i is None else i.detach() for i in inputs])
    arrays = _state(mod)
    ins = []
    for i, t in enumerate(inputs):
        nm = input_names[i] if input_names else f"in{i}"
        if t is None:
            ins.append(None)
            continue
        t = t.clone().requires_grad_(True)
        ins.append(t)
        arrays["x." + nm] = _np(t)
    for p in mod.parameters():
        p.grad = None
    y = mod(*ins)
    torch.manual_seed(11)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    arrays["y"] = _np(y)
    arrays["r"] = _np(r)
    for i, t in enumerate(ins):
        if t is not None and t.grad is not None:
            nm = input_names[i] if input_names else f"in{i}"
            arrays["gx." + nm] = _np(t.grad)
    arrays.update(_grads(mod, keep=keep_grad))
    if extra:
        arrays.update(extra)
    _save(name, **arrays)


def randomise_bias(m):
    # zero-initialised ODConv biases (odconv.py:71) would hide bias-mixing bugs
    with torch.no_grad():
        m.bias.copy_(torch.randn_like(m.bias) * 0.5)


# ------------------------------------------------------------------ (1) ODConv1d
def gen_odconv1d():
    for tag, args, kw, C, T in [
        ("odconv1d_c16_o8_k3_d2", (16, 8, 3), dict(padding=2, dilation=2), 16, 24),
        ("odconv1d_c80_o32_k7", (80, 32, 7), dict(padding=3), 80, 24),
        ("odconv1d_c8_o8_k5_s2", (8, 8, 5), dict(padding=2, stride=2), 8, 25),
    ]:
        torch.manual_seed(0)
        m = R_od.ODConv1d(*args, **kw)
        randomise_bias(m)
        torch.manual_seed(1)
        x = torch.randn(2, C, T)
        with torch.no_grad():
            alpha = m.kernel_attention(x)  # [B,K,1]
        module_case(tag, m, [x], extra={"alpha": _np(alpha[:, :, 0])}, input_names=["x"])


# ------------------------------------------------------------------ (2) ODConvTranspose1d
def gen_odconvT():
    for tag, args, kw, C, T in [
        ("odconvT_c16_o8_k16_s8", (16, 8, 16), dict(stride=8, padding=4), 16, 6),
        ("odconvT_c8_o8_k4_s2", (8, 8, 4), dict(stride=2, padding=1), 8, 11),
        ("odconvT_c8_o8_k8_s4", (8, 8, 8), dict(stride=4, padding=2), 8, 7),
        ("odconvT_c8_o4_k6_s3_op1", (8, 4, 6), dict(stride=3, padding=1, output_padding=1), 8, 10),
    ]:
        torch.manual_seed(0)
        m = R_od.ODConvTranspose1d(*args, **kw)
        randomise_bias(m)
        torch.manual_seed(1)
        x = torch.randn(2, C, T)
        with torch.no_grad():
            alpha = m.kernel_attention(x)
        module_case(tag, m, [x], extra={"alpha": _np(alpha[:, :, 0])}, input_names=["x"])


# ------------------------------------------------------------------ (3) GRC_LoRA_Block
def gen_grc():
    for tag, args, C, T in [
        ("grc_64_20_d1", (64, 20, 3, 1, 16), 64, 40),
        ("grc_64_20_d3", (64, 20, 3, 3, 16), 64, 40),
        ("grc_64_20_d5", (64, 20, 3, 5, 16), 64, 40),
        ("grc_16_16_d1_r4", (16, 16, 3, 1, 4), 16, 33),
    ]:
        torch.manual_seed(0)
        m = R_grc.GRC_LoRA_Block(*args)
        torch.manual_seed(1)
        x = torch.randn(2, C, T)
        module_case(tag, m, [x], warm=True, input_names=["x"])


# ------------------------------------------------------------------ (4) MultiReceptiveFieldBlock (eval)
def gen_mrf():
    for tag, args, kw, C, T in [
        ("mrf_64_64", (64, 64), {}, 64, 48),
        ("mrf_32_32_g2", (32, 32), dict(dilations=[1, 2], groups=2, r=4), 32, 21),
    ]:
        torch.manual_seed(0)
        m = R_grc.MultiReceptiveFieldBlock(*args, **kw)
        torch.manual_seed(1)
        x = torch.randn(2, C, T)
        module_case(tag, m, [x], warm=True, input_names=["x"])


# ------------------------------------------------------------------ (5) FiLMLayer
def gen_film():
    torch.manual_seed(0)
    m64 = R_grc.FiLMLayer(64, 64)
    m576 = R_grc.FiLMLayer(64, 576)
    m8 = R_grc.FiLMLayer(16, 600)
    torch.manual_seed(1)
    x512 = torch.randn(2, 512, 5)
    x8 = torch.randn(2, 8, 5)
    spk = torch.randn(2, 192)
    emo = torch.randn(2, 384)
    module_case("film_64_64_both", m64, [x512, spk, emo], input_names=["x", "spk", "emo"])
    module_case("film_64_64_spk", m64, [x512, spk, None], input_names=["x", "spk", "emo"])
    module_case("film_64_64_emo", m64, [x512, None, emo], input_names=["x", "spk", "emo"])
    module_case("film_64_576_both", m576, [x512, spk, emo], input_names=["x", "spk", "emo"])
    # C < feature_dim (truncate gamma/beta) and cond shorter than condition_dim (zero pad)
    module_case("film_16_600_trunc", m8, [x8, spk, emo], input_names=["x", "spk", "emo"])
    with torch.no_grad():
        y = m64(x512, None, None)
    _save("film_64_64_none", x=_np(x512), y=_np(y))


# ------------------------------------------------------------------ (6) Discriminator2D / 1D
def _disc_keep(k):
    # parameter-gradient goldens for the first two and the last conv only (the 128->256 layer alone is 1.2 MB)
    return any(k.startswith(f"conv_layers.{i}.") for i in (0, 2, 8)) or k.endswith(".bias")


def gen_disc():
    for P in [2, 3, 5, 7, 11]:
        for T in [1000, 8192]:
            # bit-exact index map of pad + view (discriminators.py:72-79): run the reference's own
            # pad/view on an arange signal (exactly representable in fp32) with identity "convs"
            d = R_disc.Discriminator2D(P)
            d.conv_layers = nn.Identity()
            sig = torch.arange(1, T + 1, dtype=torch.float32).view(1, 1, T)  # 0 marks padding
            with torch.no_grad():
                v = d(sig)
            idx = v.to(torch.int64)[0, 0] - 1  # -1 = zero padding, else source index
            _save(f"mpd_index_P{P}_T{T}", index=_np(idx))
    torch.manual_seed(1)
    x = torch.randn(2, 1, 1000)
    for P in [2, 3, 5, 7, 11]:
        torch.manual_seed(0)
        d = R_disc.Discriminator2D(P)
        # the full conv stack is 390 K parameters per period; keep only P=3 with parameters,
        # the others are covered by the whole-system fixture below via seeds
        if P == 3:
            module_case(f"disc2d_P{P}", d, [x], input_names=["x"], keep_grad=_disc_keep)
    for s in [2]:
        torch.manual_seed(0)
        d = R_disc.Discriminator1D(s)
        module_case(f"disc1d_s{s}", d, [x], input_names=["x"], keep_grad=_disc_keep)


# ------------------------------------------------------------------ (7)+(8) whole discriminator system + losses
def gen_system_losses():
    torch.manual_seed(0)
    D = R_disc.HiFiGANDiscriminators()
    torch.manual_seed(1)
    real = torch.randn(2, 1, 1000).clamp(-1, 1)
    fake = torch.tanh(torch.randn(2, 1, 1000))
    mel = torch.randn(2, 80, 4)
    gen_mel = torch.randn(2, 80, 4)
    D.train(False)
    with torch.no_grad():
        out = D(real, fake)
    arrays = {"real": _np(real), "fake": _np(fake), "mel": _np(mel), "gen_mel": _np(gen_mel)}
    # parameters: seeds only (3.9 M floats would be 15 MB) + per-tensor checksums
    for k, v in D.state_dict().items():
        arrays["chk." + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    for key in ("mpd_real", "mpd_fake", "msd_real", "msd_fake"):
        for i, t in enumerate(out[key]):
            arrays[f"out.{key}.{i}"] = _np(t)
    stub = types.SimpleNamespace(discriminators=D, fm_weight=10.0, mel_weight=45.0)
    f = fake.clone().requires_grad_(True)
    g = R_cv.ModifiedHiFiGANVocoder.compute_generator_losses(stub, real, f, mel, gen_mel)
    g["total_loss"].backward()
    for k, v in g.items():
        arrays["gloss." + k] = np.array(v.item())
    arrays["gloss.dfake"] = _np(f.grad)
    for p in D.parameters():
        p.grad = None
    f2 = fake.clone().requires_grad_(True)
    dl = R_cv.ModifiedHiFiGANVocoder.compute_discriminator_losses(stub, real, f2)
    dl["total_loss"].backward()
    for k, v in dl.items():
        arrays["dloss." + k] = np.array(v.item())
    arrays["dloss.dfake"] = _np(f2.grad)
    # a few parameter-gradient probes of the D loss (first/last layer of first MPD and MSD nets)
    for k, p in D.named_parameters():
        if k in ("mpd.discriminators.0.conv_layers.0.weight", "mpd.discriminators.0.conv_layers.8.weight",
                 "msd.discriminators.0.conv_layers.0.weight", "msd.discriminators.2.conv_layers.8.bias",
                 "mpd.discriminators.4.conv_layers.2.bias"):
            arrays["dloss.grad." + k] = _np(p.grad)
    # hinge variant quantities (conditioned_hifigan.py:262-265) evaluated with torch on the same outputs
    _save("disc_system_losses", **arrays)


# ------------------------------------------------------------------ (9) second-design blocks
def gen_second_design():
    torch.manual_seed(0)
    m = R_gen.GroupedResidualConv1D(64, 3, 3)
    torch.manual_seed(1)
    x = torch.randn(2, 64, 30)
    module_case("grouped_residual_64_k3_d3", m, [x], input_names=["x"])
    torch.manual_seed(0)
    f = R_gen.FeatureWiseLinearModulation(448, 64)
    torch.manual_seed(1)
    x = torch.randn(2, 64, 9)
    spk = torch.randn(2, 448)
    emo = torch.randn(2, 448)
    module_case("film2_448_64", f, [x, spk, emo], input_names=["x", "spk", "emo"])


# ------------------------------------------------------------------ (10) whole generator
class ComposedGenerator(nn.Module):
    """The deleted generator, re-composed from the reference's own importable classes exactly as
    SURVEY.md §A specifies (this class is fixture tooling written for this repo)."""

    def __init__(self, mel_channels=80, hidden_channels=512, kernel_size=7, upsample_factors=(8, 8, 2, 2),
                 resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5),) * 3,
                 groups=4, lora_rank=16, dropout=0.1):
        super().__init__()
        self.input_proj = R_od.ODConv1d(mel_channels, hidden_channels, kernel_size, padding=kernel_size // 2,
                                        K=4, reduction_factor=4)
        self.upsample_layers = nn.ModuleList()
        cur = hidden_channels
        n = len(upsample_factors)
        for i, f in enumerate(upsample_factors):
            out = max(cur // 2, groups * 2) if i < n - 1 else max(cur, groups * 2)
            out = out // groups * groups
            out = max(out, 64)
            self.upsample_layers.append(nn.Sequential(
                R_od.ODConvTranspose1d(cur, out, kernel_size=f * 2, stride=f, padding=f // 2,
                                       output_padding=f % 2, K=4, reduction_factor=4),
                nn.LeakyReLU(0.1)))
            cur = out
        self.mrf_blocks = nn.ModuleList()
        for i, (kernel_size, dilations) in enumerate(zip(resblock_kernel_sizes, resblock_dilation_sizes)):
            mrf_ch = max(cur, groups * len(dilations) * 2)
            mrf_ch = mrf_ch // groups * groups
            self.mrf_blocks.append(R_grc.MultiReceptiveFieldBlock(
                cur, mrf_ch, dilations=list(dilations), groups=min(groups, mrf_ch // 4), r=lora_rank,
                dropout=dropout))
            cur = mrf_ch
        self.output_proj = nn.Conv1d(cur, 1, kernel_size, padding=kernel_size // 2)  # leaked loop var
        self.final_film = R_grc.FiLMLayer(cur, cur)
        nn.init.kaiming_normal_(self.output_proj.weight, mode="fan_out", nonlinearity="leaky_relu")
        nn.init.zeros_(self.output_proj.bias)
        for layer in self.upsample_layers:
            if hasattr(layer[0], "_initialize_weights"):
                layer[0]._initialize_weights()

    def stages(self, mel, spk=None, emo=None):
        st = {}
        x = self.input_proj(mel)
        st["input_proj"] = x
        if spk is not None or emo is not None:
            x = self.final_film(x, spk, emo)
            st["film"] = x
        for i, up in enumerate(self.upsample_layers):
            x = up(x)
            st[f"up{i}"] = x
        for i, blk in enumerate(self.mrf_blocks):
            x = blk(x, spk, emo)
            st[f"mrf{i}"] = x
        x = self.output_proj(x)
        st["output_proj"] = x
        x = torch.tanh(x)
        st["wave"] = x
        return st

    def forward(self, mel, spk=None, emo=None):
        return self.stages(mel, spk, emo)["wave"]


def gen_generator_small():
    torch.manual_seed(0)
    g = ComposedGenerator(hidden_channels=64, upsample_factors=(4, 2))
    for up in g.upsample_layers:
        randomise_bias(up[0])
    randomise_bias(g.input_proj)
    g.train(False)
    torch.manual_seed(2)
    with torch.no_grad():
        g(torch.randn(1, 80, 8))  # warm-up: materialise residual_proj
    torch.manual_seed(1)
    mel = torch.randn(2, 80, 8)
    spk = torch.randn(2, 192)
    emo = torch.randn(2, 384)
    arrays = _state(g)
    arrays.update({"x.mel": _np(mel), "x.spk": _np(spk), "x.emo": _np(emo)})
    with torch.no_grad():
        st = g.stages(mel, spk, emo)
        st_nc = g.stages(mel)
    for k, v in st.items():
        arrays["stage." + k] = _np(v)
    arrays["wave_nocond"] = _np(st_nc["wave"])
    # gradient golden through the whole generator
    for p in g.parameters():
        p.grad = None
    m2 = mel.clone().requires_grad_(True)
    s2 = spk.clone().requires_grad_(True)
    y = g(m2, s2, emo)
    torch.manual_seed(11)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    arrays["r"] = _np(r)
    arrays["gx.mel"] = _np(m2.grad)
    arrays["gx.spk"] = _np(s2.grad)
    arrays.update(_grads(g))
    _save("generator_small", **arrays)


def gen_generator_full():
    """Default-size generator (12.2 M parameters): seeds, per-tensor checksums, the 9 lazily created
    residual_proj convs (they depend on the RNG state at first forward) and the output waveform."""
    for tag, kw, n_mel, T in [("generator_full_22k", {}, 80, 32),
                              ("generator_full_48k", dict(mel_channels=128, upsample_factors=(8, 8, 4, 2)), 128, 16)]:
        torch.manual_seed(0)
        g = ComposedGenerator(**kw)
        g.train(False)
        torch.manual_seed(2)
        with torch.no_grad():
            g(torch.randn(1, n_mel, 4))
        torch.manual_seed(1)
        mel = torch.randn(1, n_mel, T)
        spk = torch.randn(1, 192)
        emo = torch.randn(1, 384)
        with torch.no_grad():
            st = g.stages(mel, spk, emo)
        arrays = {"x.mel": _np(mel), "x.spk": _np(spk), "x.emo": _np(emo), "wave": _np(st["wave"])}
        for k in ("input_proj", "film", "up0", "up1", "up2", "up3", "mrf0", "mrf1", "mrf2"):
            v = st[k].double()
            arrays["stagechk." + k] = np.array([v.sum().item(), v.abs().sum().item(), float(v.numel())])
        n_par = 0
        for k, v in g.state_dict().items():
            n_par += v.numel()
            arrays["chk." + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
            if ".residual_proj." in k:
                arrays["sd." + k] = _np(v)
        arrays["n_params"] = np.array(n_par)
        _save(tag, **arrays)


def main():
    print("generating goldens from /root/reference (CPU fp32)")
    gen_odconv1d()
    gen_odconvT()
    gen_grc()
    gen_mrf()
    gen_film()
    gen_disc()
    gen_system_losses()
    gen_second_design()
    gen_generator_small()
    gen_generator_full()
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print(f"total fixture size: {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
