"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the drop-in
modules have the reference's constructor surface / state_dict keys / seeded initialisation, and the
product path refuses to run without a GPU instead of falling back."""
import inspect
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, ROOT, PKG


def test_library_exports_every_declared_symbol():
    from hifigan_modified import _native
    protos = _native.declared_symbols()
    assert len(protos) >= 18
    lib = _native.lib()
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.mv_abi_version() >= 1
    assert lib.mv_build_target() == b"gfx950"


def test_header_and_library_are_in_tree():
    from hifigan_modified import _native
    assert os.path.exists(os.path.join(ROOT, "include", "mi355x_vocoder.h"))
    assert _native.LIB_PATH.startswith(PKG)


def test_no_oracle_import_in_product():
    """The product package must never import the oracle (parity would be void)."""
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "vocoder_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_cpu_tensor_is_refused():
    import hifigan_modified as H
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        H.ODConv1d(4, 4, 3, padding=1)(torch.randn(1, 4, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        H.Discriminator1D(2)(torch.randn(1, 1, 64))


def test_constructor_surface_matches_reference():
    import hifigan_modified as H
    def params(cls):
        return [(n, p.default) for n, p in inspect.signature(cls.__init__).parameters.items() if n != "self"]
    assert params(H.ODConv1d) == [("in_channels", inspect._empty), ("out_channels", inspect._empty),
                                  ("kernel_size", inspect._empty), ("stride", 1), ("padding", 0), ("dilation", 1),
                                  ("groups", 1), ("K", 4), ("reduction_factor", 4)]
    assert params(H.ODConvTranspose1d)[3:] == [("stride", 1), ("padding", 0), ("output_padding", 0), ("dilation", 1),
                                               ("groups", 1), ("K", 4), ("reduction_factor", 4)]
    assert params(H.GRC_LoRA_Block) == [("in_channels", inspect._empty), ("out_channels", inspect._empty),
                                        ("kernel_size", inspect._empty), ("dilation", inspect._empty), ("r", 4)]
    assert params(H.MultiReceptiveFieldBlock)[2:] == [("dilations", [1, 3, 5]), ("groups", 4), ("r", 16), ("dropout", 0.1)]
    assert params(H.MultiPeriodDiscriminator) == [("periods", [2, 3, 5, 7, 11])]
    assert params(H.MultiScaleDiscriminator) == [("scales", [1, 2, 4])]
    gp = dict(params(H.ModifiedHiFiGANGenerator))
    assert gp["mel_channels"] == 80 and gp["hidden_channels"] == 512 and gp["upsample_factors"] == [8, 8, 2, 2]
    assert gp["lora_rank"] == 16 and gp["dropout"] == 0.1 and gp["groups"] == 4 and gp["kernel_size"] == 7


@pytest.mark.parametrize("name,ctor", [
    ("odconv1d_c16_o8_k3_d2", lambda H: H.ODConv1d(16, 8, 3, padding=2, dilation=2)),
    ("odconvT_c16_o8_k16_s8", lambda H: H.ODConvTranspose1d(16, 8, 16, stride=8, padding=4)),
    ("grc_64_20_d3", lambda H: H.GRC_LoRA_Block(64, 20, 3, 3, 16)),
    ("mrf_64_64", lambda H: H.MultiReceptiveFieldBlock(64, 64)),
    ("film_64_576_both", lambda H: H.FiLMLayer(64, 576)),
    ("disc2d_P3", lambda H: H.Discriminator2D(3)),
    ("disc1d_s2", lambda H: H.Discriminator1D(2)),
    ("grouped_residual_64_k3_d3", lambda H: H.GroupedResidualConv1D(64, 3, 3)),
    ("film2_448_64", lambda H: H.FeatureWiseLinearModulation(448, 64)),
    ("generator_small", lambda H: H.ModifiedHiFiGANGenerator(hidden_channels=64, upsample_factors=[4, 2])),
])
def test_state_dict_keys_and_shapes_match_reference(name, ctor):
    import hifigan_modified as H
    g = load_golden(name)
    ref = {k[3:]: v.shape for k, v in g.items() if k.startswith("sd.")}
    ours = {k: tuple(v.shape) for k, v in ctor(H).state_dict().items()}
    assert set(ours) == set(ref)
    for k in ref:
        assert tuple(ref[k]) == ours[k], k


def test_seeded_init_reproduces_reference_weights():
    import hifigan_modified as H
    g = load_golden("generator_full_22k")
    torch.manual_seed(0)
    m = H.ModifiedHiFiGANGenerator()
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 12231269
    for k, v in m.state_dict().items():
        if ".residual_proj." in k:
            continue
        got = np.array([v.double().sum().item(), v.double().abs().sum().item()])
        assert np.allclose(got, g["chk." + k], rtol=1e-9, atol=1e-9), k
    assert m.output_proj.kernel_size == (11,) and m.output_proj.padding == (5,)
    assert [l[0].out_channels for l in m.upsample_layers] == [256, 128, 64, 64]
    n_unused = sum(p.numel() for p in m.unused_parameters())
    assert n_unused == 335859  # never-used ODConv attention parameters


def test_reference_checkpoint_without_residual_proj_loads():
    import hifigan_modified as H
    m = H.MultiReceptiveFieldBlock(64, 64)
    sd = {k: v for k, v in m.state_dict().items() if "residual_proj" not in k}
    m.load_state_dict(sd, strict=True)


def test_film_condition_glue():
    import hifigan_modified as H
    f = H.FiLMLayer(64, 64)
    spk, emo = torch.randn(2, 192), torch.randn(2, 384)
    assert torch.equal(f.condition(spk, emo), spk[:, :64])       # truncation: emotion ignored
    f2 = H.FiLMLayer(16, 600)
    c = f2.condition(spk, emo)
    assert c.shape == (2, 600) and torch.equal(c[:, :192], spk) and c[:, 576:].abs().sum() == 0
    assert f.condition(None, None) is None


def test_mel_loss_rejects_wrong_shapes_before_any_launch():
    """ops.mel_loss checks what the C entry point cannot (it receives no sizes for target / filterbank): a target with one
    frame too many (a center=True front-end), another n_mels, an fb built for another n_fft or a ragged T raise ValueError
    like the reference's F.l1_loss / F.mse_loss would - instead of reading out of bounds on the device."""
    from hifigan_modified import ops
    wave, fb = torch.zeros(2, 1, 2048), torch.zeros(80, 513)
    for bad_target in (torch.zeros(2, 80, 9), torch.zeros(2, 64, 8), torch.zeros(1, 80, 8)):
        with pytest.raises(ValueError, match="target"):
            ops.mel_loss(wave, fb, bad_target, n_fft=1024, hop=256)
    with pytest.raises(ValueError, match="filterbank"):
        ops.mel_loss(wave, torch.zeros(80, 257), torch.zeros(2, 80, 8), n_fft=1024, hop=256)
    with pytest.raises(ValueError, match="multiple of hop"):
        ops.mel_loss(torch.zeros(2, 1, 2000), fb, torch.zeros(2, 80, 7), n_fft=1024, hop=256)
    with pytest.raises(ValueError, match=r"\[B, 1, T\]"):
        ops.mel_loss(torch.zeros(2, 2, 2048), fb, torch.zeros(2, 80, 8), n_fft=1024, hop=256)


def test_torch_library_operators_are_registered_and_refuse_cpu_tensors():
    """torch.ops.mi355x_vocoder.* exists with the documented schemas; there is no CPU kernel (no fallback)."""
    import hifigan_modified  # noqa: F401
    from hifigan_modified import torch_ops  # noqa: F401
    ns = torch.ops.mi355x_vocoder
    for name in ("odconv_attn", "odconv1d", "odconv_transpose1d", "conv1d", "conv2d", "group_norm", "film", "grc_mrf_block",
                 "generator_forward", "avg_pool1d", "mpd_fold", "disc_conv_stack", "gan_loss", "mel_loss", "mel_spectrogram",
                 "fused_adamw_"):
        assert hasattr(ns, name), name
    sch = str(ns.odconv_transpose1d.default._schema)
    assert "Tensor kernels" in sch and "int output_padding" in sch and "int fused" in sch
    assert "Tensor(a!) p" in str(ns.fused_adamw_.default._schema)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ns.avg_pool1d(torch.randn(1, 1, 8), 2)


def test_operators_have_fake_kernels_with_the_reference_shapes():
    """register_fake for every operator: FakeTensor tracing needs no GPU and reproduces the output sizes of the reference's modules
    (nn.Conv1d / nn.ConvTranspose1d inside odconv.py:89-106,187-204, nn.AvgPool1d discriminators.py:94, the fold :72-79)."""
    import hifigan_modified  # noqa: F401
    from hifigan_modified import torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    ns = torch.ops.mi355x_vocoder
    dev = "cuda" if torch.cuda.is_available() else "meta"
    with FakeTensorMode(allow_non_fake_inputs=True):
        e = lambda *s, **k: torch.empty(*s, device="cuda", **k)
        for (cin, cout, ks, stride, pad, dil, T) in ((16, 8, 3, 1, 2, 2, 24), (80, 64, 7, 1, 3, 1, 32), (8, 8, 5, 2, 2, 1, 31)):
            y = ns.odconv1d(e(2, cin, T), e(4, cout, cin, ks), e(4, cout), e(4, cin, 1), e(4), stride, pad, 0, dil, 0, 0.1, 0)
            assert tuple(y.shape) == tuple(torch.nn.Conv1d(cin, cout, ks, stride, pad, dil)(torch.empty(2, cin, T, device="meta")).shape)
            y = ns.conv1d(e(2, cin, T), e(cout, cin, ks), None, stride, pad, dil, 1, 0, 0.1)
            assert tuple(y.shape) == tuple(torch.nn.Conv1d(cin, cout, ks, stride, pad, dil, device="meta")(torch.empty(2, cin, T, device="meta")).shape)
        for (cin, cout, ks, stride, pad, opad, T) in ((16, 8, 16, 8, 4, 0, 10), (8, 8, 4, 2, 1, 0, 33), (8, 4, 6, 3, 1, 1, 10)):
            y = ns.odconv_transpose1d(e(2, cin, T), e(4, cin, cout, ks), e(4, cout), e(4, cin, 1), e(4), stride, pad, opad, 1, 1, 0.1, 0)
            ref = torch.nn.ConvTranspose1d(cin, cout, ks, stride, pad, opad, device="meta")(torch.empty(2, cin, T, device="meta"))
            assert tuple(y.shape) == tuple(ref.shape)
        assert tuple(ns.conv2d(e(2, 32, 3, 100), e(64, 32, 3, 3), e(64), 1, 1, 1, 0.1).shape) == (2, 64, 3, 100)
        assert tuple(ns.avg_pool1d(e(2, 1, 1001), 4).shape) == (2, 1, 250)
        for P, T in ((2, 8192), (3, 1000), (11, 8192)):
            assert tuple(ns.mpd_fold(e(2, 1, T), P).shape) == (2, 1, P, -(-T // P))
        assert ns.odconv_attn(e(3, 16, 9, dtype=torch.bfloat16), e(4, 16, 1), e(4)).dtype == torch.float32
        assert tuple(ns.mel_spectrogram(e(2, 1, 2048), e(80, 513), 1024, 256, 1e-5).shape) == (2, 80, 8)
        assert ns.gan_loss(e(2, 1, 50), None, 0, 1.0, 1.0).shape == () and ns.mel_loss(e(2, 1, 2048), e(2, 80, 8), e(80, 513), 1024, 256, 1e-5, 45.0, 0).shape == ()
        x = e(2, 64, 100, dtype=torch.float16)
        assert ns.group_norm(x, e(64), e(64), None, None, 8, 1e-5, 0, 0.1, 1.0).dtype == torch.float16
        assert tuple(ns.film(e(2, 512, 5), e(2, 64), e(128, 64), e(128), 64).shape) == (2, 512, 5)
        ws = [e(32, 1, 3, 3), e(32), e(64, 32, 3, 3), e(64), e(128, 64, 3, 3), e(128), e(256, 128, 3, 3), e(256), e(1, 256, 3, 3), e(1)]
        assert tuple(ns.disc_conv_stack(e(2, 1, 3, 334), 0.1, *ws).shape) == (2, 1, 3, 334)


def test_mixed_precision_switch_is_host_state_only():
    """set_mixed_precision only records the split point; it validates its arguments and does not touch parameters."""
    import hifigan_modified as H
    g = H.ModifiedHiFiGANGenerator()
    before = {k: v.clone() for k, v in g.state_dict().items()}
    assert g.mixed_precision is None
    assert g.set_mixed_precision("up1") is g and g.mixed_precision == ("up1", torch.float16)
    assert g.set_mixed_precision("input_proj", torch.bfloat16).mixed_precision == ("input_proj", torch.bfloat16)
    for bad in ("mrf0", "up9", "wave"):
        with pytest.raises(ValueError):
            g.set_mixed_precision(bad)
    with pytest.raises(ValueError):
        g.set_mixed_precision("up0", torch.float32)
    assert g.set_mixed_precision(None).mixed_precision is None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")              # the validated geometry does not warn ...
        g.set_mixed_precision("up1", mrf_weights="fp16").set_mixed_precision(None)
    g48 = H.ModifiedHiFiGANGenerator(mel_channels=128, hidden_channels=64, upsample_factors=[8, 8, 4, 2])
    with pytest.warns(UserWarning, match="48 kHz"):  # ... the 48 kHz one (no sub-fp32 mix inside 1e-3: DESIGN.md section 5) does
        g48.set_mixed_precision("up1")
    assert all(torch.equal(v, before[k]) for k, v in g.state_dict().items()) and set(g.state_dict()) == set(before)
