"""GPU parity of the HIP path (through the C ABI) against the golden vectors generated from the
reference and against the CPU oracle.  Tolerances (relative L2):
  fp32 storage (split bf16 MFMA operands) - THE parity-grade mode: north_star's 1e-3 on the waveform is asserted at the full C2
                 size (test_full_c2_batch_properties, 32 clips) and on both sample rates (generator_full_*); the bounds actually used are
                 tighter (1e-4 per module, 2e-4 waveform: summation order only)
  mixed storage (fp16 through the second upsampler, fp32 behind it: bench.py's headline) - north_star's 1e-3 asserted at the full C2
                 size (test_full_c2_batch_mixed_storage_meets_north_star; measured 4.6e-4 / 5.6e-4 for through = up0 / up1)
  fp16 / bf16 storage - OUT of north_star's tolerance by construction: tools/error_budget.py shows that rounding alone puts any
                 single-16-bit-operand pipeline at 1.5e-3 / 1.2e-2 (22 kHz; 48 kHz worse) - DESIGN.md section 5.  Their bounds below are
                 guard rails at about 2x the measured value of each case, so that a kernel regression shows; they are NOT a claim that
                 these modes meet 1e-3 (bench.py reports them with parity_ok = false)
MPD fold index map: exact equality."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import vocoder_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.float16: 2e-3, torch.bfloat16: 1.2e-2}
DTYPES = [torch.float32, torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def H():
    import hifigan_modified as H
    from hifigan_modified import _native
    _native.lib()  # fail loudly if the extension is missing
    return H


def load_sd(mod, g, prefix="sd."):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    mod.load_state_dict(sd, strict=True)
    return mod


def dev(g, key, dtype):
    return torch.from_numpy(g[key]).cuda().to(dtype)


def run(mod, dtype, *inputs):
    mod = mod.cuda().train(False)
    with torch.no_grad():
        return mod(*inputs)


def check(y, ref, dtype, scale=1.0, what=""):
    err = O.rel_l2(y.float().cpu(), torch.from_numpy(ref))
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert err < TOL[dtype] * scale, f"{what} {dtype}: rel-L2 {err:.3e}"
    return err


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,args,kw", [
    ("odconv1d_c16_o8_k3_d2", (16, 8, 3), dict(padding=2, dilation=2)),
    ("odconv1d_c80_o32_k7", (80, 32, 7), dict(padding=3)),
    ("odconv1d_c8_o8_k5_s2", (8, 8, 5), dict(padding=2, stride=2)),
])
def test_odconv1d(H, name, args, kw, dtype):
    g = load_golden(name)
    m = load_sd(H.ODConv1d(*args, **kw), g)
    x = dev(g, "x.x", dtype)
    y = run(m, dtype, x)
    check(y, g["y"], dtype, what=name)
    alpha = m.attention(x)
    assert alpha.dtype == torch.float32
    assert O.rel_l2(alpha.cpu(), torch.from_numpy(g["alpha"])) < (1e-5 if dtype == torch.float32 else 5e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,args,kw", [
    ("odconvT_c16_o8_k16_s8", (16, 8, 16), dict(stride=8, padding=4)),
    ("odconvT_c8_o8_k4_s2", (8, 8, 4), dict(stride=2, padding=1)),
    ("odconvT_c8_o8_k8_s4", (8, 8, 8), dict(stride=4, padding=2)),
    ("odconvT_c8_o4_k6_s3_op1", (8, 4, 6), dict(stride=3, padding=1, output_padding=1)),
])
def test_odconv_transpose1d(H, name, args, kw, dtype):
    g = load_golden(name)
    m = load_sd(H.ODConvTranspose1d(*args, **kw), g)
    y = run(m, dtype, dev(g, "x.x", dtype))
    check(y, g["y"], dtype, what=name)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,args", [("grc_64_20_d1", (64, 20, 3, 1, 16)), ("grc_64_20_d3", (64, 20, 3, 3, 16)),
                                       ("grc_64_20_d5", (64, 20, 3, 5, 16)), ("grc_16_16_d1_r4", (16, 16, 3, 1, 4))])
def test_grc_lora_block(H, name, args, dtype):
    g = load_golden(name)
    m = load_sd(H.GRC_LoRA_Block(*args), g)
    y = run(m, dtype, dev(g, "x.x", dtype))
    check(y, g["y"], dtype, what=name)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,args,kw", [("mrf_64_64", (64, 64), {}),
                                          ("mrf_32_32_g2", (32, 32), dict(dilations=[1, 2], groups=2, r=4))])
def test_mrf_block(H, name, args, kw, dtype):
    g = load_golden(name)
    m = load_sd(H.MultiReceptiveFieldBlock(*args, **kw), g)
    y = run(m, dtype, dev(g, "x.x", dtype))
    check(y, g["y"], dtype, what=name)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,args", [("film_64_64_both", (64, 64)), ("film_64_64_spk", (64, 64)),
                                       ("film_64_64_emo", (64, 64)), ("film_64_576_both", (64, 576)),
                                       ("film_16_600_trunc", (16, 600))])
def test_film(H, name, args, dtype):
    g = load_golden(name)
    m = load_sd(H.FiLMLayer(*args), g)
    spk = dev(g, "x.spk", dtype) if "x.spk" in g else None
    emo = dev(g, "x.emo", dtype) if "x.emo" in g else None
    y = run(m, dtype, dev(g, "x.x", dtype), spk, emo)
    check(y, g["y"], dtype, what=name)


def test_film_none_is_identity(H):
    g = load_golden("film_64_64_none")
    x = torch.from_numpy(g["x"]).cuda()
    assert H.FiLMLayer(64, 64).cuda()(x) is x


@pytest.mark.parametrize("P", [2, 3, 5, 7, 11])
@pytest.mark.parametrize("T", [1000, 8192])
def test_mpd_fold_index_bit_exact(H, P, T):
    from hifigan_modified import ops
    g = load_golden(f"mpd_index_P{P}_T{T}")
    sig = torch.arange(1, T + 1, dtype=torch.float32, device="cuda").view(1, 1, T)
    folded, index = ops.mpd_fold(sig, P, want_index=True)
    assert index.dtype == torch.int64
    assert np.array_equal(index.cpu().numpy(), g["index"])
    assert np.array_equal(folded[0, 0].to(torch.int64).cpu().numpy() - 1, g["index"])
    # the zero-copy (exact multiple) path must agree with the copying path
    if T % P == 0:
        assert ops.mpd_fold(sig, P).data_ptr() == sig.data_ptr()


@pytest.mark.parametrize("dtype", DTYPES)
def test_disc2d(H, dtype):
    g = load_golden("disc2d_P3")
    m = load_sd(H.Discriminator2D(3), g)
    y = run(m, dtype, dev(g, "x.x", dtype))
    check(y, g["y"], dtype, what="disc2d")


@pytest.mark.parametrize("dtype", DTYPES)
def test_disc1d(H, dtype):
    g = load_golden("disc1d_s2")
    m = load_sd(H.Discriminator1D(2), g)
    y = run(m, dtype, dev(g, "x.x", dtype))
    check(y, g["y"], dtype, what="disc1d")


def test_discriminator_system_from_seed(H):
    """Same seed -> same weights as the reference (checksums), then every one of the 16 outputs."""
    g = load_golden("disc_system_losses")
    torch.manual_seed(0)
    D = H.HiFiGANDiscriminators()
    for k, v in D.state_dict().items():
        got = np.array([v.double().sum().item(), v.double().abs().sum().item()])
        assert np.allclose(got, g["chk." + k], rtol=1e-9, atol=1e-9), k
    D = D.cuda()
    with torch.no_grad():
        out = D(torch.from_numpy(g["real"]).cuda(), torch.from_numpy(g["fake"]).cuda())
    for key, n in (("mpd_real", 5), ("mpd_fake", 5), ("msd_real", 3), ("msd_fake", 3)):
        assert len(out[key]) == n
        for i in range(n):
            check(out[key][i], g[f"out.{key}.{i}"], torch.float32, what=f"{key}.{i}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_second_design_blocks(H, dtype):
    g = load_golden("grouped_residual_64_k3_d3")
    m = load_sd(H.GroupedResidualConv1D(64, 3, 3), g)
    check(run(m, dtype, dev(g, "x.x", dtype)), g["y"], dtype, what="grouped_residual")
    g = load_golden("film2_448_64")
    m = load_sd(H.FeatureWiseLinearModulation(448, 64), g)
    check(run(m, dtype, dev(g, "x.x", dtype), dev(g, "x.spk", dtype), dev(g, "x.emo", dtype)), g["y"], dtype,
          scale=3.0, what="film2")


@pytest.mark.parametrize("dtype", DTYPES)
def test_generator_small_every_stage(H, dtype):
    g = load_golden("generator_small")
    m = load_sd(H.ModifiedHiFiGANGenerator(hidden_channels=64, upsample_factors=[4, 2]), g)
    m = m.cuda().train(False)
    mel, spk, emo = dev(g, "x.mel", dtype), dev(g, "x.spk", dtype), dev(g, "x.emo", dtype)
    with torch.no_grad():
        st = m(mel, spk, emo, return_stages=True)
        nc = m(mel)
    for k in ("film", "up0", "up1", "mrf0", "mrf1", "mrf2"):
        check(st[k], g["stage." + k], dtype, scale=2.0, what=k)
    # waveform: the random-init output conv (pre-tanh rms ~10) amplifies stage error on unsaturated samples
    wscale = {torch.float32: 2.0, torch.float16: 5.0, torch.bfloat16: 5.0}[dtype]
    check(st["wave"], g["stage.wave"], dtype, scale=wscale, what="wave")
    check(nc, g["wave_nocond"], dtype, scale=wscale, what="wave_nocond")


@pytest.mark.parametrize("fixture,kw", [("generator_full_22k", {}),
                                        ("generator_full_48k", dict(mel_channels=128, upsample_factors=[8, 8, 4, 2]))])
@pytest.mark.parametrize("dtype", DTYPES)
def test_generator_full_from_seed(H, fixture, kw, dtype):
    """Default-size generator rebuilt from the seed; only the 9 lazily-created residual_proj convs come
    from the fixture.  Waveform tolerance: north_star 1e-3 for fp32/fp16; bf16 see module docstring."""
    g = load_golden(fixture)
    torch.manual_seed(0)
    m = H.ModifiedHiFiGANGenerator(**kw)
    sd = m.state_dict()
    for k in g:
        if k.startswith("sd."):
            sd[k[3:]] = torch.from_numpy(g[k])
    m.load_state_dict(sd)
    for k, v in m.state_dict().items():
        got = np.array([v.double().sum().item(), v.double().abs().sum().item()])
        assert np.allclose(got, g["chk." + k], rtol=1e-9, atol=1e-9), k
    m = m.cuda().train(False)
    with torch.no_grad():
        st = m(dev(g, "x.mel", dtype), dev(g, "x.spk", dtype), dev(g, "x.emo", dtype), return_stages=True)
    wave = st["wave"]
    assert wave.shape == (1, 1, 8192)
    err = O.rel_l2(wave.float().cpu(), torch.from_numpy(g["wave"]))
    # fp32 storage (split MFMA operands) meets north_star's 1e-3 with a wide margin (measured 0 / 7.0e-5: the 22 kHz clip of this
    # fixture is fully saturated).  16-bit storage: guard rails at 2x the measured 48 kHz values (fp16 4.1e-3, bf16 3.4e-2), out of
    # the 1e-3 tolerance by construction (module docstring)
    bound = {torch.float32: 2e-4, torch.float16: 8e-3, torch.bfloat16: 6.5e-2}[dtype]
    print(f"[parity] {fixture} {dtype}: waveform rel-L2 {err:.3e}")
    assert err < bound, f"{fixture} {dtype}: waveform rel-L2 {err:.3e}"
    if dtype == torch.float32:
        for k in ("film", "up0", "up1", "up2", "up3", "mrf0", "mrf1", "mrf2"):
            v = st[k].double()
            chk = g["stagechk." + k]
            assert abs(v.abs().sum().item() - chk[1]) < 2e-4 * chk[1], k


def test_cpu_tensors_are_rejected(H):
    m = H.ODConv1d(8, 8, 3, padding=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 8, 16))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,B", [(48, 2), (700, 3), (8192, 2)])
def test_mrf_fused_vs_generic_and_golden(H, dtype, T, B):
    """Fused channels-last MFMA MRF (csrc/mrf_fused.hip) against the golden (T=48) and against the generic
    fp32 HIP path on longer ragged inputs (several tiles, partial last tile, multi-workgroup statistics)."""
    from hifigan_modified import functional as Fn
    g = load_golden("mrf_64_64")
    m = load_sd(H.MultiReceptiveFieldBlock(64, 64), g).cuda().train(False)
    if T == 48:
        x = dev(g, "x.x", dtype)
        ref = torch.from_numpy(g["y"])
    else:
        torch.manual_seed(3)
        x32 = torch.randn(B, 64, T, device="cuda")
        x = x32.to(dtype)
        with torch.no_grad():
            ref = Fn.mrf_block(x.float(), m, force_generic=True).cpu()
    with torch.no_grad():
        y = Fn.mrf_block(x, m)
    err = O.rel_l2(y.float().cpu(), ref)
    bound = {torch.float32: 2e-5, torch.float16: 1.5e-3, torch.bfloat16: 8e-3}[dtype]
    assert err < bound, f"fused MRF {dtype} T={T}: rel-L2 {err:.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,args,kw", [
    ("odconv1d_c16_o8_k3_d2", (16, 16, 3), dict(padding=2, dilation=2)),
    ("odconv1d_c80", (80, 32, 7), dict(padding=3)),
    ("odconvT_k16_s8", (16, 8, 16), dict(stride=8, padding=4)),
    ("odconvT_k4_s2", (8, 8, 4), dict(stride=2, padding=1)),
    ("odconvT_k8_s4", (24, 16, 8), dict(stride=4, padding=2)),
    ("odconvT_k6_s2", (32, 16, 6), dict(stride=2, padding=2)),
    # ks = 2*stride with 64 / 128 input channels: the multi-tile kernel (odconv_cl_mt_kernel) in 16-bit storage
    ("odconvT_c64_k4_s2", (64, 64, 4), dict(stride=2, padding=1)),
    ("odconvT_c128_k8_s4", (128, 64, 8), dict(stride=4, padding=2)),
    ("odconvT_c64_k4_s2_op1", (64, 32, 4), dict(stride=2, padding=1, output_padding=1)),
    ("odconvT_c128_k4_s2", (128, 96, 4), dict(stride=2, padding=1)),
    ("odconvT_c256_k16_s8", (256, 128, 16), dict(stride=8, padding=4)),
    ("odconvT_c256_k4_s2", (256, 320, 4), dict(stride=2, padding=1)),
])
@pytest.mark.parametrize("T", [7, 50, 300, 1500])
def test_odconv_fused_vs_generic(H, name, args, kw, dtype, T):
    """Fused channels-last ODConv (csrc/odconv_fused.hip) vs the generic HIP kernel (itself pinned to the goldens)."""
    from hifigan_modified import functional as Fn, ops
    from hifigan_modified import _native as N
    from hifigan_modified.fused import OdconvFused
    torch.manual_seed(5)
    cls = H.ODConvTranspose1d if name.startswith("odconvT") else H.ODConv1d
    m = cls(*args, **kw).cuda()
    with torch.no_grad():
        m.bias.copy_(torch.randn_like(m.bias) * 0.5)
    B = 3
    x32 = torch.randn(B, args[0], T, device="cuda")
    x = x32.to(dtype)
    fz = OdconvFused(m)
    assert fz.supported()
    with torch.no_grad():
        ref = m(x32, act="lrelu").cpu()
        alpha = m.attention(x)
        y1 = ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x), Fn._cache, alpha=alpha, act=N.ACT_LRELU))
        # same, with alpha formed in the prologue from pooled sums, and pooled_out accumulated
        pooled = x.float().sum(dim=2).contiguous()
        # pooled_out: uninitialised (NaN-filled here) partial sums, slots x GEMM rows per sample, each written exactly once
        pout = torch.full((B, fz.pool_floats(B, T, dtype, N.ACT_LRELU)), float("nan"), device="cuda")
        y2 = ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x), Fn._cache, pooled_in=pooled, pooled_out=pout, act=N.ACT_LRELU))
        y3 = ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x), Fn._cache, pooled_in=pooled, pooled_out=pout.clone(), act=N.ACT_LRELU))
    bound = {torch.float32: 2e-5, torch.float16: 1.5e-3, torch.bfloat16: 8e-3}[dtype]
    assert y1.shape == ref.shape
    e1, e2 = O.rel_l2(y1.float().cpu(), ref), O.rel_l2(y2.float().cpu(), ref)
    assert e1 < bound and e2 < bound, f"{name} {dtype} T={T}: {e1:.2e} {e2:.2e}"
    assert torch.equal(y2, y3)                                     # no atomics anywhere: bit-identical reruns
    assert O.rel_l2(pout.view(B, -1, args[1]).sum(dim=1).cpu(), y2.float().sum(dim=2).cpu()) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_generator_fused_pipeline_every_stage(H, dtype):
    """Whole generator through the fused channels-last pipeline vs the oracle, stage by stage (B=3 ragged T)."""
    torch.manual_seed(0)
    m = H.ModifiedHiFiGANGenerator()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(1)
    mel, spk, emo = torch.randn(3, 80, 13), torch.randn(3, 192), torch.randn(3, 384)
    with torch.no_grad():
        ref = O.generator_forward(mel, sd, "", spk, emo, return_stages=True)
    m = m.cuda().train(False)
    with torch.no_grad():
        st = m(mel.cuda().to(dtype), spk.cuda().to(dtype), emo.cuda().to(dtype), return_stages=True)
        st_g = m(mel.cuda().to(dtype), spk.cuda().to(dtype), emo.cuda().to(dtype), return_stages=True, force_generic=True)
    assert st["wave"].shape == (3, 1, 13 * 256)
    # measured: worst stage 1.1e-5 / 8.0e-4 / 6.6e-3, waveform 1.7e-5 / 1.08e-3 / 1.07e-2 (fp32 / fp16 / bf16); 16-bit = 2x guard rails
    stage_bound = {torch.float32: 5e-5, torch.float16: 1.6e-3, torch.bfloat16: 1.3e-2}[dtype]
    for k in ("film", "up0", "up1", "up2", "up3", "mrf0", "mrf1", "mrf2"):
        e = O.rel_l2(st[k].float().cpu(), ref[k])
        assert e < stage_bound, f"{k} {dtype}: {e:.2e}"
    ew = O.rel_l2(st["wave"].float().cpu(), ref["wave"])
    eg = O.rel_l2(st_g["wave"].float().cpu(), ref["wave"])
    wave_bound = {torch.float32: 2e-4, torch.float16: 2.2e-3, torch.bfloat16: 2.2e-2}[dtype]
    print(f"[parity] fused pipeline B=3 T=13 {dtype}: wave {ew:.3e} (generic {eg:.3e}) worst stage "
          f"{max(O.rel_l2(st[k].float().cpu(), ref[k]) for k in ('film', 'up0', 'up1', 'up2', 'up3', 'mrf0', 'mrf1', 'mrf2')):.3e}")
    assert ew < wave_bound, f"wave {dtype}: {ew:.2e} (generic path {eg:.2e})"
    # fewer stored roundings: the fused path must not be less accurate than the generic one (with slack for noise)
    assert ew < 1.5 * eg + 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T", [(5, 32), (8, 20), (4, 40)])
def test_odconv_kloop_first_upsampler(H, dtype, B, T):
    """K-loop weights-stationary kernel (odconv_kloop_kernel: banks used as stored, alpha applied to the accumulators)
    at the first upsampler's geometry, ragged sample groups, vs the generic fp32 HIP kernel."""
    from hifigan_modified import functional as Fn, ops
    from hifigan_modified import _native as N
    from hifigan_modified.fused import OdconvFused
    torch.manual_seed(7)
    m = H.ODConvTranspose1d(512, 256, 16, stride=8, padding=4).cuda()
    with torch.no_grad():
        m.bias.copy_(torch.randn_like(m.bias) * 0.5)
    x32 = torch.randn(B, 512, T, device="cuda")
    x = x32.to(dtype)
    fz = OdconvFused(m)
    with torch.no_grad():
        ref = m(x32, act="lrelu").cpu()
        pooled = x.float().sum(dim=2).contiguous()
        pout = torch.full((B, fz.pool_floats(B, T, dtype, N.ACT_LRELU)), float("nan"), device="cuda")
        y = ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x), Fn._cache, pooled_in=pooled, pooled_out=pout, act=N.ACT_LRELU))
    bound = {torch.float16: 1.5e-3, torch.bfloat16: 8e-3}[dtype]
    assert y.shape == ref.shape
    e = O.rel_l2(y.float().cpu(), ref)
    assert e < bound, f"kloop {dtype} B={B} T={T}: {e:.2e}"
    assert O.rel_l2(pout.view(B, -1, 256).sum(dim=1).cpu(), y.float().sum(dim=2).cpu()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T", [(16, 256), (19, 250), (3, 256), (17, 271), (16, 260), (16, 272)])
def test_odconv_sample_resident_second_upsampler(H, dtype, B, T):
    """Second upsampler's geometry (ODConvTranspose1d 256 -> 128, x8): from 16 samples up the sample-resident kernel runs
    (odconv_sample_kernel: whole sample in LDS by LDS-DMA, mixed A fragments streamed, 8-byte stores; 257..272 columns, i.e.
    T = 256..271 - both ends tested), below that - and for T = 250 / 272, whose columns do not fit the kernel's 17 column tiles - the
    multi-tile kernel; both against the generic fp32 HIP
    kernel, with the channel sums handed to the next layer checked against the stored output."""
    from hifigan_modified import functional as Fn, ops
    from hifigan_modified import _native as N
    from hifigan_modified.fused import OdconvFused
    torch.manual_seed(11)
    m = H.ODConvTranspose1d(256, 128, 16, stride=8, padding=4).cuda()
    with torch.no_grad():
        m.bias.copy_(torch.randn_like(m.bias) * 0.5)
    x32 = torch.randn(B, 256, T, device="cuda")
    x = x32.to(dtype)
    fz = OdconvFused(m)
    with torch.no_grad():
        ref = m(x32, act="lrelu").cpu()
        pooled = x.float().sum(dim=2).contiguous()
        pout = torch.full((B, fz.pool_floats(B, T, dtype, N.ACT_LRELU)), float("nan"), device="cuda")
        y = ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x), Fn._cache, pooled_in=pooled, pooled_out=pout, act=N.ACT_LRELU))
        y2 = ops.ntc_to_nct(fz.forward_cl(ops.nct_to_ntc(x), Fn._cache, pooled_in=pooled, pooled_out=pout, act=N.ACT_LRELU))
    bound = {torch.float16: 1.5e-3, torch.bfloat16: 8e-3}[dtype]
    assert y.shape == ref.shape == (B, 128, 8 * T) and torch.equal(y, y2)
    e = O.rel_l2(y.float().cpu(), ref)
    assert e < bound, f"second upsampler {dtype} B={B} T={T}: {e:.2e}"
    assert bool(torch.isfinite(pout).all())
    assert O.rel_l2(pout.view(B, -1, 128).sum(dim=1).cpu(), y.float().sum(dim=2).cpu()) < 1e-4


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float16, 2e-2)])
@pytest.mark.parametrize("Tm", [32, 344])
def test_plain_hifigan_v3_vs_cpu_restatement(H, dtype, tol, Tm):
    """BASELINE configs[0]: plain HiFi-GAN V3 generator, batch 1, random weights (344 frames = 4 s at 22.05 kHz).
    PARITY UNPINNED: fairseq's Generator is absent, so the HIP path is checked against this build's CPU restatement of the
    published architecture (oracle.plain_hifigan_forward) only.  fp16 bound: the rounding floor of 16-bit activations."""
    from hifigan_modified.plain_hifigan import PlainHiFiGANGenerator
    torch.manual_seed(0)
    g = PlainHiFiGANGenerator()
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    torch.manual_seed(1)
    mel = torch.randn(1, 80, Tm)
    ref = O.plain_hifigan_forward(mel, sd)
    g = g.cuda().to(dtype).train(False)
    y = g(mel.cuda().to(dtype))
    assert y.shape == (1, 1, Tm * 256) and y.dtype == dtype
    assert O.rel_l2(y.float().cpu(), ref) < tol
    assert g.fused_supported()                                         # every conv ran on the fused channels-last MFMA kernel
    assert O.rel_l2(g(mel.cuda().to(dtype), force_generic=True).float().cpu(), ref) < tol     # and the generic NCT path agrees too
    # the captured forward (what bench.py times) replays to the same waveform, also on new input in the static buffer
    replay = g.graphed(mel.cuda().to(dtype))
    assert torch.equal(replay(), y)
    torch.manual_seed(2)
    mel2 = torch.randn(1, 80, Tm)
    replay.mel.copy_(mel2.cuda().to(dtype))
    assert O.rel_l2(replay().float().cpu(), O.plain_hifigan_forward(mel2, sd)) < tol


def test_full_c2_batch_properties(H):
    """BASELINE configs[1] at its full size (B=32 clips x 32 mel frames -> 8192 samples, default generator) through
    size-independent properties, since the CPU oracle takes seconds per clip:
    (1) per-sample independence - every op of the path is per-sample, so clip i of the batch equals clip i vocoded alone
        (this also exercises the batch-dependent kernel selection: K-loop vs aggregate ODConv, samples per workgroup);
    (2) three of the 32 clips against the oracle (fp32 storage, north_star's 1e-3 waveform tolerance, measured ~2e-5);
    (3) the waveform is finite and inside tanh's range."""
    torch.manual_seed(0)
    gen = H.ModifiedHiFiGANGenerator()
    sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    gen = gen.cuda().train(False)
    torch.manual_seed(1)
    mel, spk, emo = torch.randn(32, 80, 32), torch.randn(32, 192), torch.randn(32, 384)
    with torch.no_grad():
        wave = gen(mel.cuda(), spk.cuda(), emo.cuda())
        assert wave.shape == (32, 1, 8192)
        assert bool(torch.isfinite(wave).all()) and float(wave.abs().max()) <= 1.0
        for i in (0, 13, 31):
            solo = gen(mel[i:i + 1].cuda(), spk[i:i + 1].cuda(), emo[i:i + 1].cuda())
            assert O.rel_l2(wave[i:i + 1].cpu(), solo.cpu()) < 1e-4, i
            ref = O.generator_forward(mel[i:i + 1], sd, "", spk[i:i + 1], emo[i:i + 1])
            assert O.rel_l2(wave[i:i + 1].cpu(), ref) < 1e-3, i


@pytest.mark.parametrize("through", ["up0", "up1"])
def test_full_c2_batch_mixed_storage_meets_north_star(H, through):
    """The mixed storage mode that bench.py headlines (ModifiedHiFiGANGenerator.set_mixed_precision: fp16 storage up to and including
    `through`, fp32 storage with split MFMA operands behind it) at BASELINE configs[1]'s full size: four of the 32 clips against the
    oracle within north_star's 1e-3 waveform rel-L2 (measured 4.6e-4 / 5.6e-4; tools/error_budget.py predicts 4.4e-4 / 5.9e-4), the
    output is fp32, two runs agree bit for bit, and switching the mix off restores the all-fp32 result."""
    torch.manual_seed(0)
    gen = H.ModifiedHiFiGANGenerator()
    sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    gen = gen.cuda().train(False).set_mixed_precision(through)
    assert gen.mixed_precision == (through, torch.float16)
    torch.manual_seed(1)
    mel, spk, emo = torch.randn(32, 80, 32), torch.randn(32, 192), torch.randn(32, 384)
    with torch.no_grad():
        wave = gen(mel.cuda(), spk.cuda(), emo.cuda())
        again = gen(mel.cuda(), spk.cuda(), emo.cuda())
        assert wave.shape == (32, 1, 8192) and wave.dtype == torch.float32 and torch.equal(wave, again)
        errs = []
        for i in (0, 7, 13, 31):
            ref = O.generator_forward(mel[i:i + 1], sd, "", spk[i:i + 1], emo[i:i + 1])
            errs.append(O.rel_l2(wave[i:i + 1].cpu(), ref))
        print(f"[parity] mixed through {through}: waveform rel-L2 vs oracle {[f'{e:.2e}' for e in errs]}")
        assert max(errs) < 1e-3, errs
        full = gen.set_mixed_precision(None)(mel.cuda(), spk.cuda(), emo.cuda())
        assert gen.mixed_precision is None
        assert O.rel_l2(full[:1].cpu(), O.generator_forward(mel[:1], sd, "", spk[:1], emo[:1])) < 1e-4
    with pytest.raises(ValueError):
        gen.set_mixed_precision("mrf0")


def test_full_c5_48k_batch_meets_north_star(H):
    """BASELINE configs[4]'s per-GPU share at its full size (128-mel, upsample [8,8,4,2], B=32 x 16 frames -> 8192 samples) in the
    parity-grade mode: two of the 32 clips against the oracle within north_star's 1e-3 (measured ~7e-5; this configuration saturates
    tanh less than the 22 kHz one, so it is the harder parity case - fp16 storage sits at 5e-3 here), per-sample independence."""
    torch.manual_seed(0)
    gen = H.ModifiedHiFiGANGenerator(mel_channels=128, upsample_factors=[8, 8, 4, 2])
    sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    gen = gen.cuda().train(False)
    torch.manual_seed(1)
    mel, spk, emo = torch.randn(32, 128, 16), torch.randn(32, 192), torch.randn(32, 384)
    with torch.no_grad():
        wave = gen(mel.cuda(), spk.cuda(), emo.cuda())
        assert wave.shape == (32, 1, 8192) and wave.dtype == torch.float32
        for i in (3, 30):
            ref = O.generator_forward(mel[i:i + 1], sd, "", spk[i:i + 1], emo[i:i + 1], upsample_factors=(8, 8, 4, 2))
            e = O.rel_l2(wave[i:i + 1].cpu(), ref)
            assert e < 1e-3, (i, e)
            solo = gen(mel[i:i + 1].cuda(), spk[i:i + 1].cuda(), emo[i:i + 1].cuda())
            assert O.rel_l2(wave[i:i + 1].cpu(), solo.cpu()) < 1e-4, i


@pytest.mark.parametrize("kind,arg", [("2d", 2), ("2d", 11), ("1d", 1)])
def test_full_size_discriminator_is_affine_when_slope_is_one(H, kind, arg):
    """Discriminator stacks at the training size (B=32 x 8192 samples, bf16 MFMA path): with LeakyReLU slope 1 the stack is
    affine, so f(a x + (1-a) y) = a f(x) + (1-a) f(y) up to rounding - a size-independent check of the channels-last MFMA
    kernels (all tile paths, image edges, the tap-row head) where an fp64 reference of the whole batch would take minutes."""
    from hifigan_modified import disc_fused, functional as Fn
    torch.manual_seed(0)
    m = (H.Discriminator2D(arg) if kind == "2d" else H.Discriminator1D(arg)).cuda()
    torch.manual_seed(1)
    x = torch.randn(32, 1, 8192, device="cuda").clamp(-1, 1)
    y = torch.randn(32, 1, 8192, device="cuda").clamp(-1, 1)
    a = 0.25

    def f(t):
        t = t.to(torch.bfloat16)
        t0 = (Fn._MpdFold.apply(t, arg) if 8192 % arg else t.view(32, 1, arg, 8192 // arg)) if kind == "2d" else t
        with torch.no_grad():
            return disc_fused.disc_stack(t0, m, slope=1.0).float()

    lhs = f(a * x + (1 - a) * y)
    rhs = a * f(x) + (1 - a) * f(y)
    assert O.rel_l2(lhs.cpu(), rhs.cpu()) < 2e-2


@pytest.mark.parametrize("Tm", [344, 1000])
def test_long_utterance_takes_the_unfused_prologue(H, Tm):
    """A long utterance (4 s = 344 frames; 11.6 s = 1000 frames) does not fit the one-workgroup-per-sample prologue kernel
    (mv_gen_prologue returns MV_ERR_UNSUPPORTED) and the channels-last pipeline falls back to its separate HIP launches;
    ragged tile edges of every fused kernel are exercised at lengths that are not multiples of the tile sizes."""
    torch.manual_seed(0)
    gen = H.ModifiedHiFiGANGenerator()
    sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    gen = gen.cuda().train(False)
    torch.manual_seed(1)
    mel, spk, emo = torch.randn(1, 80, Tm), torch.randn(1, 192), torch.randn(1, 384)
    with torch.no_grad():
        wave = gen(mel.cuda(), spk.cuda(), emo.cuda())
    assert wave.shape == (1, 1, Tm * 256)
    ref = O.generator_forward(mel, sd, "", spk, emo)
    assert O.rel_l2(wave.cpu(), ref) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,ks", [(2, 100, 11), (1, 257, 7), (3, 1000, 11), (2, 5, 3)])
def test_output_conv_packed_images_vs_fp64(H, dtype, B, T, ks):
    """mv_conv_out_pack_all + mv_conv_out_act_packed_cl (Conv1d(64, 1, ks, padding=ks//2) + tanh on the channels-last stream): the
    16-bit images run on v_dot2c_f32_{bf16,f16}, the fp32 image on scalar FMAs; all against fp64 with the weights rounded like the
    kernel rounds them."""
    import ctypes
    from hifigan_modified import ops
    from hifigan_modified import _native as N
    torch.manual_seed(ks * 100 + T)
    C = 64
    w = torch.randn(1, C, ks, device="cuda") / (C * ks) ** 0.5
    bias = 0.05
    x = torch.randn(B, T, C, device="cuda").to(dtype)
    packed = torch.empty(N.lib().mv_conv_out_packed_bytes(C, ks), device="cuda", dtype=torch.uint8)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N.call("mv_conv_out_pack_all", P(w), ops._DT[w.dtype], P(packed), C, ks, ops._stream())
    y = torch.empty(B, 1, T, device="cuda", dtype=dtype)
    N.call("mv_conv_out_act_packed_cl", P(x), P(packed), bias, P(y), B, T, C, ks, ks // 2, N.ACT_TANH, ops._dt(x), ops._stream())
    wq = w.to(dtype).double() if dtype != torch.float32 else w.double()
    ref = torch.tanh(torch.nn.functional.conv1d(x.double().transpose(1, 2), wq, torch.tensor([bias], device="cuda", dtype=torch.float64),
                                                padding=ks // 2))
    tol = {torch.float32: 1e-6, torch.float16: 6e-4, torch.bfloat16: 4e-3}[dtype]
    assert O.rel_l2(y.double().cpu(), ref.cpu()) < tol
    assert N.lib().mv_conv_out_act_packed_cl(P(x), P(packed), ctypes.c_float(bias), P(y), B, T, 32, ks, ks // 2, N.ACT_TANH, ops._dt(x),
                                             ops._stream()) == -3        # only the 64-channel stream is built


def test_mixed_mode_entry_points_match_their_cast_forms(H):
    """The two entry points that let the mixed storage mode change type without cast launches, against cast + the plain entry point:
    mv_odconv_cl_fwd_in16 (fp32 ODConvTranspose1d fed an fp16 stream: widening is exact; the two entries may pick different kernels -
    the streaming form with f16 hi + lo operands vs the multi-tile form with bf16 hi + lo - so they agree to fp32-grade rounding, and
    the widening entry agrees with an fp64 transposed convolution of the same fp16 values) and
    mv_gen_prologue_in (fp16 prologue fed fp32 mel / embeddings: x_cl and the FiLM projection see the same rounded inputs; the
    attention weights are formed from the unrounded mel, i.e. agree to fp16 rounding of the mean)."""
    from ctypes import c_void_p
    from hifigan_modified import _native as N, ops, functional as Fn
    from hifigan_modified.fused import generator_fused_for
    torch.manual_seed(0)
    gen = H.ModifiedHiFiGANGenerator().cuda().train(False)
    fz = generator_fused_for(gen)
    P = lambda t: None if t is None else c_void_p(t.data_ptr())
    B = 4
    # --- the third upsampler (128 -> 64 channels, x2): multi-tile kernel geometry
    u = fz.ups[2]
    x16 = torch.randn(B, 300, u.mod.in_channels, device="cuda").half()
    alpha = torch.softmax(torch.randn(B, u.mod.K, device="cuda"), 1)
    y_cast = u.forward_cl(x16.float(), Fn._cache, alpha=alpha, act=N.ACT_LRELU)
    y_in16 = u.forward_cl(x16, Fn._cache, alpha=alpha, act=N.ACT_LRELU, storage=torch.float32)
    assert y_in16.dtype == torch.float32 and O.rel_l2(y_in16.cpu(), y_cast.cpu()) < 2e-5
    wagg = torch.einsum("bk,kcoj->bcoj", alpha.double(), u.mod.kernels.double())
    bagg = alpha.double() @ u.mod.bias.double()
    ref = torch.stack([torch.nn.functional.conv_transpose1d(x16[i].double().t()[None], wagg[i], bagg[i], stride=u.mod.stride,
                                                            padding=u.mod.padding)[0] for i in range(B)])
    ref = torch.nn.functional.leaky_relu(ref, 0.1).transpose(1, 2)
    assert O.rel_l2(y_in16.double().cpu(), ref.cpu()) < 2e-6, O.rel_l2(y_in16.double().cpu(), ref.cpu())
    # a geometry without the widening variant falls back to the cast inside forward_cl (first upsampler: K-loop kernel)
    u0 = fz.ups[0]
    x0 = torch.randn(B, 32, u0.mod.in_channels, device="cuda").half()
    a0 = torch.softmax(torch.randn(B, u0.mod.K, device="cuda"), 1)
    assert torch.equal(u0.forward_cl(x0.float(), Fn._cache, alpha=a0, act=N.ACT_LRELU),
                       u0.forward_cl(x0, Fn._cache, alpha=a0, act=N.ACT_LRELU, storage=torch.float32))
    # --- prologue
    mel, spk, emo = torch.randn(B, 80, 32, device="cuda"), torch.randn(B, 192, device="cuda"), torch.randn(B, 384, device="cuda")
    att = gen.input_proj.kernel_attention[1]
    fp = gen.final_film.condition_projection
    dt = torch.float16
    c = Fn._cache
    outs = []
    for inputs in ((mel, spk, emo), (mel.half(), spk.half(), emo.half())):
        m, s, e = (t.contiguous() for t in inputs)
        alpha0 = torch.empty(B, att.weight.shape[0], device="cuda")
        x = torch.empty(B, 32, 80, device="cuda", dtype=dt)
        film = torch.empty(B, fp.out_features, device="cuda", dtype=dt)
        rc = N.lib().mv_gen_prologue_in(P(m), P(c.get(att.weight, dt)), P(c.get(att.bias, dt)), P(s), P(e), P(c.get(fp.weight, dt)),
                                        P(c.get(fp.bias, dt)), P(alpha0), P(x), P(film), None, 0, B, 80, 32, att.weight.shape[0], 192, 384,
                                        fp.in_features, fp.out_features, ops._DT[m.dtype], ops._DT[dt], ops._stream())
        assert rc == 0, rc
        outs.append((alpha0, x, film))
    torch.cuda.synchronize()
    (a_f32, x_f32, f_f32), (a_f16, x_f16, f_f16) = outs
    assert torch.equal(x_f32, x_f16)                                   # the channels-last copy rounds the same values
    assert O.rel_l2(f_f32.float().cpu(), f_f16.float().cpu()) < 2e-3   # projection of unrounded vs fp16-rounded embeddings
    assert float((a_f32 - a_f16).abs().max()) < 2e-3
    rc = N.lib().mv_gen_prologue_in(P(mel.bfloat16()), None, None, None, None, None, None, None, None, None, None, 0, B, 80, 32, 4, 0, 0, 0, 0,
                                    N.MV_BF16, N.MV_F16, ops._stream())
    assert rc == -2                                                    # MV_ERR_DTYPE: inputs are the storage type or fp32 only
