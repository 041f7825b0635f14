"""GPU gradient parity: backward HIP kernels (through the C ABI and the autograd Functions) against the gradient
goldens generated from the reference (tests/golden/make_goldens.py: (y*r).sum().backward()).
fp32 storage: rel-L2 <= 2e-4 per tensor (scalar gradients such as lora_scaling are cancelling sums: 1e-3)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import vocoder_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import hifigan_modified as H
    from hifigan_modified import _native
    _native.lib()
    return H


def load_sd(mod, g, prefix="sd."):
    mod.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}, strict=True)
    return mod.cuda().train(False)


def check_all(g, mod, y, inputs, tol=2e-4):
    assert O.rel_l2(y.detach().cpu(), torch.from_numpy(g["y"])) < 1e-4
    r = torch.from_numpy(g["r"]).cuda()
    (y * r).sum().backward()
    for name, x in inputs.items():
        if "gx." + name in g:
            e = O.rel_l2(x.grad.cpu(), torch.from_numpy(g["gx." + name]))
            assert e < tol, f"gx.{name}: {e:.2e}"
    nograd = set(g["nograd"].tolist())
    n = 0
    for k, p in mod.named_parameters():
        if "grad." + k in g:
            assert p.grad is not None, k
            ref = torch.from_numpy(g["grad." + k])
            e = O.rel_l2(p.grad.cpu(), ref) if ref.abs().max() > 0 else float(p.grad.abs().max())
            assert e < (1e-3 if ref.numel() == 1 else tol), f"grad.{k}: {e:.2e}"
            n += 1
        elif k in nograd:
            assert p.grad is None, f"{k} must stay grad-less (unused in the reference)"
    assert n > 0


def gin(g, key):
    return torch.from_numpy(g[key]).cuda().requires_grad_(True)


@pytest.mark.parametrize("name,args,kw", [
    ("odconv1d_c16_o8_k3_d2", (16, 8, 3), dict(padding=2, dilation=2)),
    ("odconv1d_c80_o32_k7", (80, 32, 7), dict(padding=3)),
    ("odconv1d_c8_o8_k5_s2", (8, 8, 5), dict(padding=2, stride=2)),
])
def test_odconv1d_grads(H, name, args, kw):
    g = load_golden(name)
    m = load_sd(H.ODConv1d(*args, **kw), g)
    x = gin(g, "x.x")
    check_all(g, m, m(x), {"x": x})


@pytest.mark.parametrize("name,args,kw", [
    ("odconvT_c16_o8_k16_s8", (16, 8, 16), dict(stride=8, padding=4)),
    ("odconvT_c8_o8_k4_s2", (8, 8, 4), dict(stride=2, padding=1)),
    ("odconvT_c8_o8_k8_s4", (8, 8, 8), dict(stride=4, padding=2)),
    ("odconvT_c8_o4_k6_s3_op1", (8, 4, 6), dict(stride=3, padding=1, output_padding=1)),
])
def test_odconv_transpose1d_grads(H, name, args, kw):
    g = load_golden(name)
    m = load_sd(H.ODConvTranspose1d(*args, **kw), g)
    x = gin(g, "x.x")
    check_all(g, m, m(x), {"x": x})


@pytest.mark.parametrize("name,args", [("grc_64_20_d1", (64, 20, 3, 1, 16)), ("grc_64_20_d3", (64, 20, 3, 3, 16)),
                                       ("grc_64_20_d5", (64, 20, 3, 5, 16)), ("grc_16_16_d1_r4", (16, 16, 3, 1, 4))])
def test_grc_grads(H, name, args):
    g = load_golden(name)
    m = load_sd(H.GRC_LoRA_Block(*args), g)
    x = gin(g, "x.x")
    check_all(g, m, m(x), {"x": x})


@pytest.mark.parametrize("name,args,kw", [("mrf_64_64", (64, 64), {}),
                                          ("mrf_32_32_g2", (32, 32), dict(dilations=[1, 2], groups=2, r=4))])
def test_mrf_grads(H, name, args, kw):
    g = load_golden(name)
    m = load_sd(H.MultiReceptiveFieldBlock(*args, **kw), g)
    x = gin(g, "x.x")
    check_all(g, m, m(x), {"x": x})


@pytest.mark.parametrize("name,args", [("film_64_64_both", (64, 64)), ("film_64_64_spk", (64, 64)),
                                       ("film_64_576_both", (64, 576)), ("film_16_600_trunc", (16, 600))])
def test_film_grads(H, name, args):
    g = load_golden(name)
    m = load_sd(H.FiLMLayer(*args), g)
    x = gin(g, "x.x")
    spk = gin(g, "x.spk") if "x.spk" in g else None
    emo = gin(g, "x.emo") if "x.emo" in g else None
    y = m(x, spk, emo)
    ins = {"x": x}
    if spk is not None:
        ins["spk"] = spk
    check_all(g, m, y, ins)


def test_disc_grads(H):
    g = load_golden("disc2d_P3")
    m = load_sd(H.Discriminator2D(3), g)
    x = gin(g, "x.x")
    check_all(g, m, m(x), {"x": x}, tol=5e-4)
    g = load_golden("disc1d_s2")
    m = load_sd(H.Discriminator1D(2), g)
    x = gin(g, "x.x")
    check_all(g, m, m(x), {"x": x}, tol=5e-4)


def test_generator_small_grads(H):
    g = load_golden("generator_small")
    m = load_sd(H.ModifiedHiFiGANGenerator(hidden_channels=64, upsample_factors=[4, 2]), g)
    mel, spk = gin(g, "x.mel"), gin(g, "x.spk")
    emo = torch.from_numpy(g["x.emo"]).cuda()
    y = m(mel, spk, emo)
    g2 = dict(g)
    g2["y"] = g["stage.wave"]
    check_all(g2, m, y, {"mel": mel, "spk": spk}, tol=5e-4)


def test_lsgan_losses_and_gradients(H):
    """compute_discriminator_losses / compute_generator_losses (complete_vocoder.py:89-184) on the seeded system."""
    g = load_golden("disc_system_losses")
    torch.manual_seed(0)
    D = H.HiFiGANDiscriminators()
    voc = H.ModifiedHiFiGANVocoder.__new__(H.ModifiedHiFiGANVocoder)
    torch.nn.Module.__init__(voc)
    voc.discriminators, voc.fm_weight, voc.mel_weight = D.cuda(), 10.0, 45.0
    real = torch.from_numpy(g["real"]).cuda()
    fake = torch.from_numpy(g["fake"]).cuda().requires_grad_(True)
    dl = voc.compute_discriminator_losses(real, fake)
    for k, v in dl.items():
        assert abs(float(v) - float(g["dloss." + k])) < 2e-4 * max(1.0, abs(float(g["dloss." + k]))), k
    dl["total_loss"].backward()
    # d loss / d fake runs back through 5 LeakyReLU convs x 8 discriminators: pre-activations within fp32 noise of 0
    # flip the 0.1/1 slope, so this gradient (|g| ~ 1e-7) only agrees to ~2e-3
    assert O.rel_l2(fake.grad.cpu(), torch.from_numpy(g["dloss.dfake"])) < 5e-3
    for k, p in D.named_parameters():
        if "dloss.grad." + k in g:
            assert O.rel_l2(p.grad.cpu(), torch.from_numpy(g["dloss.grad." + k])) < 1e-3, k
    fake2 = torch.from_numpy(g["fake"]).cuda().requires_grad_(True)
    mel, gen_mel = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["gen_mel"]).cuda()
    gl = voc.compute_generator_losses(real, fake2, mel, gen_mel)
    for k, v in gl.items():
        assert abs(float(v) - float(g["gloss." + k])) < 2e-4 * max(1.0, abs(float(g["gloss." + k]))), k
    gl["total_loss"].backward()
    assert O.rel_l2(fake2.grad.cpu(), torch.from_numpy(g["gloss.dfake"])) < 5e-3


def test_mel_loss_value_and_gradient_vs_oracle(H):
    """The STFT/log-mel L1 kernel against the oracle's explicit-DFT definition (parity unpinned vs the reference)."""
    from hifigan_modified import functional as Fn
    from hifigan_modified.mel import mel_filterbank
    torch.manual_seed(4)
    wave = (torch.randn(2, 1, 2048) * 0.3).clamp(-1, 1)
    target = torch.randn(2, 80, 8)
    w_ref = wave.clone().double().requires_grad_(True)
    ref = O.mel_l1_loss(w_ref, target.double())
    ref.backward()
    fb = mel_filterbank(device="cuda")
    assert O.rel_l2(fb.cpu(), torch.from_numpy(O.mel_filterbank()).float()) < 1e-6
    w = wave.cuda().requires_grad_(True)
    loss = Fn.mel_l1(w, target.cuda(), fb)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
    assert O.rel_l2(w.grad.cpu(), w_ref.grad.float()) < 2e-3
    mel = Fn.mel_spectrogram(wave.cuda(), fb)
    assert O.rel_l2(mel.cpu(), O.mel_spectrogram(wave.double()).float()) < 1e-4


def test_flat_adamw_matches_torch(H):
    from hifigan_modified.optim import FlatAdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(7, 5, device="cuda")), torch.nn.Parameter(torch.randn(33, device="cuda"))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FlatAdamW(ps, lr=1e-2, betas=(0.8, 0.99), weight_decay=1e-2)
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.8, 0.99), weight_decay=1e-2)
    for it in range(3):
        for p, q in zip(ps, ref):
            gr = torch.randn_like(q)
            p.grad, q.grad = gr.clone(), gr.clone()
        opt.step()
        topt.step()
    for p, q in zip(ps, ref):
        assert O.rel_l2(p.detach().cpu(), q.detach().cpu()) < 1e-6


def test_train_step_runs_and_learns(H):
    """VocoderTrainer (variant B) on a tiny generator: finite losses, parameters move, unused params untouched."""
    torch.manual_seed(0)
    voc = H.ModifiedHiFiGANVocoder(hidden_channels=64, upsample_factors=[4, 2], dropout=0.1)
    tr = H.VocoderTrainer(voc, device=torch.device("cuda"))
    torch.manual_seed(1)
    mel = torch.randn(2, 80, 128, device="cuda")
    real = torch.randn(2, 1, 1024, device="cuda").clamp(-1, 1)
    w0 = voc.generator.output_proj.weight.detach().clone()
    u0 = next(voc.generator.unused_parameters()).detach().clone()
    hist = []
    for _ in range(3):
        out = tr.train_step(mel, real, torch.randn(2, 192, device="cuda"), torch.randn(2, 384, device="cuda"))
        hist.append(tr.to_floats(out))
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    assert (voc.generator.output_proj.weight.detach() - w0).abs().max() > 0
    assert torch.equal(next(voc.generator.unused_parameters()).detach(), u0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind,arg,T", [("2d", 3, 1000), ("2d", 2, 512), ("2d", 11, 700), ("1d", 2, 1000), ("1d", 1, 640)])
@pytest.mark.parametrize("slope", [1.0, 0.1])
def test_disc_fused_mfma_vs_fp32(H, dtype, kind, arg, T, slope):
    """Channels-last MFMA discriminator stack (16-bit storage; csrc/disc_fused.hip): forward, input gradient and every
    parameter gradient against an fp64 torch-CPU evaluation of the same stack.
    slope = 1.0 makes the stack linear, so only rounding remains (tight bounds: this is the kernel-correctness check).
    slope = 0.1 (the real LeakyReLU): 16-bit rounding flips the slope of pre-activations that sit within rounding of 0,
    which costs ~sqrt(flipped fraction) in relative gradient error (measured 2-4 % fp16, 6-11 % bf16)."""
    import torch.nn.functional as F
    from hifigan_modified import disc_fused, functional as Fn
    torch.manual_seed(0)
    m = (H.Discriminator2D(arg) if kind == "2d" else H.Discriminator1D(arg))
    torch.manual_seed(1)
    x32 = torch.randn(3, 1, T).clamp(-1, 1)
    # fp64 CPU reference
    ps = {k: v.detach().double().requires_grad_(True) for k, v in m.named_parameters()}
    xr = x32.double().requires_grad_(True)
    if kind == "2d":
        h = O.mpd_fold(xr, arg)
        for li, idx in enumerate((0, 2, 4, 6, 8)):
            h = F.conv2d(h, ps[f"conv_layers.{idx}.weight"], ps[f"conv_layers.{idx}.bias"], padding=1)
            if li < 4:
                h = torch.where(h >= 0, h, h * slope)
    else:
        h = xr[:, :, :T // arg * arg].reshape(3, 1, T // arg, arg).mean(3) if arg > 1 else xr
        for li, idx in enumerate((0, 2, 4, 6, 8)):
            h = F.conv1d(h, ps[f"conv_layers.{idx}.weight"], ps[f"conv_layers.{idx}.bias"], padding=7)
            if li < 4:
                h = torch.where(h >= 0, h, h * slope)
    torch.manual_seed(2)
    r = torch.randn_like(h)
    (h * r).sum().backward()
    # HIP path
    m = m.cuda()
    x = x32.cuda().to(dtype).requires_grad_(True)
    if kind == "2d":
        x0 = Fn._MpdFold.apply(x, arg) if T % arg else x.view(3, 1, arg, T // arg)
    else:
        x0 = Fn._AvgPool.apply(x, arg) if arg > 1 else x
    y = disc_fused.disc_stack(x0, m, slope=slope)
    assert y.shape == h.shape and y.dtype == dtype
    (y.float() * r.float().cuda()).sum().backward()
    errs = {"y": O.rel_l2(y.float().cpu(), h.detach().float()), "gx": O.rel_l2(x.grad.float().cpu(), xr.grad.float())}
    for k, p in m.named_parameters():
        errs[k] = O.rel_l2(p.grad.cpu(), ps[k].grad.float())
    if slope == 1.0:
        fwd_tol, g_tol = (2e-3, 4e-3) if dtype == torch.float16 else (1.5e-2, 3e-2)
    else:
        fwd_tol, g_tol = (3e-3, 1e-1) if dtype == torch.float16 else (2e-2, 2.5e-1)   # kink-flip noise varies with atomics order
    assert errs["y"] < fwd_tol, errs
    assert all(v < g_tol for k, v in errs.items() if k != "y"), {k: f"{v:.1e}" for k, v in errs.items()}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,ks,dil,T", [(64, 64, 3, 1, 1000), (64, 64, 7, 3, 777), (64, 64, 11, 5, 1024), (192, 64, 1, 1, 300),
                                               (32, 128, 5, 2, 130), (64, 256, 15, 1, 515), (64, 20, 3, 3, 900), (64, 20, 3, 5, 257), (60, 64, 1, 1, 640), (64, 1, 11, 1, 1000)])
def test_conv1d_mfma_route_vs_fp64(H, dtype, cin, cout, ks, dil, T):
    """16-bit dense 'same' Conv1d (the GRC / fusion convs, grc_lora.py:36-41,148) on the channels-last MFMA kernels:
    forward, data gradient, weight gradient (transposed-LDS-read GEMM with W-dilation) and bias gradient against an fp64
    torch-CPU conv on the same 16-bit-rounded inputs.  The op is linear, so only rounding remains."""
    import torch.nn.functional as F
    from hifigan_modified import functional as Fn, ops
    torch.manual_seed(0)
    B = 3
    x = torch.randn(B, cin, T).to(dtype)
    w = (torch.randn(cout, cin, ks) / (cin * ks) ** 0.5).to(dtype)
    b = torch.randn(cout).to(dtype)
    r = torch.randn(B, cout, T).to(dtype)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv1d(xr, wr, br, padding=dil * (ks - 1) // 2, dilation=dil)
    (yr * r.double()).sum().backward()
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    assert ops.mfma_conv1d_ok(xd, wd, 1, dil * (ks - 1) // 2, dil, 1)
    y = Fn.conv1d(xd, wd, bd, padding=dil * (ks - 1) // 2, dilation=dil)
    (y.float() * r.cuda().float()).sum().backward()
    eps = 2e-2 if dtype == torch.bfloat16 else 3e-3
    assert O.rel_l2(y.detach().cpu(), yr.detach()) < eps
    assert O.rel_l2(xd.grad.cpu(), xr.grad) < eps
    assert O.rel_l2(wd.grad.cpu(), wr.grad) < eps
    assert O.rel_l2(bd.grad.cpu(), br.grad) < eps


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,ks,stride,pad,T", [(64, 64, 4, 2, 1, 50), (128, 64, 4, 2, 1, 33), (256, 128, 16, 8, 4, 20), (64, 32, 8, 4, 2, 40)])
def test_odconv_transpose_mfma_data_gradient_vs_fp64(H, dtype, cin, cout, ks, stride, pad, T):
    """16-bit ODConvTranspose1d (odconv.py:172-205): the data gradient runs on the fused MFMA kernel as the adjoint two-tap
    ODConv over rows of `stride` output steps (kernel_size = 2*stride), the bank gradients as per-sample two-tap MFMA
    weight-gradient GEMMs + the alpha-chain reduction.  Checked against an fp64 CPU evaluation of the reference arithmetic
    on the same 16-bit-rounded parameters and input: x.grad, kernels.grad, bias.grad and the attention-head gradients."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    m = H.ODConvTranspose1d(cin, cout, ks, stride=stride, padding=pad)
    with torch.no_grad():
        m.bias.normal_(0, 0.1)
    m = m.to(dtype)
    torch.manual_seed(1)
    x = torch.randn(3, cin, T).to(dtype)
    # fp64 reference: alpha = softmax(Wa mean_t x + ba); y = sum_k alpha_k convT(x, W_k) + alpha_k b_k
    xr = x.double().requires_grad_(True)
    att = m.kernel_attention[1]
    Wk, bk, Wa, ba = (t.detach().double().requires_grad_(True) for t in (m.kernels, m.bias, att.weight, att.bias))
    a = torch.softmax(F.conv1d(xr.mean(2, keepdim=True), Wa, ba).squeeze(-1), dim=1)
    yr = sum(a[:, k].view(-1, 1, 1) * F.conv_transpose1d(xr, Wk[k], bk[k], stride=stride, padding=pad) for k in range(Wk.shape[0]))
    torch.manual_seed(2)
    r = torch.randn_like(yr)
    (yr * r).sum().backward()
    md = m.cuda()
    xd = x.cuda().requires_grad_(True)
    y = md(xd)
    assert y.shape == yr.shape
    (y.float() * r.cuda().float()).sum().backward()
    from hifigan_modified.fused import OdconvFused
    assert OdconvFused(md).dgrad_supported()
    eps = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert O.rel_l2(y.detach().cpu(), yr.detach()) < eps
    assert O.rel_l2(xd.grad.cpu(), xr.grad) < eps
    # bank gradients and the attention chain (d alpha = <per-sample gradient, W_k>) from the MFMA per-sample GEMMs
    assert O.rel_l2(md.kernels.grad.cpu(), Wk.grad) < eps
    assert O.rel_l2(md.bias.grad.cpu(), bk.grad) < eps
    assert O.rel_l2(md.kernel_attention[1].weight.grad.cpu(), Wa.grad) < 3 * eps
    assert O.rel_l2(md.kernel_attention[1].bias.grad.cpu(), ba.grad) < 3 * eps


def test_second_design_blocks_grads(H):
    """GroupedResidualConv1D (generator.py:112-172) and FeatureWiseLinearModulation (generator.py:177-199) against the
    reference's gradient goldens: input gradients (for film2 also both embeddings) and every parameter gradient, through the
    folded dense-kernel form on the HIP backward kernels."""
    g = load_golden("grouped_residual_64_k3_d3")
    m = load_sd(H.GroupedResidualConv1D(64, 3, 3), g)
    x = gin(g, "x.x")
    check_all(g, m, m(x), {"x": x})
    g = load_golden("film2_448_64")
    m = load_sd(H.FeatureWiseLinearModulation(448, 64), g)
    x, spk, emo = gin(g, "x.x"), gin(g, "x.spk"), gin(g, "x.emo")
    check_all(g, m, m(x, spk, emo), {"x": x, "spk": spk, "emo": emo})


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("args,kw,T", [((64, 64), {}, 700), ((32, 32), dict(dilations=[1, 2], groups=2, r=4), 333)])
def test_mrf_merged_conv_training_path_vs_fp64_oracle(H, dtype, args, kw, T):
    """16-bit training of MultiReceptiveFieldBlock: the three folded branch convs and their residual projections run as ONE
    dense conv (kernel 2*max(d)+1, zeros on the unused taps) on the MFMA kernels.  Output, input gradient and every parameter
    gradient against the fp64 autograd of the oracle on the same (16-bit-rounded) input.  SiLU/GroupNorm are smooth, so the
    differences are 16-bit rounding (bf16: a few 1e-2 on the small LoRA gradients)."""
    from hifigan_modified import functional as Fn
    torch.manual_seed(0)
    m = H.MultiReceptiveFieldBlock(*args, **kw).train(False)
    sd = {k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    torch.manual_seed(1)
    x = torch.randn(3, args[0], T).to(dtype)
    xr = x.double().requires_grad_(True)
    yr = O.mrf_block(xr, sd, "", tuple(kw.get("dilations", [1, 3, 5])))
    torch.manual_seed(2)
    r = torch.randn_like(yr)
    (yr * r).sum().backward()
    md = m.cuda()
    xd = x.cuda().requires_grad_(True)
    assert Fn._mrf_merged_ok(xd, md)
    y = md(xd)
    (y.float() * r.cuda().float()).sum().backward()
    eps = 4e-2 if dtype == torch.bfloat16 else 6e-3
    assert O.rel_l2(y.detach().cpu(), yr.detach()) < eps
    assert O.rel_l2(xd.grad.cpu(), xr.grad) < eps
    for k, p in md.named_parameters():
        ref = sd[k].grad
        assert p.grad is not None and ref is not None, k
        # lora_scaling is ONE scalar: a cancelling sum over the whole folded kernel, where 16-bit rounding of the operands
        # leaves 10 % in fp16 and 24 % in bf16 (measured identically on the per-branch path, so it is rounding, not the merge)
        tol = (0.5 if dtype == torch.bfloat16 else 0.2) if ref.numel() == 1 else (3 * eps if "lora" in k else eps)
        assert O.rel_l2(p.grad.cpu(), ref) < tol, (k, O.rel_l2(p.grad.cpu(), ref))


def _philox4x32_10_np(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11) on uint32 numpy arrays: the published algorithm, restated here as the test's checker."""
    c = [np.asarray(v, dtype=np.uint64) for v in ctr]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M = np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M, p0 & M]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return [v.astype(np.uint32) for v in c]


def test_philox_restatement_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds.
    z = _philox4x32_10_np([[0], [0], [0], [0]], (0, 0))
    assert [int(v[0]) for v in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    z = _philox4x32_10_np([[f], [f], [f], [f]], (f, f))
    assert [int(v[0]) for v in z] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    z = _philox4x32_10_np([[0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344]], (0xa4093822, 0x299f31d0))
    assert [int(v[0]) for v in z] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("n,p", [(8 * 4096 + 5, 0.1), (3, 0.5), (1 << 20, 0.25), (64, 0.0)])
def test_dropout_mask_kernel_is_philox_bit_exact(H, n, p):
    """mv_dropout_mask: bit-exact against the numpy Philox restatement (counter = index/8, 16 bits per element)."""
    import ctypes
    seed = 0x1234567_89abcdef
    m = torch.empty(n, device="cuda", dtype=torch.uint8)
    from hifigan_modified import _native as N_, ops as ops_
    N_.call("mv_dropout_mask", ctypes.c_void_p(m.data_ptr()), n, float(p), seed, ops_._stream())
    nblk = (n + 7) // 8
    blk = np.arange(nblk, dtype=np.uint64)
    r = _philox4x32_10_np([blk & np.uint64(0xffffffff), blk >> np.uint64(32), np.zeros(nblk), np.zeros(nblk)],
                          (seed & 0xffffffff, seed >> 32))
    u16 = np.stack([(r[e >> 1] >> np.uint32(16 * (e & 1))) & np.uint32(0xffff) for e in range(8)], axis=1).reshape(-1)[:n]
    ref = (u16 >= np.uint32(int(p * 65536 + 0.5))).astype(np.uint8)
    assert np.array_equal(m.cpu().numpy(), ref)
    if n >= 1 << 20:
        assert abs(ref.mean() - (1 - p)) < 3e-3


def test_dropout_mask_follows_torch_manual_seed(H):
    from hifigan_modified import functional as F_
    torch.manual_seed(7); a = F_.dropout_mask((4, 64, 100), 0.1, "cuda"); b = F_.dropout_mask((4, 64, 100), 0.1, "cuda")
    torch.manual_seed(7); a2 = F_.dropout_mask((4, 64, 100), 0.1, "cuda")
    assert torch.equal(a, a2) and not torch.equal(a, b)
    assert abs(a.float().mean().item() - 0.9) < 0.02


@pytest.mark.parametrize("n_fft,hop,n_mels,sr,T", [(1024, 256, 80, 22050, 4096), (2048, 512, 128, 48000, 8192), (64, 16, 20, 22050, 256),
                                                   (960, 240, 80, 22050, 1920), (48, 16, 10, 16000, 160)])
@pytest.mark.parametrize("kind", ["l1", "mse"])
def test_mel_loss_fft_and_dft_paths_vs_oracle(H, n_fft, hop, n_mels, sr, T, kind):
    """Power-of-two n_fft runs the in-LDS radix-2 rFFT (forward and adjoint), anything else the direct DFT: both against the oracle's
    explicit-DFT definition in fp64, value and d loss / d wave."""
    from hifigan_modified import functional as Fn
    from hifigan_modified.mel import mel_filterbank
    torch.manual_seed(n_fft + T)
    wave = (torch.randn(2, 1, T) * 0.3).clamp(-1, 1)
    target = torch.randn(2, n_mels, T // hop)
    kw = dict(sr=sr, n_fft=n_fft, hop=hop, n_mels=n_mels, fmax=sr / 2)
    w_ref = wave.clone().double().requires_grad_(True)
    d = O.mel_spectrogram(w_ref, **kw) - target.double()
    ref = d.abs().mean() if kind == "l1" else (d * d).mean()
    ref.backward()
    fb = mel_filterbank(sr, n_fft, n_mels, 0.0, sr / 2, device="cuda")
    w = wave.cuda().requires_grad_(True)
    loss = (Fn.mel_l1 if kind == "l1" else Fn.mel_mse)(w, target.cuda(), fb, n_fft=n_fft, hop=hop)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
    assert O.rel_l2(w.grad.cpu(), w_ref.grad.float()) < 2e-3
    mel = Fn.mel_spectrogram(wave.cuda(), fb, n_fft=n_fft, hop=hop)
    assert O.rel_l2(mel.cpu(), O.mel_spectrogram(wave.double(), **kw).float()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H_,W,kh,kw", [(7, 300, 3, 3), (2, 129, 3, 3), (1, 515, 1, 15), (11, 128, 3, 3)])
def test_head_partial_sums_from_the_conv_epilogue(H, dtype, H_, W, kh, kw):
    """mv_dconv_cl_fwd_head (last hidden conv 128->256 + the head's per-tap partial sums out of its LDS epilogue tile, then mv_dhead_sum)
    against the separate launches mv_dconv_cl_fwd + mv_dhead_fwd (discriminators.py:63-65 / :104-106): same activation, and - the
    MFMAs see the same rounded operands in the same order - the same head output bit for bit."""
    from ctypes import c_void_p
    from hifigan_modified import _native as N, ops, disc_fused
    torch.manual_seed(3)
    B, Cin, Cout = 3, 128, 256
    x = torch.randn(B, H_, W, Cin, device="cuda").to(dtype)
    w4 = torch.nn.Parameter(torch.randn(Cout, Cin, kh, kw, device="cuda") / (Cin * kh * kw) ** 0.5)
    b4 = (torch.randn(Cout, device="cuda") * 0.1).to(dtype)
    w5 = torch.nn.Parameter(torch.randn(1, Cout, kh, kw, device="cuda") / (Cout * kh * kw) ** 0.5)
    b5 = torch.randn(1, device="cuda").to(dtype)
    P = lambda t: c_void_p(t.data_ptr())
    pk = disc_fused._packs.get(w4, dtype, 0)
    hp = disc_fused._packs.head_mfma(w5, dtype)
    dt, st = ops._dt(x), ops._stream()
    y0 = torch.empty(B, H_, W, Cout, device="cuda", dtype=dtype)
    z0 = torch.zeros(16, B * H_ * W, device="cuda")
    o0 = torch.empty(B, H_, W, device="cuda", dtype=dtype)
    N.call("mv_dconv_cl_fwd", P(x), P(pk), P(b4), None, P(y0), B, H_, W, Cin, Cout, kh, kw, 1, N.ACT_LRELU, 0.1, dt, st)
    N.call("mv_dhead_fwd", P(y0), P(hp), P(b5), P(z0), P(o0), B, H_, W, Cout, kh, kw, dt, st)
    y1 = torch.empty_like(y0)
    z1 = torch.zeros_like(z0)
    o1 = torch.empty_like(o0)
    rc = N.lib().mv_dconv_cl_fwd_head(P(x), P(pk), P(b4), P(y1), P(hp), P(z1), kh, kw, B, H_, W, Cin, Cout, kh, kw, N.ACT_LRELU, 0.1, dt, st)
    assert rc == 0, rc
    N.call("mv_dhead_sum", P(z1), P(b5), P(o1), B, H_, W, kh, kw, dt, st)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(z0[:kh * kw], z1[:kh * kw])
    assert torch.equal(o0, o1)
