"""The kernels that the size tables select ONLY at the training shapes (B = 64 = real + fake of 32 clips x 8192 samples), checked at
those shapes, and one full-size training step (BASELINE configs[2] / SURVEY C3).

Reference arithmetic: discriminators.py:56-66 (Conv2d 3x3 pad 1 + LeakyReLU 0.1, periods 2..11) and :97-107 (Conv1d k15 pad 7),
their data gradients (the same kernels on flipped weights with LeakyReLU' of the saved activation) and weight gradients; compared on
a 2-sample slice against fp32 torch convolutions / autograd of the same bf16-rounded operands (bounds: bf16 operand rounding of the
OUTPUT only, 4e-3; weight gradients accumulate in fp32, 2e-3).  These are the shapes at which dconv_cl_wide_kernel<bf16,1,4,8,64,4>
(128->256 forward), <bf16,1,4,8,32,2> (256->128 data gradient), the XCD tile order, the 256-row k15 variants and the row-ring 3x3
weight gradient (dconv_wgrad3_kernel<bf16,4>) are picked; smaller test shapes take other instantiations."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DT = torch.bfloat16
B, T = 64, 8192


@pytest.fixture(scope="module")
def ops():
    from hifigan_modified import ops, _native
    _native.lib()
    return ops


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.mark.parametrize("P", [2, 7, 11])
@pytest.mark.parametrize("kind,cin,cout", [("fwd", 128, 256), ("dgrad", 256, 128), ("fwd", 64, 128), ("dgrad", 128, 64)])
def test_conv3x3_training_shapes(ops, P, kind, cin, cout):
    from hifigan_modified import _native as N
    Hh, W = P, T // P
    torch.manual_seed(0)
    x = torch.randn(B, Hh, W, cin, device="cuda").to(DT)
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    pk = ops.dconv_pack(w, DT, 0)
    sv = torch.randn(B, Hh, W, cout, device="cuda").to(DT) if kind == "dgrad" else None
    y = ops.dconv_cl(x, pk, None, cout, 3, 3, 1, N.ACT_NONE, 0.1, sv)
    for sl in (slice(0, 1), slice(B - 1, B)):             # first and last sample: both ends of the (XCD-remapped) grid
        ref = torch.nn.functional.conv2d(x[sl].permute(0, 3, 1, 2).float(), w.to(DT).float(), padding=1).permute(0, 2, 3, 1)
        if sv is not None:
            ref = torch.where(sv[sl].float() >= 0, ref, ref * 0.1)
        assert _rel(y[sl], ref) < 4e-3, (P, kind, cin, cout, _rel(y[sl], ref))


@pytest.mark.parametrize("Tn", [8192, 4096, 2048])
@pytest.mark.parametrize("kind,cin,cout", [("fwd", 128, 256), ("dgrad", 256, 128), ("fwd", 64, 128), ("dgrad", 128, 64), ("fwd", 32, 64)])
def test_conv_k15_training_shapes(ops, Tn, kind, cin, cout):
    from hifigan_modified import _native as N
    torch.manual_seed(0)
    x = torch.randn(B, 1, Tn, cin, device="cuda").to(DT)
    w = torch.randn(cout, cin, 1, 15, device="cuda") / (cin * 15) ** 0.5
    pk = ops.dconv_pack(w, DT, 0)
    sv = torch.randn(B, 1, Tn, cout, device="cuda").to(DT) if kind == "dgrad" else None
    y = ops.dconv_cl(x, pk, None, cout, 1, 15, 1, N.ACT_NONE, 0.1, sv)
    for sl in (slice(0, 1), slice(B - 1, B)):
        ref = torch.nn.functional.conv2d(x[sl].permute(0, 3, 1, 2).float(), w.to(DT).float(), padding=(0, 7)).permute(0, 2, 3, 1)
        if sv is not None:
            ref = torch.where(sv[sl].float() >= 0, ref, ref * 0.1)
        assert _rel(y[sl], ref) < 4e-3, (Tn, kind, cin, cout, _rel(y[sl], ref))


@pytest.mark.parametrize("P", [2, 7, 11])
@pytest.mark.parametrize("cin,cout", [(128, 256), (64, 128), (32, 64)])
def test_wgrad3x3_training_shapes(ops, P, cin, cout):
    """Row-ring 3x3 weight gradient over the whole 64-sample batch against torch autograd on fp32 copies of the same bf16 operands
    (the contraction runs over all 64 x P x T/P positions, so the whole batch is the reference)."""
    Hh, W = P, T // P
    torch.manual_seed(1)
    x = torch.randn(B, Hh, W, cin, device="cuda").to(DT)
    g = (torch.randn(B, Hh, W, cout, device="cuda") / 8).to(DT)
    gw, gb = ops.dconv_wgrad_cl(x, g, 3, 3, 1, want_bias=True)
    w = torch.zeros(cout, cin, 3, 3, device="cuda", requires_grad=True)
    bias = torch.zeros(cout, device="cuda", requires_grad=True)
    step = 16                                              # fp32 torch conv over the batch in slices (memory)
    for i in range(0, B, step):
        y = torch.nn.functional.conv2d(x[i:i + step].permute(0, 3, 1, 2).float(), w, bias, padding=1)
        y.backward(g[i:i + step].permute(0, 3, 1, 2).float())
    assert _rel(gw, w.grad) < 2e-3 and _rel(gb, bias.grad) < 2e-3, (P, cin, cout, _rel(gw, w.grad), _rel(gb, bias.grad))


def test_full_size_train_step_c3():
    """BASELINE configs[2] at its full size: VocoderTrainer.train_step on the default generator + MPD + MSD, B = 32 clips x 8192
    samples, bf16 activations (what bench.py times).  Finite losses, and the loss dict of the first step (same weights, same batch)
    agrees with the fp32-activation step: D 1 %, G 3 %, mel 3 % (bf16 rounding of activations at random init; measured 0.2 % / 0.6 % /
    0.9 %).  Then the captured form of the same step replays to the same losses as an eager step from the same state."""
    import hifigan_modified as H
    torch.manual_seed(1)
    mel, real = torch.randn(32, 80, 32).cuda(), torch.randn(32, 1, 8192).clamp(-1, 1).cuda()
    spk, emo = torch.randn(32, 192).cuda(), torch.randn(32, 384).cuda()
    out = {}
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        torch.manual_seed(0)
        voc = H.ModifiedHiFiGANVocoder(dropout=0.0)
        tr = H.VocoderTrainer(voc, device=torch.device("cuda"))
        out[tag] = tr.train_step(mel.to(dt), real.to(dt), spk.to(dt), emo.to(dt))
        assert all(v == v and abs(v) < 1e6 for v in out[tag].values()), (tag, out[tag])
        if tag == "bf16":
            second = tr.train_step(mel.to(dt), real.to(dt), spk.to(dt), emo.to(dt))
            assert all(v == v and abs(v) < 1e6 for v in second.values())
            assert second["discriminator_loss"] < out[tag]["discriminator_loss"]      # one AdamW step on the same batch lowers the D loss
        del tr, voc
        torch.cuda.empty_cache()
    print(f"[train] full-size first step: bf16 {out['bf16']}  fp32 {out['fp32']}")
    for key, tol in (("discriminator_loss", 0.01), ("generator_loss", 0.03), ("mel_loss", 0.03)):
        assert abs(out["bf16"][key] - out["fp32"][key]) <= tol * abs(out["fp32"][key]), (key, out)
