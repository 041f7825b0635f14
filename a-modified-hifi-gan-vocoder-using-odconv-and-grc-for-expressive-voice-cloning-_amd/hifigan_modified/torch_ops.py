"""PyTorch custom operators of the MI355X vocoder path: namespace ``mi355x_vocoder`` (``torch.ops.mi355x_vocoder.*``).

north_star asks for the HIP kernels to be "exposed to Python as PyTorch-ROCm custom ops"; SURVEY.md section 8(b) lists the
schemas.  Every operator below is a thin shim: its CUDA implementation calls the same C-ABI cores (through ``ops.py`` /
``fused.py`` / ``disc_fused.py``) that the ctypes binding of INTEGRATION.md calls, and every differentiable one has its
backward registered with ``torch.library.register_autograd`` - the backward formulas are the HIP backward kernels of
``functional.py`` (one definition: the operator's forward/backward ARE the ``forward`` / ``backward`` static methods of
the corresponding ``torch.autograd.Function`` class, run with a recording context).

The drop-in modules (``ODConv1d``, ``ODConvTranspose1d``, ``GRC_LoRA_Block``, ``MultiReceptiveFieldBlock``, ``FiLMLayer``, the
discriminators, the loss helpers of the trainers, ``FlatAdamW``) dispatch through these operators (``functional.py``).
The CPU "kernel" of every operator raises: this path has no CPU / PyTorch fallback.

Operators (schemas as registered; ``act``: 0 none, 1 LeakyReLU, 2 tanh, 3 SiLU; ``fused`` / ``handle``: 0 or a handle from
``object_handle`` naming a packed-weight cache object - Python objects cannot cross the dispatcher):
  odconv_attn, odconv1d, odconv_transpose1d          odconv.py:36-40, :73-108, :172-205
  conv1d, conv2d, group_norm, film                   grc_lora.py:32-68,108-129,157-163; discriminators.py:56-66,97-107
  grc_mrf_block, generator_forward                   fused inference: grc_lora.py:157-163, SURVEY.md appendix A
  avg_pool1d, mpd_fold, disc_conv_stack              discriminators.py:72-79,94,56-66,97-107
  gan_loss, mel_loss, mel_spectrogram                complete_vocoder.py:89-184, conditioned_hifigan.py:262-265; DESIGN.md section 2
  fused_adamw_                                       conditioned_hifigan.py:219 (torch.optim.AdamW semantics)
"""
from __future__ import annotations

import threading
import weakref

import torch

from . import _native as N
from . import ops

NS = "mi355x_vocoder"
_LIB = torch.library.Library(NS, "DEF")
_tls = threading.local()

# ------------------------------------------------------------------------------------------------ object handles
_HANDLES = {}


def object_handle(obj) -> int:
    """Integer name of a Python-side cache object (packed weights of one module) that an operator argument can carry."""
    if obj is None:
        return 0
    h = id(obj)
    if h not in _HANDLES:
        try:
            _HANDLES[h] = weakref.ref(obj, lambda _r, h=h: _HANDLES.pop(h, None))
        except TypeError:
            _HANDLES[h] = lambda obj=obj: obj
    return h


def _obj(h):
    if not h:
        return None
    r = _HANDLES.get(h)
    o = r() if r is not None else None
    if o is None:
        raise RuntimeError(f"{NS}: stale object handle {h}")
    return o


# ------------------------------------------------------------------------------------------------ generic adapter
class _Ctx:
    """Stands in for the autograd context while a Function's forward / backward runs inside an operator."""

    def __init__(self):
        self.saved_tensors = ()
        self.needs_input_grad = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def set_materialize_grads(self, v):
        pass


def _refuse_cpu(name):
    def refuse(*args):
        raise RuntimeError(f"{NS}::{name}: tensors must live on the MI355X (cuda) - this path has no CPU fallback")
    return refuse


def _register(name, schema, cls, pack, fmap):
    """Define `name` with `schema`; CUDA impl = cls.forward(recording ctx, *pack(*args)); backward = cls.backward.
    fmap[j] = index of the operator argument that the Function's j-th forward argument is (None: synthesised, e.g. a cfg tuple)."""
    _LIB.define(name + schema)

    def impl(*args):
        rec = _Ctx()
        out = cls.forward(rec, *pack(*args))
        if torch.is_tensor(out) and any(torch.is_tensor(a) and a.data_ptr() == out.data_ptr() and a.numel() for a in args):
            out = out.clone()        # the schema promises a fresh tensor (e.g. the MPD fold of a length that divides evenly is a view)
        # the record is what setup_context (which runs right after this, in this thread, iff an input requires grad and grad mode is
        # on) turns into the autograd context; when no argument requires a gradient nothing is kept alive.  (Grad mode itself cannot
        # be asked here: the dispatcher runs this kernel below the autograd key with grad mode off.)
        need = any(torch.is_tensor(a) and a.requires_grad for a in args)
        _tls.last = (rec, out) if need else None
        return out

    _LIB.impl(name, impl, "CUDA")
    _LIB.impl(name, _refuse_cpu(name), "CPU")

    def setup(ctx, inputs, output):
        rec, out = getattr(_tls, "last", None) or (None, None)
        _tls.last = None
        # paired by identity: the very storage the forward returned (not just a tensor of the same shape)
        if rec is None or (torch.is_tensor(output) and (out.data_ptr() != output.data_ptr() or tuple(out.shape) != tuple(output.shape))):
            raise RuntimeError(f"{NS}::{name}: forward record does not belong to this call")
        ctx.mv_attrs = {k: v for k, v in rec.__dict__.items() if k not in ("saved_tensors", "needs_input_grad")}
        ctx.mv_nsaved = len(rec.saved_tensors)
        ctx.save_for_backward(*rec.saved_tensors)

    def backward(ctx, *grads):
        shim = _Ctx()
        shim.__dict__.update(ctx.mv_attrs)
        shim.saved_tensors = ctx.saved_tensors
        need_op = ctx.needs_input_grad
        shim.needs_input_grad = tuple(False if i is None else bool(need_op[i]) for i in fmap)
        res = cls.backward(shim, *grads)
        res = res if isinstance(res, tuple) else (res,)
        out = [None] * len(need_op)
        for j, i in enumerate(fmap):
            if i is not None and j < len(res) and need_op[i]:
                out[i] = res[j]
        return tuple(out)

    torch.library.register_autograd(f"{NS}::{name}", backward, setup_context=setup, lib=_LIB)


def _forward_only(name, schema, fn):
    """Inference-only operator: asking for a gradient through it raises instead of silently returning zeros."""
    _LIB.define(name + schema)
    _LIB.impl(name, fn, "CUDA")
    _LIB.impl(name, _refuse_cpu(name), "CPU")

    def backward(ctx, *grads):
        raise NotImplementedError(f"{NS}::{name} is an inference operator (no backward); the modules use the differentiable "
                                  "operators when a gradient is required")

    torch.library.register_autograd(f"{NS}::{name}", backward, setup_context=lambda ctx, inputs, output: None, lib=_LIB)


def _install():
    from . import functional as Fn
    from . import disc_fused

    # ---- ODConv (odconv.py:73-108, :172-205)
    od_schema = ("(Tensor x, Tensor kernels, Tensor bias, Tensor att_w, Tensor att_b, int stride, int padding, int output_padding, "
                 "int dilation, int act, float slope, int fused) -> Tensor")
    od_map = [0, 1, 2, 3, 4, None, None]
    _register("odconv1d", od_schema, Fn._ODConv,
              lambda x, k, b, aw, ab, s, p, op, d, act, sl, fz: (x, k, b, aw, ab, (False, s, p, 0, d, act, sl), _obj(fz)), od_map)
    _register("odconv_transpose1d", od_schema, Fn._ODConv,
              lambda x, k, b, aw, ab, s, p, op, d, act, sl, fz: (x, k, b, aw, ab, (True, s, p, op, d, act, sl), _obj(fz)), od_map)
    _forward_only("odconv_attn", "(Tensor x, Tensor att_w, Tensor att_b) -> Tensor", lambda x, aw, ab: Fn.odconv_attention(x, aw, ab))

    # ---- plain convolutions, GroupNorm, FiLM
    _register("conv1d", "(Tensor x, Tensor weight, Tensor? bias, int stride, int padding, int dilation, int groups, int act, float slope) -> Tensor",
              Fn._Conv1d, lambda x, w, b, s, p, d, g, act, sl: (x, w, b, (s, p, d, g, act, sl)), [0, 1, 2, None])
    _register("conv2d", "(Tensor x, Tensor weight, Tensor? bias, int pad_h, int pad_w, int act, float slope) -> Tensor",
              Fn._Conv2d, lambda x, w, b, ph, pw, act, sl: (x, w, b, ((ph, pw), act, sl)), [0, 1, 2, None])
    _register("group_norm", "(Tensor x, Tensor weight, Tensor bias, Tensor? res, Tensor? mask, int groups, float eps, int act, float slope, "
              "float mask_scale) -> Tensor", Fn._GroupNorm,
              lambda x, w, b, res, mask, G, eps, act, sl, ms: (x, w, b, res, mask, (G, eps, act, sl, ms)), [0, 1, 2, 3, 4, None])
    _register("film", "(Tensor x, Tensor cond, Tensor proj_w, Tensor proj_b, int feature_dim) -> Tensor", Fn._Film,
              lambda x, c, w, b, F: (x, c, w, b, F), [0, 1, 2, 3, 4])

    # ---- discriminator pieces
    _register("avg_pool1d", "(Tensor x, int scale) -> Tensor", Fn._AvgPool, lambda x, s: (x, s), [0, 1])
    _register("mpd_fold", "(Tensor x, int period) -> Tensor", Fn._MpdFold, lambda x, p: (x, p), [0, 1])
    names = ", ".join(f"Tensor w{i}, Tensor b{i}" for i in range(1, 6))
    _register("disc_conv_stack", f"(Tensor x, float slope, {names}) -> Tensor", disc_fused._DiscStack,
              lambda x, slope, *params: (x, slope, *params), list(range(12)))

    # ---- losses
    _register("gan_loss", "(Tensor x, Tensor? y, int kind, float c, float weight) -> Tensor", Fn._Loss,
              lambda x, y, kind, c, w: (x, y, kind, c, w), [0, 1, 2, 3, 4])
    _register("mel_loss", "(Tensor wave, Tensor target, Tensor fb, int n_fft, int hop, float clampv, float weight, int kind) -> Tensor",
              Fn._MelL1, lambda *a: a, list(range(8)))
    _forward_only("mel_spectrogram", "(Tensor wave, Tensor fb, int n_fft, int hop, float clampv) -> Tensor",
                  lambda wave, fb, n_fft, hop, clampv: ops.mel_loss(wave, fb, None, n_fft, hop, clampv, 1.0, backward=False, want_mel=True)[1])

    # ---- fused inference operators
    def grc_mrf_block(x, handle):
        return ops.ntc_to_nct(_obj(handle).forward_cl(ops.nct_to_ntc(x)))
    _forward_only("grc_mrf_block", "(Tensor x, int handle) -> Tensor", grc_mrf_block)

    def generator_forward(mel, spk, emo, handle):
        return _obj(handle).forward(mel, spk, emo, cache=Fn._cache)
    _forward_only("generator_forward", "(Tensor mel, Tensor? speaker_emb, Tensor? emotion_emb, int handle) -> Tensor", generator_forward)

    # ---- optimizer
    _LIB.define("fused_adamw_(Tensor(a!) p, Tensor g, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, float beta2, "
                "float eps, float weight_decay, int step, float grad_scale) -> ()")

    def fused_adamw_(p, g, m, v, lr, b1, b2, eps, wd, step, gscale):
        from ctypes import c_void_p
        if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()) or \
                any(t.dtype != torch.float32 or t.numel() != p.numel() for t in (p, g, m, v)):
            raise RuntimeError(f"{NS}::fused_adamw_: p, g, exp_avg, exp_avg_sq must be contiguous fp32 tensors of one size")
        N.call("mv_adamw_flat", c_void_p(p.data_ptr()), c_void_p(g.data_ptr()), c_void_p(m.data_ptr()), c_void_p(v.data_ptr()),
               p.numel(), float(lr), float(b1), float(b2), float(eps), float(wd), int(step), float(gscale), ops._stream())
    _LIB.impl("fused_adamw_", fused_adamw_, "CUDA")
    _LIB.impl("fused_adamw_", _refuse_cpu("fused_adamw_"), "CPU")

    # the step count as a device tensor (int32, already incremented): what a captured training step replays
    _LIB.define("fused_adamw_dev_(Tensor(a!) p, Tensor g, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, float beta2, "
                "float eps, float weight_decay, Tensor step, float grad_scale) -> ()")

    def fused_adamw_dev_(p, g, m, v, lr, b1, b2, eps, wd, step, gscale):
        from ctypes import c_void_p
        if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()) or \
                any(t.dtype != torch.float32 or t.numel() != p.numel() for t in (p, g, m, v)):
            raise RuntimeError(f"{NS}::fused_adamw_dev_: p, g, exp_avg, exp_avg_sq must be contiguous fp32 tensors of one size")
        if step.dtype != torch.int32 or step.numel() != 1 or not step.is_cuda:
            raise RuntimeError(f"{NS}::fused_adamw_dev_: step must be a one-element int32 GPU tensor")
        N.call("mv_adamw_flat_dev", c_void_p(p.data_ptr()), c_void_p(g.data_ptr()), c_void_p(m.data_ptr()), c_void_p(v.data_ptr()),
               p.numel(), float(lr), float(b1), float(b2), float(eps), float(wd), c_void_p(step.data_ptr()), float(gscale), ops._stream())
    _LIB.impl("fused_adamw_dev_", fused_adamw_dev_, "CUDA")
    _LIB.impl("fused_adamw_dev_", _refuse_cpu("fused_adamw_dev_"), "CPU")


def _install_fakes():
    """Shape / dtype functions (torch.library.register_fake): FakeTensor tracing (torch.compile, torch.export, opcheck) sees through
    every operator without running a kernel.  Formulas = the reference modules' (nn.Conv1d / ConvTranspose1d / AvgPool1d output sizes,
    discriminators.py:72-79 for the fold)."""
    fake = lambda name: torch.library.register_fake(f"{NS}::{name}", lib=_LIB)

    @fake("odconv1d")
    def _(x, kernels, bias, att_w, att_b, stride, padding, output_padding, dilation, act, slope, fused):
        ks = kernels.shape[3]
        return x.new_empty(x.shape[0], kernels.shape[1], (x.shape[2] + 2 * padding - dilation * (ks - 1) - 1) // stride + 1)

    @fake("odconv_transpose1d")
    def _(x, kernels, bias, att_w, att_b, stride, padding, output_padding, dilation, act, slope, fused):
        ks = kernels.shape[3]
        return x.new_empty(x.shape[0], kernels.shape[2], (x.shape[2] - 1) * stride - 2 * padding + dilation * (ks - 1) + output_padding + 1)

    @fake("odconv_attn")
    def _(x, att_w, att_b):
        return x.new_empty(x.shape[0], att_w.shape[0], dtype=torch.float32)

    @fake("conv1d")
    def _(x, weight, bias, stride, padding, dilation, groups, act, slope):
        ks = weight.shape[2]
        return x.new_empty(x.shape[0], weight.shape[0], (x.shape[2] + 2 * padding - dilation * (ks - 1) - 1) // stride + 1)

    @fake("conv2d")
    def _(x, weight, bias, pad_h, pad_w, act, slope):
        return x.new_empty(x.shape[0], weight.shape[0], x.shape[2] + 2 * pad_h - weight.shape[2] + 1, x.shape[3] + 2 * pad_w - weight.shape[3] + 1)

    @fake("group_norm")
    def _(x, weight, bias, res, mask, groups, eps, act, slope, mask_scale):
        return torch.empty_like(x)

    @fake("film")
    def _(x, cond, proj_w, proj_b, feature_dim):
        return torch.empty_like(x)

    @fake("avg_pool1d")
    def _(x, scale):
        return x.new_empty(x.shape[0], x.shape[1], x.shape[2] // scale)

    @fake("mpd_fold")
    def _(x, period):
        return x.new_empty(x.shape[0], x.shape[1], period, (x.shape[2] + period - 1) // period)

    def _stack(x, slope, *params):
        return x.new_empty((x.shape[0], 1) + tuple(x.shape[2:]))
    torch.library.register_fake(f"{NS}::disc_conv_stack", _stack, lib=_LIB)

    @fake("gan_loss")
    def _(x, y, kind, c, weight):
        return x.new_empty((), dtype=torch.float32)

    @fake("mel_loss")
    def _(wave, target, fb, n_fft, hop, clampv, weight, kind):
        return wave.new_empty((), dtype=torch.float32)

    @fake("mel_spectrogram")
    def _(wave, fb, n_fft, hop, clampv):
        return wave.new_empty(wave.shape[0], fb.shape[0], wave.shape[-1] // hop, dtype=torch.float32)

    @fake("grc_mrf_block")
    def _(x, handle):
        return torch.empty_like(x)

    @fake("generator_forward")
    def _(mel, speaker_emb, emotion_emb, handle):
        total = 1
        for f in _obj(handle).gen.upsample_factors:
            total *= f
        return mel.new_empty(mel.shape[0], 1, mel.shape[2] * total)

    @fake("fused_adamw_")
    def _(p, g, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale):
        return None

    @fake("fused_adamw_dev_")
    def _(p, g, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale):
        return None


_install()
_install_fakes()
OPS = torch.ops.mi355x_vocoder
