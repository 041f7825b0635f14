"""Flat-arena AdamW (torch.optim.AdamW semantics: conditioned_hifigan.py:219, train_config.yaml:65-68).

All trainable parameters of a group live in ONE fp32 buffer (each nn.Parameter becomes a view of it), so an
optimizer step is a single HIP launch (mv_adamw_flat) and the data-parallel exchange is an all-reduce of ONE flat
gradient buffer, cut into buckets.  Gradients produced by backward are gathered into the flat buffer by one
multi-tensor launch (mv_multi_gather).  Parameters that never receive a gradient (the unused ODConv attention heads,
SURVEY.md §5) are left out of the arena exactly like torch.optim skips ``grad is None``.
"""
from __future__ import annotations

from ctypes import c_void_p

import numpy as np
import torch

from . import _native as N
from . import ops

_DESC = np.dtype([("src", np.uint64), ("dst_off", np.int64), ("n", np.int64), ("dtype", np.int32), ("pad", np.int32)])


class FlatAdamW:
    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, exclude=()):
        excl = {id(p) for p in exclude}
        self.params = [p for p in params if p.requires_grad and id(p) not in excl]
        if not self.params:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise RuntimeError("FlatAdamW needs fp32 parameters on one GPU (move the model first)")
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8          # keep every view (and its 16-bit shadow) 16-byte aligned
        self.numel = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                ops.copy_rows(p.detach().reshape(1, 1, -1), view.view(1, 1, -1))
                p.data = view
        self.step_count = 0
        # 16-bit shadows of the arena (one per activation dtype in use): refreshed by ONE cast launch per optimizer step, so
        # the ~200 per-parameter casts of a training step disappear; ops._ParamCache hands out views of them
        self._shadows = {}
        self._shadow_state = None            # (param epoch, per-parameter _version list) the shadows correspond to
        self._index = {id(p): i for i, p in enumerate(self.params)}
        ops.register_shadow_owner(self, self.params)
        self._descs = np.zeros(len(self.params), dtype=_DESC)
        self._descs["dst_off"] = self.offsets
        self._descs["n"] = [p.numel() for p in self.params]
        self._max_len = int(self._descs["n"].max())

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """param.grad tensors (fresh per backward) -> flat fp32 buffer; missing grads contribute zeros."""
        d = self._descs
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                d["src"][i], d["dtype"][i] = 0, 0
            else:
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g
                d["src"][i], d["dtype"][i] = g.data_ptr(), ops._DT[g.dtype]
        table = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(self.flat_g.device, non_blocking=True)
        N.call("mv_multi_gather", c_void_p(table.data_ptr()), len(self.params), self._max_len,
               c_void_p(self.flat_g.data_ptr()), ops._stream())
        self._table = table   # keep alive until the launch has consumed it
        return self.flat_g

    def step(self, grad_scale=1.0, gathered=False):
        if not gathered:
            self.gather_grads()
        self.step_count += 1
        N.call("mv_adamw_flat", c_void_p(self.flat_p.data_ptr()), c_void_p(self.flat_g.data_ptr()),
               c_void_p(self.exp_avg.data_ptr()), c_void_p(self.exp_avg_sq.data_ptr()), self.numel, float(self.lr),
               float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
               int(self.step_count), float(grad_scale), ops._stream())
        ops.bump_param_epoch(self)       # in-place arena update: cached casts / packed weights of THESE parameters must refresh
        self._refresh_shadows()

    def _refresh_shadows(self):
        for dt, sh in self._shadows.items():
            N.call("mv_cast", c_void_p(self.flat_p.data_ptr()), N.MV_F32, c_void_p(sh.data_ptr()), ops._DT[dt], self.numel, ops._stream())
        self._shadow_state = (self._state_epoch(), [p._version for p in self.params])

    def _state_epoch(self):
        return (ops._FINE[0], getattr(self, "_epoch", 0))

    def owns(self, p):
        i = self._index.get(id(p))
        return i is not None and self.params[i] is p

    def shadow_view(self, p, dtype):
        """The `dtype` copy of parameter p as a view of the arena's shadow, or None when p was modified behind the arena's back
        (load_state_dict / copy_ since the last refresh) - the caller then casts it the slow way."""
        i = self._index.get(id(p))
        if i is None or self.params[i] is not p:
            return None
        st = self._shadow_state
        if dtype not in self._shadows or st is None or st[0] != self._state_epoch() or st[1][i] != p._version:
            if st is not None and st[0] == self._state_epoch() and st[1][i] != p._version:
                return None
            if dtype not in self._shadows:
                self._shadows[dtype] = torch.empty(self.numel, device=self.flat_p.device, dtype=dtype)
            self._refresh_shadows()
            if self._shadow_state[1][i] != p._version:
                return None
        o = self.offsets[i]
        return self._shadows[dtype][o:o + p.numel()].view(p.shape)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
