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
        params = list(params)
        keep = [i for i, p in enumerate(params) if p.requires_grad and id(p) not in excl]
        self.torch_index = keep                  # position of each arena parameter in the list a torch.optim optimizer would enumerate
        self.params = [params[i] for i in keep]
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
        self.step_count = 0                  # optimizer steps taken (the largest per-parameter count)
        self.steps = [0] * len(self.params)  # per parameter, like torch.optim.AdamW's state[p]['step']: a parameter whose .grad is None
        #                                      in a step is left untouched, so its bias correction lags the others'
        self._n_all = len(params)            # parameters a torch.optim optimizer would enumerate (param_groups[0]['params'])
        self._capture = None                 # state of a captured training step (capture_begin / capture_end)
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
        self._has_grad = None        # per parameter, set by the gathers of the current step (None = every parameter)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """param.grad tensors (fresh per backward) -> flat fp32 buffer; missing grads contribute zeros."""
        d = self._descs
        self._has_grad = [p.grad is not None for p in self.params]
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                d["src"][i], d["dtype"][i] = 0, 0
            else:
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g
                d["src"][i], d["dtype"][i] = g.data_ptr(), ops._DT[g.dtype]
        table = self._upload(d)
        N.call("mv_multi_gather", c_void_p(table.data_ptr()), len(self.params), self._max_len,
               c_void_p(self.flat_g.data_ptr()), ops._stream())
        self._table = table   # keep alive until the launch has consumed it
        return self.flat_g

    def _upload(self, descs):
        """Descriptor table -> device.  Under stream capture the copy becomes a node of the HIP graph: its source is a pinned
        host buffer that lives (unchanged) as long as the graph does - the gradient addresses it holds are the capture pool's, which
        every replay reuses."""
        raw = torch.from_numpy(descs.view(np.uint8).reshape(-1).copy())
        cap = self._capture
        if cap is not None:
            if not cap["pinned"]:
                raise RuntimeError("FlatAdamW: more gradient gathers in the captured step than capture_begin() prepared tables for")
            pin = cap["pinned"].pop()           # pinned BEFORE the capture began (host allocations invalidate a capture)
            pin[:raw.numel()].copy_(raw)
            cap["keep"].append(pin)
            return pin[:raw.numel()].to(self.flat_g.device, non_blocking=True)
        # eager path: a ring of pinned staging buffers, so that the copy is a real asynchronous H2D (from pageable memory it is staged
        # synchronously - a host stall per bucket while the backward is producing the next gradients).  A slot is reused only after the
        # copy that last read it has completed (event per slot; in steady state it has, long ago).
        ring = getattr(self, "_pin_ring", None)
        if ring is None:
            ring = self._pin_ring = {"slots": [], "next": 0}
        n = raw.numel()
        if len(ring["slots"]) < 16:
            ring["slots"].append([torch.empty(max(n, 4096), dtype=torch.uint8).pin_memory(), None])
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["next"] % 16]
            ring["next"] += 1
            if slot[1] is not None:
                slot[1].synchronize()
            if slot[0].numel() < n:
                slot[0] = torch.empty(n, dtype=torch.uint8).pin_memory()
        slot[0][:n].copy_(raw)
        dev = slot[0][:n].to(self.flat_g.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.flat_g.device))
        slot[1] = ev
        return dev

    def gather_range(self, i0, i1):
        """Gather the gradients of parameters i0..i1-1 only (one launch): the per-bucket form used by
        parallel.OverlappedGradSync while the backward is still running.  Missing gradients contribute zeros."""
        d = self._descs[i0:i1].copy()
        if self._has_grad is None or len(self._has_grad) != len(self.params):
            self._has_grad = [True] * len(self.params)
        for j, p in enumerate(self.params[i0:i1]):
            g = p.grad
            self._has_grad[i0 + j] = g is not None
            if g is None:
                d["src"][j], d["dtype"][j] = 0, 0
            else:
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g
                d["src"][j], d["dtype"][j] = g.data_ptr(), ops._DT[g.dtype]
        table = self._upload(d)
        N.call("mv_multi_gather", c_void_p(table.data_ptr()), i1 - i0, int(d["n"].max()),
               c_void_p(self.flat_g.data_ptr()), ops._stream())
        self._tables = getattr(self, "_tables", [])[-64:] + [table]   # keep alive until the launches have consumed them

    def step(self, grad_scale=1.0, gathered=False):
        if not gathered:
            self.gather_grads()
        # torch.optim.AdamW leaves a parameter whose .grad is None untouched (no weight decay, no moment decay, no step): update only
        # the runs of parameters that received a gradient this step and share a step count - one launch when all did, the usual case
        hg = self._has_grad if self._has_grad is not None else [True] * len(self.params)
        runs, i, n = [], 0, len(self.params)
        while i < n:
            if hg[i]:
                j = i
                while j + 1 < n and hg[j + 1] and self.steps[j + 1] == self.steps[i]:
                    j += 1
                end = self.numel if j == n - 1 else self.offsets[j + 1]
                runs.append((i, j + 1, self.offsets[i], end - self.offsets[i]))
                i = j + 1
            else:
                i += 1
        from .torch_ops import OPS
        cap = self._capture
        for (i0, i1, off, cnt) in runs:   # torch.ops.mi355x_vocoder.fused_adamw_ (in place on the arena slices; mv_adamw_flat)
            t = self.steps[i0] + 1
            args = (self.flat_p[off:off + cnt], self.flat_g[off:off + cnt], self.exp_avg[off:off + cnt], self.exp_avg_sq[off:off + cnt],
                    float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay))
            if cap is None:
                OPS.fused_adamw_(*args, int(t), float(grad_scale))
                for k in range(i0, i1):
                    self.steps[k] = t
            else:
                # being captured (nothing executes now): the count lives on the device and the graph itself advances it; the host's
                # counts move in after_replay()
                if len(cap["runs"]) >= len(cap["steps"]):
                    raise RuntimeError("FlatAdamW: the captured step does not update the arena as one run per step count")
                st = cap["steps"][len(cap["runs"])]
                st.add_(1)
                OPS.fused_adamw_dev_(*args, st, float(grad_scale))
                cap["runs"].append((i0, i1))
        self.step_count = max(self.steps)
        self._has_grad = None
        ops.bump_param_epoch(self)       # in-place arena update: cached casts / packed weights of THESE parameters must refresh
        self._refresh_shadows()

    # ---- captured training steps (VocoderTrainer / HiFiGANTrainer with use_graph): everything the step launches is recorded once
    def capture_begin(self):
        """Call right before a training step is captured into a HIP graph (torch.cuda.graph): the gathers' descriptor tables become
        pinned and persistent, and AdamW reads its step count from device tensors that the graph itself advances."""
        dev = self.flat_p.device
        planned = self._planned_runs()
        # one device counter per run, set OUTSIDE the capture to the count the first replay starts from
        steps = [torch.full((1,), self.steps[i0], dtype=torch.int32, device=dev) for (i0, _) in planned]
        nbytes = self._descs.nbytes
        pinned = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(4)]
        self._capture = {"keep": [], "steps": steps, "runs": [], "planned": planned, "pinned": pinned}

    def _planned_runs(self):
        """Runs of equal step count over ALL parameters (a captured step must give every parameter of the arena a gradient)."""
        runs, i, n = [], 0, len(self.params)
        while i < n:
            j = i
            while j + 1 < n and self.steps[j + 1] == self.steps[i]:
                j += 1
            runs.append((i, j + 1))
            i = j + 1
        return runs

    def capture_end(self):
        cap, self._capture = self._capture, None
        if cap is None:
            return None
        if cap["runs"] != cap["planned"]:
            raise RuntimeError("FlatAdamW: the captured step left parameters without a gradient: a captured step must update every "
                               "parameter of the arena")
        return {"keep": cap["keep"], "steps": cap["steps"], "runs": cap["runs"]}

    def after_replay(self, state):
        """Host-side bookkeeping after one replay of a captured step: step counts and cache epochs."""
        for (i0, i1) in state["runs"]:
            for k in range(i0, i1):
                self.steps[k] += 1
        self.step_count = max(self.steps)
        ops.bump_param_epoch(self)
        self._shadow_state = (self._state_epoch(), [p._version for p in self.params])   # the replay refreshed the shadows itself

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

    def torch_state_dict(self):
        """The same state in torch.optim.AdamW's layout (what the reference's load path expects)."""
        state = {}
        for k, (p, o, ti) in enumerate(zip(self.params, self.offsets, self.torch_index)):
            if self.steps[k] == 0:
                continue                      # torch creates a parameter's state at its first update
            state[ti] = {"step": torch.tensor(float(self.steps[k])), "exp_avg": self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                         "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone()}
        n_all = self._n_all
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(n_all))}
        return {"state": state, "param_groups": [group]}

    def state_dict(self):
        return {"step": self.step_count, "steps": list(self.steps), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        """Accepts this class's flat layout, or a torch.optim.AdamW state_dict ({'state': {i: {step, exp_avg, exp_avg_sq}},
        'param_groups': [...]}, the 'optimizer_state_dict' of a reference checkpoint: conditioned_hifigan.py:292-299) built over
        the same parameter list: entry i belongs to the i-th parameter the torch optimizer enumerated (self.torch_index)."""
        if "state" in sd and "param_groups" in sd:
            g0 = sd["param_groups"][0]
            self.lr, self.betas, self.eps = g0.get("lr", self.lr), tuple(g0.get("betas", self.betas)), g0.get("eps", self.eps)
            self.weight_decay = g0.get("weight_decay", self.weight_decay)
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            self.steps = [0] * len(self.params)
            with torch.no_grad():
                for k, (p, o, ti) in enumerate(zip(self.params, self.offsets, self.torch_index)):
                    st = sd["state"].get(ti, sd["state"].get(str(ti)))
                    if st is None:
                        continue
                    if st["exp_avg"].numel() != p.numel():
                        raise ValueError(f"optimizer state {ti}: {tuple(st['exp_avg'].shape)} does not fit parameter {tuple(p.shape)}")
                    self.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    self.steps[k] = int(st["step"])
            self.step_count = max(self.steps) if self.steps else 0
            return
        self.step_count = int(sd["step"])
        self.steps = [int(x) for x in sd["steps"]] if "steps" in sd else [self.step_count] * len(self.params)
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
