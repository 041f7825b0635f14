"""Error budget of the 16-bit generator pipeline, computed on the CPU from the oracle (no GPU needed).

The HIP pipeline rounds in four kinds of places: (W) packed weights, (A) the activation operand of every MFMA, which
is also what a stage stores to HBM, (C) the concat tensor `c` inside an MRF block before the fusion conv, and (O) the
emitted waveform.  This script re-runs the oracle's arithmetic in fp32 with those roundings injected one group at a
time, so the waveform rel-L2 of every single source - and of any mix - can be read off before a kernel is written.

    python tools/error_budget.py            # 22 kHz default generator, B=2 x 32 frames
    python tools/error_budget.py 48k

Modes: fp32 (no rounding), bf16, fp16, fp16x2 (hi + lo fp16 pair, 22 significant bits), bf16x2, bf16x3.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd"))
import torch
import torch.nn.functional as F
from oracle import vocoder_oracle as O


def q(t, mode):
    if mode == "fp32":
        return t
    if mode in ("bf16", "fp16"):
        dt = torch.bfloat16 if mode == "bf16" else torch.float16
        return t.to(dt).float()
    base, n = mode[:-2], int(mode[-1])
    dt = torch.bfloat16 if base == "bf16" else torch.float16
    acc = torch.zeros_like(t)
    for _ in range(n):
        acc = acc + (t - acc).to(dt).float()
    return acc


STAGES = ["mel", "input_proj", "up0", "up1", "up2", "up3", "mrf0", "mrf1", "mrf2", "wave"]
LAYERS = ["input_proj", "up0", "up1", "up2", "up3", "mrf0", "mrf1", "mrf2", "out"]


def forward(mel, sd, spk, emo, ups_f, act, wgt, cmode, kloop_layers=("input_proj", "up0"), opmode=None, fmode=None):
    """act[stage] = storage/operand mode of that stage's OUTPUT; wgt[layer] = weight mode (an MRF layer may give a dict
    {conv, res, fus}); cmode[i] = mode of MRF i's concat; fmode = storage mode of the chain's pre-GroupNorm fusion output f."""
    ups, mrfs, cur = O.generator_channel_plan(512, ups_f, 4, ((1, 3, 5),) * 3)

    def odconv(x, prefix, name, transposed, **kw):
        W, bias = sd[prefix + "kernels"], sd[prefix + "bias"]
        alpha = O.odconv_attention(x, sd[prefix + "kernel_attention.1.weight"], sd[prefix + "kernel_attention.1.bias"])
        if name in kloop_layers:      # banks used as stored, alpha applied to the fp32 accumulators
            Wq = q(W, wgt[name])
            Wb = torch.einsum("bk,k...->b...", alpha, Wq)
        else:                         # aggregate form: mixed in fp32 from the rounded banks, rounded again as an operand
            Wb = q(torch.einsum("bk,k...->b...", alpha, q(W, wgt[name])), wgt[name])
        bb = alpha @ bias
        fn = F.conv_transpose1d if transposed else F.conv1d
        return torch.cat([fn(x[i:i + 1], Wb[i], bb[i], **kw) for i in range(x.shape[0])], 0)

    x = q(mel, act["mel"])
    x = odconv(x, "input_proj.", "input_proj", False, padding=3)
    x = O.film(x, sd, "final_film.", spk, emo)
    x = q(x, act["input_proj"])
    for i, (_, _, f) in enumerate(ups):
        x = odconv(x, f"upsample_layers.{i}.0.", f"up{i}", True, stride=f, padding=f // 2, output_padding=f % 2)
        x = q(O.leaky_relu(x, 0.1), act[f"up{i}"])
    for i, (_, _, dil) in enumerate(mrfs):
        pre = f"mrf_blocks.{i}."
        wm = wgt[f"mrf{i}"]
        wmc, wmr, wmf = (wm["conv"], wm["res"], wm["fus"]) if isinstance(wm, dict) else (wm, wm, wm)
        xs = x                                   # as stored (residual path reads this)
        if opmode is not None:                   # MFMA operand view of the stored stream
            x = q(xs, opmode)
        branches = []
        for j, d in enumerate(dil):
            p = f"{pre}conv_layers.{j}."
            Wc = sd[p + "conv.weight"]
            out_ch, cin_g, k = Wc.shape
            G = x.shape[1] // cin_g
            # folded dense kernel in fp32 (mv_mrf_pack), rounded once
            Wd = torch.zeros(out_ch, x.shape[1], k)
            for o in range(out_ch):
                g = o // (out_ch // G)
                Wd[o, g * cin_g:(g + 1) * cin_g] = Wc[o]
            L = sd[p + "lora_A"] @ sd[p + "lora_B"]
            Wd[:, :, k // 2] += sd[p + "lora_scaling"] * L.t()
            Pw = sd[p + "output_projection.weight"][:, :, 0]
            Weff = torch.einsum("oq,qck->ock", Pw, Wd)
            beff = Pw @ sd[p + "conv.bias"] + sd[p + "output_projection.bias"]
            v = F.conv1d(x, q(Weff, wmc), beff, padding=d, dilation=d)
            w = O.group_norm(v, O.norm_groups_for(out_ch), sd[p + "norm.weight"], sd[p + "norm.bias"])
            a = O.silu(w)
            res = F.conv1d(x, q(sd[p + "residual_proj.weight"], wmr), sd[p + "residual_proj.bias"])
            branches.append(a + res)
        c = q(torch.cat(branches, 1), cmode[i])
        f_ = F.conv1d(c, q(sd[pre + "fusion.weight"], wmf), sd[pre + "fusion.bias"])
        if fmode is not None:
            f_ = q(f_, fmode)
        n = O.group_norm(f_, O.norm_groups_for(f_.shape[1]), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
        x = q(n + xs, act[f"mrf{i}"])
    Wo = sd["output_proj.weight"]
    y = F.conv1d(x, q(Wo, wgt["out"]), sd["output_proj.bias"], padding=Wo.shape[2] // 2)
    return q(torch.tanh(y), act["wave"]), y


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "22k"
    import hifigan_modified as H
    kw, nmel, T = ({}, 80, 32) if which == "22k" else (dict(mel_channels=128, upsample_factors=[8, 8, 4, 2]), 128, 16)
    torch.manual_seed(0)
    g = H.ModifiedHiFiGANGenerator(**kw)
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    ups_f = tuple(g.upsample_factors)
    B = int(os.environ.get("EB_BATCH", "2"))
    torch.manual_seed(1)
    mel, spk, emo = torch.randn(B, nmel, T), torch.randn(B, 192), torch.randn(B, 384)
    with torch.no_grad():
        ref = O.generator_forward(mel, sd, "", spk, emo, upsample_factors=ups_f, return_stages=True)
    print(f"{which}: pre-tanh rms {ref['output_proj'].pow(2).mean().sqrt():.2f}, |wave|>0.99: {(ref['wave'].abs() > 0.99).float().mean():.2f}")
    for k in ("up3", "mrf0", "mrf1", "mrf2"):
        print(f"  rms {k}: {ref[k].pow(2).mean().sqrt():.3f}", end="")
    print()

    def run(act=None, wgt=None, cm=None, base="fp32", **kwargs):
        a = {s: base for s in STAGES}
        w = {l: base for l in LAYERS}
        c = [base] * 3
        a.update(act or {})
        w.update(wgt or {})
        if cm:
            c = cm
        with torch.no_grad():
            wave, pre = forward(mel, sd, spk, emo, ups_f, a, w, c, **kwargs)
        return O.rel_l2(wave, ref["wave"]), O.rel_l2(pre, ref["output_proj"])

    e = run()
    print(f"all fp32 (restatement check): wave {e[0]:.2e} pre-tanh {e[1]:.2e}")
    for m in ("bf16", "fp16"):
        e = run(base=m)
        print(f"everything {m}: wave {e[0]:.2e} pre-tanh {e[1]:.2e}")
        e = run(base=m, act={"wave": "fp32"})
        print(f"everything {m}, fp32 waveform: wave {e[0]:.2e}")
        print(f"  single sources in {m} (all else fp32): wave rel-L2")
        for s in STAGES:
            print(f"    act {s:11s} {run(act={s: m})[0]:.2e}")
        for l in LAYERS:
            print(f"    wgt {l:11s} {run(wgt={l: m})[0]:.2e}")
        for i in range(3):
            c = ["fp32"] * 3
            c[i] = m
            print(f"    concat mrf{i}   {run(cm=c)[0]:.2e}")
    # candidate mixes
    print("mixes (wave fp32 out unless noted):")
    mixes = {
        "fp16 all + fp32 wave": dict(base="fp16", act={"wave": "fp32"}),
        "fp16 all, fp16x2 weights": dict(base="fp16", act={"wave": "fp32"}, wgt={l: "fp16x2" for l in LAYERS}),
        "fp16 all, fp16x2 weights mrf+out only": dict(base="fp16", act={"wave": "fp32"}, wgt={l: "fp16x2" for l in ("mrf0", "mrf1", "mrf2", "out")}),
        "fp16 acts, fp32 stream up3..mrf2": dict(base="fp16", act={"wave": "fp32", "up3": "fp32", "mrf0": "fp32", "mrf1": "fp32", "mrf2": "fp32"}),
        "fp16 acts, fp16x2 stream up3..mrf2": dict(base="fp16", act={"wave": "fp32", "up3": "fp16x2", "mrf0": "fp16x2", "mrf1": "fp16x2", "mrf2": "fp16x2"}),
        "fp16 acts, fp16x2 stream + fp16x2 wgt mrf/out": dict(base="fp16", act={"wave": "fp32", "up3": "fp16x2", "mrf0": "fp16x2", "mrf1": "fp16x2", "mrf2": "fp16x2"},
                                                              wgt={l: "fp16x2" for l in ("mrf0", "mrf1", "mrf2", "out")}),
        "fp16 acts, fp16x2 stream + fp16x2 wgt all": dict(base="fp16", act={"wave": "fp32", "up3": "fp16x2", "mrf0": "fp16x2", "mrf1": "fp16x2", "mrf2": "fp16x2"},
                                                          wgt={l: "fp16x2" for l in LAYERS}),
        "fp16x2 acts+wgts everywhere, fp16 concat": dict(base="fp16x2", act={"wave": "fp32"}, cm=["fp16"] * 3),
        "fp16x2 acts everywhere, fp16 wgts": dict(base="fp16x2", act={"wave": "fp32"}, wgt={l: "fp16" for l in LAYERS}, cm=["fp16"] * 3),
        "bf16x2 everywhere": dict(base="bf16x2", act={"wave": "fp32"}),
        "bf16x3 everywhere (current fp32 mode)": dict(base="bf16x3", act={"wave": "fp32"}),
    }
    for name, kwm in mixes.items():
        e = run(**kwm)
        print(f"  {name:50s} wave {e[0]:.2e} pre-tanh {e[1]:.2e}")


if __name__ == "__main__":
    main()
