"""Whole-step training parity (e): one optimizer step of each trainer on a small generator
(dropout 0, fp32 storage, fixed seeds) against an ORACLE STEP = oracle forward (oracle/vocoder_oracle.py, CPU) + torch autograd
+ torch.optim.AdamW, in the reference's order:

  variant B  complete_vocoder.py:199-233   G forward once -> D step on the detached fake -> G step with the discriminators
                                           re-evaluated AFTER their update; LSGAN + 10 x output-L1 + 45 x mel term
  variant A  conditioned_hifigan.py:225-290  one AdamW over G + MPD + MSD; 45 L1 + 45 MSE(log-mel) + hinge per sub-discriminator

Compared: every entry of the loss dicts (rel 3e-4) and the parameter UPDATE of every tensor group.  AdamW's first update is
lr * g / (|g| + eps): it only keeps the sign of well-resolved gradients, so the update is compared as a vector (rel-L2) and the
few elements whose gradient is within fp32 noise of zero may differ by at most 2 * lr.
Also: data-parallel equivalence (two gloo ranks on this one GPU, half a batch each, buckets reduced under the backward, must
reproduce the one-rank full-batch step), the AdamW 'grad is None' rule and the torch.optim state_dict layout."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as O

pytestmark = pytest.mark.gpu

CFG = dict(hidden_channels=64, upsample_factors=[4, 2])
HOP = 8
LR = 2e-4


@pytest.fixture(scope="module")
def H():
    import hifigan_modified as H
    from hifigan_modified import _native
    _native.lib()
    assert torch.cuda.is_available(), "GPU tests need cuda:0"
    return H


def _inputs(B=2, T=128, seed=1):
    torch.manual_seed(seed)
    return (torch.randn(B, 80, T), (torch.randn(B, 1, T * HOP) * 0.5).clamp(-1, 1), torch.randn(B, 192), torch.randn(B, 384))


def _leaves(sd, prefixes):
    return {k: v.detach().cpu().clone().float().requires_grad_(True) for k, v in sd.items()
            if k.startswith(prefixes) and v.is_floating_point() and "embedding_extractor" not in k}


def _update_check(name, before, after_hip, after_ref, lr, tol):
    """rel-L2 of the parameter update over a group of tensors + max element difference."""
    num = den = 0.0
    worst = 0.0
    for k in after_ref:
        d_h = (after_hip[k].double() - before[k].double())
        d_r = (after_ref[k].double() - before[k].double())
        num += float((d_h - d_r).pow(2).sum())
        den += float(d_r.pow(2).sum())
        worst = max(worst, float((d_h - d_r).abs().max()))
    rel = (num / max(den, 1e-300)) ** 0.5
    assert den > 0, f"{name}: reference update is zero"
    assert rel < tol, f"{name}: update rel-L2 {rel:.3e} (max element diff {worst:.3e})"
    assert worst <= 2.05 * lr, f"{name}: an element moved by more than a sign flip: {worst:.3e}"
    return rel


@pytest.mark.parametrize("mel_mode", ["placeholder", "stft"])
def test_vocoder_trainer_one_step_matches_oracle_step(H, mel_mode):
    torch.manual_seed(0)
    voc = H.ModifiedHiFiGANVocoder(dropout=0.0, **CFG)
    sd0 = {k: v.detach().cpu().clone() for k, v in voc.state_dict().items()}
    mel, real, spk, emo = _inputs()

    # ---- oracle step (CPU): complete_vocoder.py:207-226
    P = _leaves(sd0, ("generator.", "discriminators."))
    gk = [k for k in P if k.startswith("generator.")]
    dk = [k for k in P if k.startswith("discriminators.")]
    optG = torch.optim.AdamW([P[k] for k in gk], lr=LR, betas=(0.8, 0.99), weight_decay=1e-4)
    optD = torch.optim.AdamW([P[k] for k in dk], lr=LR, betas=(0.8, 0.99), weight_decay=1e-4)
    fake = O.generator_forward(mel, P, "generator.", spk, emo, hidden_channels=64, upsample_factors=(4, 2))
    outs = O.discriminators_forward(real, fake.detach(), P, "discriminators.")
    dl = O.lsgan_discriminator_losses(outs)
    dl["total_loss"].backward()
    optD.step()
    optD.zero_grad()
    outs = O.discriminators_forward(real, fake, P, "discriminators.")          # re-evaluated AFTER the D update
    if mel_mode == "stft":
        target = O.mel_spectrogram(real, hop=HOP).detach()
        gl = O.lsgan_generator_losses(outs, target, O.mel_spectrogram(fake, hop=HOP))
    else:
        gl = O.lsgan_generator_losses(outs, mel, mel)                          # the reference's placeholder: term == 0
    gl["total_loss"].backward()
    optG.step()
    unused_ref = [k for k in gk if P[k].grad is None]

    # ---- the product step on the GPU
    tr = H.VocoderTrainer(voc, device=torch.device("cuda"), mel_mode=mel_mode)
    out = tr.train_step(mel.cuda(), real.cuda(), spk.cuda(), emo.cuda())
    assert set(out) == {"generator_loss", "discriminator_loss", "mel_loss"} and all(isinstance(v, float) for v in out.values())
    ref = {"generator_loss": float(gl["total_loss"]), "discriminator_loss": float(dl["total_loss"]), "mel_loss": float(gl["mel_loss"])}
    for k in ref:
        assert abs(out[k] - ref[k]) <= 3e-4 * max(1.0, abs(ref[k])), (k, out[k], ref[k])
    g_l, d_l = tr.last_losses
    for k in ("mpd_loss", "msd_loss", "mpd_fm_loss", "msd_fm_loss", "mel_loss"):
        assert abs(float(g_l[k]) - float(gl[k])) <= 3e-4 * max(1.0, abs(float(gl[k]))), k
    for k in ("mpd_real_loss", "mpd_fake_loss", "msd_real_loss", "msd_fake_loss"):
        assert abs(float(d_l[k]) - float(dl[k])) <= 3e-4 * max(1.0, abs(float(dl[k]))), k
    sd1 = {k: v.detach().float().cpu() for k, v in voc.state_dict().items()}
    after_ref = {k: P[k].detach() for k in P}
    # parameters the reference never gives a gradient (unused ODConv attention heads) stay bit-identical on both sides
    for k in unused_ref:
        assert torch.equal(sd1[k], sd0[k]) and torch.equal(after_ref[k], sd0[k]), k
    used_g = [k for k in gk if k not in unused_ref]
    rd = _update_check("discriminators", sd0, sd1, {k: after_ref[k] for k in dk}, LR, 0.05)
    rg = _update_check("generator", sd0, sd1, {k: after_ref[k] for k in used_g}, LR, 0.08)
    print(f"[{mel_mode}] update rel-L2: D {rd:.2e}  G {rg:.2e}")


def test_hifigan_trainer_losses_and_step_match_oracle(H):
    """Variant A: hinge per sub-discriminator + 45 L1 + 45 MSE(log-mel), ONE AdamW over everything (torch defaults)."""
    torch.manual_seed(0)
    model = H.ConditionedHiFiGAN(dropout=0.0, device="cuda", **CFG)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    mel, real, spk, emo = _inputs()
    mel = mel * 0.5 - 3.0                       # a log-mel-like target, so the MSE term is O(1)
    P = _leaves(sd0, ("generator.",))
    opt = torch.optim.AdamW(list(P.values()), lr=LR)
    fake = O.generator_forward(mel, P, "generator.generator.", spk, emo, hidden_channels=64, upsample_factors=(4, 2))
    outs = O.discriminators_forward(real, fake, P, "generator.")
    ref = {"feature_loss": (fake - real).abs().mean(), "mel_loss": ((O.mel_spectrogram(fake, hop=HOP) - mel) ** 2).mean(),
           "mpd_loss": O.hinge_generator_loss(outs["mpd_fake"]), "msd_loss": O.hinge_generator_loss(outs["msd_fake"])}
    total_ref = 45.0 * ref["feature_loss"] + 45.0 * ref["mel_loss"] + ref["mpd_loss"] + ref["msd_loss"]
    total_ref.backward()
    opt.step()

    model = model.to("cuda")
    tr = H.HiFiGANTrainer(model, learning_rate=LR, device="cuda")
    total, parts = tr.train_step(mel.cuda(), real.cuda(), spk.cuda(), emo.cuda())
    assert isinstance(total, float) and set(parts) == set(ref)
    assert abs(total - float(total_ref)) <= 3e-4 * abs(float(total_ref)), (total, float(total_ref))
    for k in ref:
        assert abs(float(parts[k]) - float(ref[k])) <= 3e-4 * max(1.0, abs(float(ref[k]))), k
    sd1 = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    moved = {k: P[k].detach() for k in P if P[k].grad is not None}
    for k in P:
        if P[k].grad is None:
            assert torch.equal(sd1[k], sd0[k]), k
    assert any(k.startswith("generator.mpd.") for k in moved) and any(k.startswith("generator.msd.") for k in moved)
    r = _update_check("variant A (G + MPD + MSD)", sd0, sd1, moved, LR, 0.08)
    print(f"[variant A] update rel-L2 {r:.2e}")


def test_flat_adamw_skips_gradless_parameters_and_reads_torch_state(H):
    """torch.optim.AdamW leaves a parameter with .grad None untouched (no decay of weights or moments); FlatAdamW must too.
    And a torch-layout optimizer state_dict (the reference checkpoints' 'optimizer_state_dict') loads into the arena."""
    from hifigan_modified.optim import FlatAdamW
    torch.manual_seed(0)
    mk = lambda: [torch.nn.Parameter(torch.randn(7, 5, device="cuda")), torch.nn.Parameter(torch.randn(33, device="cuda")),
                  torch.nn.Parameter(torch.randn(4, 3, 2, device="cuda")), torch.nn.Parameter(torch.randn(9, device="cuda"))]
    ps = mk()
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FlatAdamW(ps, lr=1e-2, betas=(0.8, 0.99), weight_decay=1e-1)
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.8, 0.99), weight_decay=1e-1)
    for it in range(3):
        for i, (p, q) in enumerate(zip(ps, ref)):
            if i == 1 or (i == 3 and it == 1):       # parameter 1 never gets a gradient, parameter 3 misses one step
                p.grad = q.grad = None
                continue
            gr = torch.randn_like(q)
            p.grad, q.grad = gr.clone(), gr.clone()
        opt.step()
        topt.step()
    assert torch.equal(ps[1].detach(), ref[1].detach())           # untouched, bit for bit
    for i in (0, 2):
        assert O.rel_l2(ps[i].detach().cpu(), ref[i].detach().cpu()) < 1e-6
    # parameter 3 skipped a step: like torch, the arena keeps a per-parameter step count for the bias correction
    assert O.rel_l2(ps[3].detach().cpu(), ref[3].detach().cpu()) < 1e-6
    assert opt.steps == [3, 0, 3, 2] and opt.step_count == 3
    tsd = opt.torch_state_dict()
    assert sorted(tsd["state"]) == [0, 2, 3] and float(tsd["state"][3]["step"]) == 2.0 and tsd["param_groups"][0]["params"] == [0, 1, 2, 3]
    # torch layout in, torch layout out
    ps2 = mk()
    opt2 = FlatAdamW(ps2, lr=1e-2, betas=(0.8, 0.99), weight_decay=1e-1)
    opt2.load_state_dict(topt.state_dict())
    st = topt.state_dict()["state"]
    assert opt2.step_count == 3
    for i, (p, o) in enumerate(zip(opt2.params, opt2.offsets)):
        if i in st:
            assert torch.equal(opt2.exp_avg[o:o + p.numel()].view(p.shape), st[i]["exp_avg"])
            assert torch.equal(opt2.exp_avg_sq[o:o + p.numel()].view(p.shape), st[i]["exp_avg_sq"])
        else:
            assert float(opt2.exp_avg[o:o + p.numel()].abs().max()) == 0.0
    t2 = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in ps2], lr=1e-2)
    t2.load_state_dict(opt.torch_state_dict())                    # the reverse direction loads with torch's own loader
    assert torch.equal(t2.state_dict()["state"][0]["exp_avg"], opt.exp_avg[:35].view(7, 5))


# ------------------------------------------------------------------------------------------------ data parallel
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from conftest import PKG, ROOT  # noqa: F401  (puts the package on sys.path)
    import hifigan_modified as H
    from hifigan_modified.parallel import broadcast_parameters, init_distributed
    init_distributed("gloo")                    # two ranks share cuda:0; the collective goes through gloo
    dev = torch.device("cuda", 0)
    torch.manual_seed(rank)                     # different initial weights per rank: the broadcast must fix that
    voc = H.ModifiedHiFiGANVocoder(dropout=0.0, **CFG).to(dev)
    broadcast_parameters(voc)
    tr = H.VocoderTrainer(voc, device=dev, mel_mode="stft", bucket_mib=1)      # grad_sync defaults to "overlap" at world > 1
    assert tr.grad_sync == "overlap"
    mel, real, spk, emo = _inputs(B=4)
    sl = slice(rank * 2, rank * 2 + 2)
    out = tr.train_step(mel[sl].to(dev), real[sl].to(dev), spk[sl].to(dev), emo[sl].to(dev))
    ovg, ovd = tr.overlap_sync(tr.generator_optimizer), tr.overlap_sync(tr.discriminator_optimizer)
    # numpy arrays travel through the queue by value (torch tensors would be passed as shared-memory handles that die with this process)
    sd = {k: v.detach().float().cpu().numpy() for k, v in voc.state_dict().items() if k.startswith(("generator.", "discriminators."))}
    q.put((rank, out, sd, len(ovg.buckets), len(ovd.buckets), ovg.reduced_bytes + ovd.reduced_bytes))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_equal_one_rank_full_batch(H):
    """configs[3] in miniature: the global batch of 4 clips split over 2 ranks (all-reduce(mean) of the gradient buckets, launched
    from grad-ready hooks while the backward is still running) must give the weights of one rank stepping on all 4 clips."""
    import torch.multiprocessing as mp
    # one-rank reference: rank 0's initial weights, full batch
    torch.manual_seed(0)
    voc = H.ModifiedHiFiGANVocoder(dropout=0.0, **CFG)
    sd0 = {k: v.detach().cpu().clone() for k, v in voc.state_dict().items()}
    tr = H.VocoderTrainer(voc, device=torch.device("cuda"), mel_mode="stft")
    assert tr.grad_sync is False
    mel, real, spk, emo = _inputs(B=4)
    out1 = tr.train_step(mel.cuda(), real.cuda(), spk.cuda(), emo.cuda())
    sd1 = {k: v.detach().float().cpu() for k, v in voc.state_dict().items()}

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, o0, s0, nbg, nbd, nbytes), (_, o1, s1, _, _, _) = res
    s0 = {k: torch.from_numpy(v) for k, v in s0.items()}
    s1 = {k: torch.from_numpy(v) for k, v in s1.items()}
    assert nbg >= 2 and nbd >= 2 and nbytes > 0                   # several buckets per optimizer really went through the collective
    for k in s0:                                                  # both ranks hold the same weights after the step, bit for bit
        assert torch.equal(s0[k], s1[k]), k
    # per-rank losses are half-batch means; their average is the full-batch loss
    for k in out1:
        assert abs(0.5 * (o0[k] + o1[k]) - out1[k]) <= 5e-4 * max(1.0, abs(out1[k])), k
    keys = [k for k in s0 if not torch.equal(sd1[k], sd0[k])]
    assert keys
    dk = [k for k in keys if k.startswith("discriminators.")]
    gk = [k for k in keys if k.startswith("generator.")]
    rd = _update_check("DP discriminators", sd0, s0, {k: sd1[k] for k in dk}, LR, 0.05)
    rg = _update_check("DP generator", sd0, s0, {k: sd1[k] for k in gk}, LR, 0.08)
    print(f"[dp2 vs dp1] update rel-L2: D {rd:.2e}  G {rg:.2e}")
