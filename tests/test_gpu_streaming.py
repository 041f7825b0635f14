"""The callers and data formats either side of the path: chunked vocoding, the wave->mel front end
and clip sampler, and the checkpoint formats - all through the HIP kernels (C ABI), on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import vocoder_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import hifigan_modified as H
    from hifigan_modified import _native
    _native.lib()
    return H


def small_gen(H, dtype=torch.float32):
    torch.manual_seed(0)
    return H.ModifiedHiFiGANGenerator(hidden_channels=64, upsample_factors=[4, 2]).cuda().to(dtype).train(False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T", [96, 77])
def test_chunked_independent_equals_chunk_by_chunk_forwards(H, dtype, T):
    """policy='independent' = the reference's streaming_forward (streamspeech_integration.py:377-414): one ordinary forward
    per 32-frame chunk.  Stacking the chunks on the batch axis changes nothing but summation order (every op is
    per-sample; the pooled sums use fp32 atomics and the ODConv kernel picks its tiling from the batch size): fp32 storage
    agrees to 1e-5 rel-L2, bf16 to the rounding-flip level (1e-2).  The ragged tail is vocoded as its own shorter utterance."""
    gen = small_gen(H, dtype)
    torch.manual_seed(1)
    mel = torch.randn(2, 80, T, device="cuda").to(dtype)
    spk, emo = torch.randn(2, 192, device="cuda").to(dtype), torch.randn(2, 384, device="cuda").to(dtype)
    cv = H.ChunkedVocoder(gen, chunk_frames=32)
    got = cv(mel, spk, emo)
    assert got.shape == (2, 1, T * 8) and got.dtype == dtype
    with torch.no_grad():
        pieces = [gen(mel[:, :, t:t + 32].contiguous(), spk, emo) for t in range(0, T, 32)]
    want = torch.cat(pieces, dim=2)
    assert O.rel_l2(got.float().cpu(), want.float().cpu()) < (1e-5 if dtype == torch.float32 else 1e-2)


def test_chunked_context_policy_is_closer_to_the_full_forward(H):
    """policy='context' (extension): windows with 8 real context frames per side.  GroupNorm / ODConv pooling still see
    the window, so it never equals the full-utterance forward; it must be closer to it than independent chunks are at the
    chunk boundaries (zero-padded convolutions), and both deviations are finite and recorded."""
    gen = small_gen(H)
    torch.manual_seed(1)
    mel = torch.randn(1, 80, 128, device="cuda")
    spk, emo = torch.randn(1, 192, device="cuda"), torch.randn(1, 384, device="cuda")
    with torch.no_grad():
        full = gen(mel, spk, emo)
    ind = H.ChunkedVocoder(gen, 32)(mel, spk, emo)
    ctx = H.ChunkedVocoder(gen, 32, policy="context", context_frames=8)(mel, spk, emo)
    assert ind.shape == ctx.shape == full.shape
    hop = 8
    # boundary region: 2 frames either side of each interior chunk boundary
    idx = torch.cat([torch.arange((b - 2) * hop, (b + 2) * hop) for b in (32, 64, 96)]).cuda()
    e_ind = float((ind - full)[..., idx].norm() / full[..., idx].norm())
    e_ctx = float((ctx - full)[..., idx].norm() / full[..., idx].norm())
    assert np.isfinite([e_ind, e_ctx]).all()
    assert e_ctx < e_ind, (e_ctx, e_ind)


def test_chunked_vocoder_argument_checks(H):
    gen = small_gen(H)
    with pytest.raises(ValueError):
        H.ChunkedVocoder(gen, 32, policy="bogus")
    with pytest.raises(ValueError):
        H.ChunkedVocoder(gen, 0)
    with pytest.raises(ValueError):
        H.ChunkedVocoder(gen, 32)(torch.randn(80, 32, device="cuda"))


def test_mel_front_end_matches_oracle_and_loss_targets(H):
    """MelFrontEnd = the mel/STFT-loss kernel in `want_mel` mode: equals the oracle's explicit-DFT log-mel (which the CPU
    suite pins to torch.stft) to fp32 rounding."""
    torch.manual_seed(3)
    wave = torch.randn(2, 1, 4096).clamp(-1, 1)
    fe = H.MelFrontEnd()
    mel = fe(wave.cuda())
    assert mel.shape == (2, 80, 16) and mel.dtype == torch.float32
    ref = O.mel_spectrogram(wave, 22050, 1024, 256, 80, 0.0, 8000.0)
    assert O.rel_l2(mel.cpu(), ref) < 2e-5


def test_clip_sampler_is_deterministic_per_rank_and_pads_short_utterances(H):
    torch.manual_seed(4)
    utts = [torch.randn(30000), torch.randn(5000), torch.randn(8192)]
    fe = H.MelFrontEnd()
    a = H.ClipSampler(utts, 8192, fe, seed=7, rank=0)
    b = H.ClipSampler(utts, 8192, fe, seed=7, rank=0)
    c = H.ClipSampler(utts, 8192, fe, seed=7, rank=1)
    pa, pb, pc = a.draw(16), b.draw(16), c.draw(16)
    assert pa == pb and pa != pc
    a2 = H.ClipSampler(utts, 8192, fe, seed=7, rank=0)
    wave, mel = a2.sample(16)
    assert wave.shape == (16, 1, 8192) and mel.shape == (16, 80, 32)
    for r, (i, s) in enumerate(pa):
        seg = utts[i][s:s + 8192]
        assert torch.equal(wave[r, 0, :seg.numel()].cpu(), seg)
        assert float(wave[r, 0, seg.numel():].abs().sum()) == 0.0     # short utterance: zero right-padding
    assert O.rel_l2(mel.cpu(), fe(wave).cpu()) == 0.0


def test_conditioned_hifigan_checkpoint_round_trip(H, tmp_path):
    """save_model / load_model (conditioned_hifigan.py:196-208): same dict keys as the reference's checkpoints, loaded
    with weights_only=True, identical weights and the same waveform afterwards; get_model_info reports the reference's fields."""
    torch.manual_seed(0)
    m = H.ConditionedHiFiGAN(hidden_channels=64, upsample_factors=[4, 2], device="cuda").to("cuda")
    info = m.get_model_info()
    assert {"total_parameters", "trainable_parameters", "architecture", "conditioning", "config"} <= set(info)
    assert info["total_parameters"] == sum(p.numel() for p in m.parameters())
    path = os.path.join(tmp_path, "cond.pt")
    m.save_model(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"model_state_dict", "config", "model_info"}
    assert any(k.startswith("generator.generator.mrf_blocks.0.conv_layers.0.residual_proj.") for k in ck["model_state_dict"])
    assert any(k.startswith("generator.mpd.discriminators.4.conv_layers.8.") for k in ck["model_state_dict"])
    torch.manual_seed(123)
    m2 = H.ConditionedHiFiGAN(hidden_channels=64, upsample_factors=[4, 2], device="cuda").to("cuda")
    cfg, inf = m2.load_model(path)
    assert inf["total_parameters"] == info["total_parameters"]
    torch.manual_seed(1)
    mel, spk, emo = torch.randn(2, 80, 16, device="cuda"), torch.randn(2, 192, device="cuda"), torch.randn(2, 384, device="cuda")
    m.train(False); m2.train(False)
    for (k, a), (_, b2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b2), k                    # the restored weights are bit-identical
    with torch.no_grad():                               # ... and so is the waveform, up to the fp32 atomic order of the pooled sums
        assert O.rel_l2(m(mel, speaker_emb=spk, emotion_emb=emo), m2(mel, speaker_emb=spk, emotion_emb=emo)) < 1e-4


def test_vocoder_trainer_checkpoint_resumes_identically(H, tmp_path):
    """complete_vocoder.py:235-248: {'vocoder_state_dict','generator_optimizer_state_dict','discriminator_optimizer_state_dict'}.
    A trainer restored from the checkpoint continues exactly like the one that wrote it (weights + AdamW moments + step)."""
    def make():
        torch.manual_seed(0)
        voc = H.ModifiedHiFiGANVocoder(hidden_channels=64, upsample_factors=[4, 2], dropout=0.0)
        return H.VocoderTrainer(voc, device=torch.device("cuda"))
    torch.manual_seed(1)
    mel = torch.randn(2, 80, 128, device="cuda")
    real = torch.randn(2, 1, 1024, device="cuda").clamp(-1, 1)
    spk, emo = torch.randn(2, 192, device="cuda"), torch.randn(2, 384, device="cuda")
    a = make()
    a.train_step(mel, real, spk, emo)
    path = os.path.join(tmp_path, "trainer.pt")
    a.save_checkpoint(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"vocoder_state_dict", "generator_optimizer_state_dict", "discriminator_optimizer_state_dict"}
    # the optimizer entries are in torch.optim.AdamW's own layout: the reference's load path (complete_vocoder.py:463-468) takes them
    for key, mod in (("generator_optimizer_state_dict", a.vocoder.generator), ("discriminator_optimizer_state_dict", a.vocoder.discriminators)):
        topt = torch.optim.AdamW([torch.nn.Parameter(p.detach().cpu().clone()) for p in mod.parameters()], lr=2e-4, betas=(0.8, 0.99), weight_decay=1e-4)
        topt.load_state_dict(ck[key])
        assert len(topt.state) > 0 and all(float(st["step"]) == 1.0 for st in topt.state.values())
    b = make()
    b.load_checkpoint(path)
    # the restored state is bit-identical: weights, AdamW moments, step counters
    for (n, p), (_, q) in zip(a.vocoder.named_parameters(), b.vocoder.named_parameters()):
        assert torch.equal(p, q), n
    for oa, ob in ((a.generator_optimizer, b.generator_optimizer), (a.discriminator_optimizer, b.discriminator_optimizer)):
        sa, sb = oa.state_dict(), ob.state_dict()
        assert sa["step"] == sb["step"] == 1
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    # and the next step continues the same trajectory (up to the fp32 atomic summation order of the gradient kernels)
    la = a.to_floats(a.train_step(mel, real, spk, emo))
    lb = b.to_floats(b.train_step(mel, real, spk, emo))
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(la[k])), k
    for (n, p), (_, q) in zip(a.vocoder.named_parameters(), b.vocoder.named_parameters()):
        assert float((p - q).detach().abs().max()) <= 1e-3 * max(1e-3, float(p.detach().abs().max())), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hifigan_trainer_variant_a_steps_and_checkpoints(H, tmp_path, dtype):
    """conditioned_hifigan.py:210-299 (variant A: ONE AdamW over G + MPD + MSD; 45 L1 + 45 MSE(log-mel) + hinge per
    sub-discriminator).  Two steps on a small model: finite loss breakdown with the reference's keys, parameters of the
    generator AND of both discriminator families move, and save_checkpoint writes the reference's dict keys."""
    torch.manual_seed(0)
    model = H.ConditionedHiFiGAN(hidden_channels=64, upsample_factors=[4, 2], device="cuda").to("cuda")
    tr = H.HiFiGANTrainer(model, learning_rate=2e-4, device="cuda")
    torch.manual_seed(1)
    mel = torch.randn(2, 80, 128, device="cuda").to(dtype)
    real = torch.randn(2, 1, 1024, device="cuda").clamp(-1, 1).to(dtype)
    spk, emo = torch.randn(2, 192, device="cuda").to(dtype), torch.randn(2, 384, device="cuda").to(dtype)
    g0 = model.generator.generator.output_proj.weight.detach().clone()
    p0 = model.generator.mpd.discriminators[0].conv_layers[0].weight.detach().clone()
    s0 = model.generator.msd.discriminators[2].conv_layers[8].weight.detach().clone()
    for _ in range(2):
        total, parts = tr.train_step(mel, real, spk, emo)
        assert np.isfinite(total)
        assert all(np.isfinite(float(v.detach() if torch.is_tensor(v) else v)) for v in parts.values()) and len(parts) >= 3
    assert float((model.generator.generator.output_proj.weight.detach() - g0).abs().max()) > 0
    assert float((model.generator.mpd.discriminators[0].conv_layers[0].weight.detach() - p0).abs().max()) > 0
    assert float((model.generator.msd.discriminators[2].conv_layers[8].weight.detach() - s0).abs().max()) > 0
    path = os.path.join(tmp_path, "variant_a.pt")
    tr.save_checkpoint(path, epoch=3, loss=total)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"} and ck["epoch"] == 3
    topt = torch.optim.AdamW([torch.nn.Parameter(p.detach().float().cpu().clone()) for p in model.parameters()], lr=2e-4)
    topt.load_state_dict(ck["optimizer_state_dict"])              # torch.optim.AdamW's layout (conditioned_hifigan.py:292-299)
    assert len(topt.state) > 0 and all(float(st["step"]) == 2.0 for st in topt.state.values())


def test_pack_cache_follows_each_optimizer_separately():
    """Flat-arena AdamW updates weights behind torch's back: the discriminator's packed MFMA weights must be rebuilt after ITS step,
    survive the generator optimizer's step untouched, and always equal a from-scratch pack."""
    import hifigan_modified as H
    from hifigan_modified import disc_fused
    torch.manual_seed(0)
    voc = H.ModifiedHiFiGANVocoder(hidden_channels=64)
    tr = H.VocoderTrainer(voc, device=torch.device("cuda"))
    torch.manual_seed(1)
    mel = torch.randn(2, 80, 8, device="cuda").bfloat16()
    real = torch.randn(2, 1, 2048, device="cuda").clamp(-1, 1).bfloat16()
    spk, emo = torch.randn(2, 192, device="cuda").bfloat16(), torch.randn(2, 256, device="cuda").bfloat16()
    probe = torch.randn(2, 1, 2048, device="cuda").clamp(-1, 1).bfloat16()

    def d_out():
        with torch.no_grad():
            return torch.cat([o.float().flatten() for o in voc.discriminators.mpd(probe) + voc.discriminators.msd(probe)])

    def fresh():
        saved = disc_fused._packs.d
        disc_fused._packs.d = {}
        try:
            return d_out()
        finally:
            disc_fused._packs.d = saved

    y0 = d_out()
    tr.train_step(mel, real, spk, emo)                     # D step then G step
    y1 = d_out()
    assert not torch.equal(y0, y1)                         # the D update is visible through the cache
    assert torch.equal(y1, fresh())                        # and equals a from-scratch pack
    n_entries = len(disc_fused._packs.d)
    def zero_grads(opt):                                   # explicit zero gradients (a None gradient leaves a parameter untouched,
        for p in opt.params:                               # like torch.optim.AdamW): weight decay alone then moves the weights
            p.grad = torch.zeros_like(p)
    zero_grads(tr.generator_optimizer)
    tr.generator_optimizer.step()                          # G-only update
    assert torch.equal(d_out(), y1) and len(disc_fused._packs.d) == n_entries
    zero_grads(tr.discriminator_optimizer)
    tr.discriminator_optimizer.step()                      # D-only update (weight decay)
    y2 = d_out()
    assert not torch.equal(y2, y1) and torch.equal(y2, fresh())
