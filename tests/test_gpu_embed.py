"""GPU parity of the conditioning producers: product modules (HIP kernels through the C ABI) against the
golden vectors generated from the reference's embedding_extractors.py, and the new kernels one by one against fp64 torch math.
Tolerances (rel-L2): fp32 storage 1e-4 (bf16x3-free generic fp32 kernels), fp16 5e-3, bf16 3e-2 on L2-normalised embeddings."""
import ctypes

import pytest
import torch

from conftest import load_golden
from embed_cases import CASES, build
from oracle.vocoder_oracle import rel_l2

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-4, torch.float16: 5e-3, torch.bfloat16: 3e-2}
DTYPES = [torch.float32, torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def H():
    import hifigan_modified as H
    from hifigan_modified import _native
    _native.lib()
    return H


def _x(g, dtype):
    return torch.from_numpy(g["x"]).cuda().to(dtype)


def _err(y, g, key, sl=None):
    y = y.float().cpu()
    if sl is not None:
        y = y[sl]
    return rel_l2(y, torch.from_numpy(g[key]))


@pytest.mark.parametrize("dtype", DTYPES)
def test_se_module_and_res2_block(H, dtype):
    g = load_golden("embed_se_module_c64")
    m = build("se", g, channels=64).cuda()
    assert _err(m(_x(g, dtype)), g, "y") < TOL[dtype]
    g = load_golden("embed_se_res2_c256_d3")
    m = build("res2", g, channels=256, dilation=3).cuda()
    assert _err(m(_x(g, dtype)), g, "y") < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["embed_ecapa_h512_t100", "embed_ecapa_h256_t37"])
def test_ecapa_vs_reference_golden(H, name, dtype):
    g = load_golden(name)
    m = build("ecapa", g, **CASES[name][1]).cuda()
    x = _x(g, dtype)
    pooled = m.pooled_cl(x)
    assert _err(pooled, g, "pooled") < 2 * TOL[dtype]
    emb, logits = m(x)
    assert logits is None and emb.dtype == dtype and emb.shape == (2, 192)
    assert _err(emb, g, "embedding") < TOL[dtype]
    m.train(True)                                  # frozen producer: same embedding, logits returned for arity
    emb2, logits = m(x)
    assert torch.equal(emb, emb2) and logits.shape == (2, 16)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["embed_emotion_h512_t100", "embed_emotion_h128_t37"])
def test_emotion2vec_vs_reference_golden(H, name, dtype):
    g = load_golden(name)
    m = build("emotion", g, **CASES[name][1]).cuda()
    frame, utt, logits = m(_x(g, dtype))
    assert logits is None and frame.shape == tuple(g["frame"].shape) and utt.shape == (2, 256)
    assert _err(frame, g, "frame") < 2 * TOL[dtype]
    assert _err(utt, g, "utterance") < 2 * TOL[dtype]
    # training mode: the reference feeds the [B, embedding_dim] utterance embedding to Linear(hidden_dim, .) (:250 vs :209) and
    # raises a shape error unless the two sizes agree; same behaviour here (never an out-of-bounds read)
    m.train(True)
    with pytest.raises(RuntimeError):
        m(_x(g, dtype))
    m2 = build("emotion", None, hidden_dim=256, embedding_dim=256).cuda().train(True)
    fr, ut, logits = m2(_x(g, dtype))
    assert logits.shape == (2, 8) and torch.isfinite(logits.float()).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_combined_extractor_and_vocoder_hook(H, dtype):
    g = load_golden("embed_extractor_t32")
    m = build("extractor", g).cuda()
    spk, emo = m(_x(g, dtype))
    assert _err(spk, g, "speaker") < TOL[dtype] and _err(emo, g, "emotion") < 2 * TOL[dtype]
    # ModifiedHiFiGANVocoder.forward(extract_embeddings=True) (complete_vocoder.py:65-69) conditions on the extracted embeddings
    torch.manual_seed(0)
    voc = H.ModifiedHiFiGANVocoder().cuda().train(False)
    voc.embedding_extractor.load_state_dict(m.state_dict())
    x = _x(g, dtype)
    with torch.no_grad():
        out = voc(x)
        assert torch.equal(out["speaker_embedding"], spk) and torch.equal(out["emotion_embedding"], emo)
        ref = voc(x, spk, emo, extract_embeddings=False)["generated_waveform"]
        # same inputs, two runs: bit-identical (the ODConv pooling sums are fixed-order partials, no float atomics on the inference path)
        rt = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1e-2}[dtype]
        assert torch.equal(out["generated_waveform"], ref) and ref.shape == (2, 1, 32 * 256)
        plain = voc(x, extract_embeddings=False)
        assert plain["speaker_embedding"] is None
        assert rel_l2(plain["generated_waveform"].float().cpu(), ref.float().cpu()) > 10 * rt      # FiLM skipped without embeddings
        only_spk = voc(x, speaker_embedding=spk.flip(0))
        assert torch.equal(only_spk["speaker_embedding"], spk.flip(0)) and torch.equal(only_spk["emotion_embedding"], emo)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,nh,hd", [(2, 1, 8, 64), (2, 37, 8, 64), (3, 64, 8, 64), (2, 100, 8, 16), (1, 345, 4, 32), (2, 130, 2, 64)])
def test_mha_kernel_vs_fp64(H, dtype, B, T, nh, hd):
    from hifigan_modified import _native as N_, ops
    torch.manual_seed(B * 1000 + T)
    Hd = nh * hd
    qkv = (torch.randn(B, T, 3 * Hd, device="cuda") * 1.5).to(dtype)
    out = torch.empty(B, T, Hd, device="cuda", dtype=dtype)
    N_.call("mv_mha_fwd", ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, T, nh, hd, ops._dt(qkv), ops._stream())
    q, k, v = (t.double().view(B, T, nh, hd).transpose(1, 2) for t in qkv.split(Hd, dim=-1))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1) @ v).transpose(1, 2).reshape(B, T, Hd)
    tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.2e-2}[dtype]
    assert rel_l2(out.double().cpu(), ref.cpu()) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(7, 512), (200, 128), (5, 2048), (3, 1536)])
def test_add_layernorm_vs_fp64(H, dtype, rows, C):
    from hifigan_modified import _native as N_, ops
    torch.manual_seed(rows)
    x, r = torch.randn(rows, C, device="cuda").to(dtype), torch.randn(rows, C, device="cuda").to(dtype)
    gam, bet = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    y = torch.empty_like(x)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N_.call("mv_add_layernorm", P(x), P(r), P(gam), P(bet), P(y), rows, C, 1e-5, ops._dt(x), ops._stream())
    ref = torch.nn.functional.layer_norm(x.double() + r.double(), (C,), gam.double(), bet.double(), 1e-5)
    assert rel_l2(y.double().cpu(), ref.cpu()) < {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 6e-3}[dtype]
    N_.call("mv_add_layernorm", P(x), None, P(gam), P(bet), P(y), rows, C, 1e-5, ops._dt(x), ops._stream())
    ref = torch.nn.functional.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5)
    assert rel_l2(y.double().cpu(), ref.cpu()) < {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 6e-3}[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,C", [(2, 96, 1536), (3, 33, 768), (1, 2, 64)])
def test_attentive_statistics_pooling_vs_fp64(H, dtype, B, T, C):
    from hifigan_modified import _native as N_, ops
    torch.manual_seed(T)
    x, lg = torch.randn(B, T, C, device="cuda").to(dtype), (torch.randn(B, T, C, device="cuda") * 2).to(dtype)
    ws = torch.empty(N_.lib().mv_asp_workspace_bytes(B, T), device="cuda", dtype=torch.uint8)
    pooled = torch.empty(B, 2 * C, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N_.call("mv_asp_pool", P(x), P(lg), P(ws), P(pooled), B, T, C, ops._dt(x), ops._stream())
    a = x.double() * torch.softmax(lg.double(), dim=2)
    ref = torch.cat([a.mean(dim=1), a.std(dim=1)], dim=1)
    assert rel_l2(pooled.double().cpu(), ref.cpu()) < 2e-5


def test_extractor_rejects_bad_input_and_follows_weight_updates(H):
    m = build("ecapa", None, hidden_dim=256, num_speakers=4).cuda()
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 80, 5, device="cuda"))            # valid k=5 conv + unbiased std need >= 6 frames
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 80, 50))                           # CPU tensor: no fallback
    x = torch.randn(2, 80, 50, device="cuda").half()
    e0 = m(x)[0]
    with torch.no_grad():
        m.bn1.running_mean.add_(0.5)                        # buffers are part of the folded weights: cache must notice
    e1 = m(x)[0]
    assert not torch.equal(e0, e1)
    with torch.no_grad():
        m.bn1.running_mean.sub_(0.5)
    assert rel_l2(m(x)[0].float().cpu(), e0.float().cpu()) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N,act", [(1024, 512, 512, "relu"), (1000, 2048, 512, "none"), (37, 512, 2048, "relu"), (64, 1536, 512, "tanh"),
                                         (200, 256, 768, "none"), (33, 768, 256, "relu"), (96, 1024, 64, "none")])
def test_skinny_gemm_vs_fp64(H, dtype, M, K, N, act):
    from hifigan_modified import _native as N_, ops
    torch.manual_seed(M + K)
    x = torch.randn(M, K, device="cuda").to(dtype)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5)
    b = (torch.randn(N, device="cuda") * 0.1).to(dtype)
    packed = ops.dconv_pack(w.view(N, K, 1, 1), dtype, 0)
    y = torch.empty(M, N, device="cuda", dtype=dtype)
    kind = {"relu": N_.ACT_LRELU, "none": N_.ACT_NONE, "tanh": N_.ACT_TANH}[act]
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N_.call("mv_gemm_cl_skinny", P(x), P(packed), P(b), P(y), M, K, N, kind, 0.0, ops._dt(x), ops._stream())
    ref = x.double() @ w.to(dtype).double().t() + b.double()
    ref = {"relu": torch.relu, "none": lambda t: t, "tanh": torch.tanh}[act](ref)
    assert rel_l2(y.double().cpu(), ref.cpu()) < {torch.float16: 1e-3, torch.bfloat16: 6e-3}[dtype]
    # unsupported shapes are refused, not mis-computed
    assert N_.lib().mv_gemm_cl_skinny(P(x), P(packed), P(b), P(y), M, K + 32, N, kind, 0.0, ops._dt(x), ops._stream()) == -3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("channels,dil,T", [(512, 4, 150), (512, 2, 64), (256, 3, 65), (512, 1, 7), (256, 4, 300)])
def test_fused_res2_chain_matches_conv_by_conv(H, dtype, channels, dil, T):
    from hifigan_modified import embedding_extractors as EE
    m = build("res2", None, channels=channels, dilation=dil).cuda()
    torch.manual_seed(T)
    x = torch.randn(3, channels, T, device="cuda").to(dtype)
    y = m(x)
    EE._UNFUSED_RES2 = True
    try:
        y_ref = m(x)
    finally:
        EE._UNFUSED_RES2 = False
    # the fused chain adds xs[i] + ys[i-1] in fp32 (one rounding), the launch-per-conv path rounds ys[i-1] first
    assert rel_l2(y.float().cpu(), y_ref.float().cpu()) < {torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]


def test_trainer_step_without_embeddings_uses_the_extractor(H):
    """complete_vocoder.py:207: VocoderTrainer.train_step calls self.vocoder(mel) with no embeddings -> extracted ones condition
    the generator; the frozen extractor is not touched by either optimizer."""
    torch.manual_seed(0)
    voc = H.ModifiedHiFiGANVocoder(hidden_channels=64)
    tr = H.VocoderTrainer(voc, device=torch.device("cuda"))
    before = {k: v.detach().clone() for k, v in voc.embedding_extractor.state_dict().items()}
    torch.manual_seed(1)
    mel = torch.randn(2, 80, 8, device="cuda")
    real = torch.randn(2, 1, 8 * 256, device="cuda").clamp(-1, 1)
    out = tr.train_step(mel, real)
    assert all(torch.isfinite(torch.tensor(float(v))) for v in out.values())
    after = voc.embedding_extractor.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    assert all(p.grad is None for p in voc.embedding_extractor.parameters())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_long_batch_takes_the_tiled_gemm_path_and_matches_the_oracle(H, dtype):
    """B*T > 4096 positions: the k=1 layers run on the tiled MFMA conv kernel (flattened over the batch) instead of the split-K GEMM.
    Samples are independent, so the first two rows of the big batch must match the CPU oracle on those two samples alone."""
    from oracle import embed_oracle as E
    m = build("extractor", None).cuda()
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    torch.manual_seed(5)
    mel = torch.randn(16, 80, 300, device="cuda")
    spk, emo = m(mel.to(dtype))
    spk_o, emo_o = E.embedding_extractor(mel[:2].cpu(), sd)
    assert rel_l2(spk[:2].float().cpu(), spk_o) < TOL[dtype] and rel_l2(emo[:2].float().cpu(), emo_o) < 2 * TOL[dtype]
    # and the same samples through the short-sequence path agree with the long-batch result
    spk_s, emo_s = m(mel[:2].to(dtype))
    assert rel_l2(spk_s.float().cpu(), spk[:2].float().cpu()) < TOL[dtype]
    assert rel_l2(emo_s.float().cpu(), emo[:2].float().cpu()) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_extractor_edge_shapes(H, dtype):
    from oracle import embed_oracle as E
    torch.manual_seed(9)
    emo_m = build("emotion", None, hidden_dim=128).cuda()
    sd = {k: v.detach().float().cpu() for k, v in emo_m.state_dict().items()}
    for B, T in [(1, 1), (1, 2), (3, 65)]:                     # single frame, single sample, one past the attention key block
        x = torch.randn(B, 80, T)
        fr, ut, _ = emo_m(x.cuda().to(dtype))
        fr_o, ut_o = E.emotion2vec(x, sd)
        assert fr.shape == (B, T, 256) and rel_l2(fr.float().cpu(), fr_o) < 2 * TOL[dtype], (B, T)
        assert rel_l2(ut.float().cpu(), ut_o) < 2 * TOL[dtype], (B, T)
    spk_m = build("ecapa", None, hidden_dim=256, num_speakers=4).cuda()
    sd = {k: v.detach().float().cpu() for k, v in spk_m.state_dict().items()}
    for B, T in [(1, 6), (2, 7), (1, 129)]:                    # shortest legal input (2 frames after the valid conv), tile edges
        x = torch.randn(B, 80, T)
        e = spk_m(x.cuda().to(dtype))[0]
        assert rel_l2(e.float().cpu(), E.ecapa_tdnn(x, sd)) < 3 * TOL[dtype], (B, T)
    with pytest.raises(RuntimeError):
        spk_m(torch.randn(2, 80, device="cuda"))               # not [B, C, T]


def test_extractor_with_16bit_parameters_matches_fp32_masters(H):
    """`module.half()` (parameters stored in fp16) must give the embeddings of the fp32-master module on the same fp16 input: every
    kernel that takes fp32 weights gets a converted copy, never the raw 16-bit storage."""
    m = build("extractor", None).cuda()
    torch.manual_seed(11)
    x = torch.randn(2, 80, 40, device="cuda").half()
    spk, emo = m(x)
    import copy
    mh = copy.deepcopy(m).half()
    spk_h, emo_h = mh(x)
    assert rel_l2(spk_h.float().cpu(), spk.float().cpu()) < 3e-3 and rel_l2(emo_h.float().cpu(), emo.float().cpu()) < 6e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_stream_extractor_equals_one_stream_eager_and_captured(H, dtype):
    """EmbeddingExtractor runs its two encoders as parallel stream branches (embedding_extractors.py): the embeddings are the bits of
    the one-stream forward, issued eagerly and replayed from a captured graph (GraphedExtractor), and repeat bit for bit."""
    from hifigan_modified import embedding_extractors as E
    from hifigan_modified.graphs import GraphedExtractor
    torch.manual_seed(3)
    ex = H.EmbeddingExtractor().cuda().train(False)
    mel = torch.randn(5, 80, 48, device="cuda").to(dtype)
    if dtype != torch.float32:
        ex = ex.to(dtype)
    saved = E._TWO_STREAMS
    try:
        E._TWO_STREAMS = False
        s1, e1 = [t.clone() for t in ex(mel)]
        E._TWO_STREAMS = True
        s2, e2 = [t.clone() for t in ex(mel)]
        s3, e3 = [t.clone() for t in ex(mel)]
        g = GraphedExtractor(ex, mel)
        sg, eg = [t.clone() for t in g(mel)]
        sg2, eg2 = [t.clone() for t in g(mel)]
    finally:
        E._TWO_STREAMS = saved
    torch.cuda.synchronize()
    for a, b in ((s1, s2), (e1, e2), (s2, s3), (e2, e3), (s1, sg), (e1, eg), (sg, sg2), (eg, eg2)):
        assert torch.equal(a, b)
