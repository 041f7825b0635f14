"""Chunked ("streaming") vocoding - the caller side of the path.

The reference's streaming entry point (``streamspeech_integration.py:377-414``) hands the vocoder one mel chunk at a
time (32 frames in its demo) and runs an ordinary forward on it: every chunk is an independent utterance, so GroupNorm
statistics and the ODConv pooling are taken over the chunk, not over the sentence.  Because every op of the generator is
per-sample, the chunks of one utterance can be stacked along the batch axis and vocoded in ONE forward with the results
of the chunk-by-chunk calls (up to fp32 summation order) - that is what ``policy="independent"`` (the reference's behaviour) does: a long
utterance becomes a [n_chunks, n_mel, chunk] batch that fills the GPU instead of n_chunks tiny launches.

``policy="context"`` is this build's extension for quality: each chunk is vocoded together with ``context_frames`` of
real mel on either side (zeros beyond the utterance) and only the centre samples are kept, which hides the boundary
effects of the zero-padded convolutions (receptive field of the default generator: ~9 mel frames per side).  The
global-in-T ops (GroupNorm, ODConv attention pooling) still see the window, not the sentence; a window therefore never
equals the full-utterance forward exactly - the deviation is measured in tests/test_gpu_streaming.py and DESIGN.md.
"""
from __future__ import annotations

import torch


class ChunkedVocoder:
    def __init__(self, generator, chunk_frames: int = 32, policy: str = "independent", context_frames: int = 8):
        if policy not in ("independent", "context"):
            raise ValueError("policy must be 'independent' (reference behaviour) or 'context'")
        if chunk_frames < 1 or context_frames < 0:
            raise ValueError("chunk_frames >= 1 and context_frames >= 0 required")
        self.generator = generator
        self.chunk_frames = int(chunk_frames)
        self.policy = policy
        self.context_frames = int(context_frames) if policy == "context" else 0
        hop = 1
        for f in generator.upsample_factors:
            hop *= int(f)
        self.hop = hop

    @staticmethod
    def _rep(e, n):
        return None if e is None else e.repeat_interleave(n, dim=0)

    @torch.no_grad()
    def vocode(self, mel: torch.Tensor, speaker_emb=None, emotion_emb=None) -> torch.Tensor:
        """mel [B, n_mel, T] -> waveform [B, 1, T*hop]; the utterance is cut into chunk_frames pieces (a ragged tail is
        vocoded as its own, shorter chunk - exactly what feeding it to the reference would do)."""
        if mel.dim() != 3:
            raise ValueError("mel must be [B, n_mel, T]")
        B, C, T = mel.shape
        cf, ctx, hop = self.chunk_frames, self.context_frames, self.hop
        nfull, tail = T // cf, T % cf
        out = torch.empty(B, 1, T * hop, device=mel.device, dtype=mel.dtype)
        src = mel
        if ctx:
            src = torch.zeros(B, C, T + 2 * ctx, device=mel.device, dtype=mel.dtype)   # host-side glue: zero context beyond the utterance
            src[:, :, ctx:ctx + T] = mel
        if nfull:
            win = cf + 2 * ctx
            # [B, C, nfull, win] windows -> batch of B*nfull utterances
            w = src[:, :, :nfull * cf + 2 * ctx].unfold(2, win, cf)                      # [B, C, nfull, win]
            batch = w.permute(0, 2, 1, 3).reshape(B * nfull, C, win).contiguous()
            wav = self.generator(batch, self._rep(speaker_emb, nfull), self._rep(emotion_emb, nfull))
            wav = wav[:, :, ctx * hop:(ctx + cf) * hop]
            out[:, :, :nfull * cf * hop] = wav.reshape(B, nfull, cf * hop).reshape(B, 1, nfull * cf * hop)
        if tail:
            t0 = nfull * cf
            piece = src[:, :, t0:t0 + tail + 2 * ctx].contiguous()
            wav = self.generator(piece, speaker_emb, emotion_emb)
            out[:, :, t0 * hop:] = wav[:, :, ctx * hop:(ctx + tail) * hop]
        return out

    __call__ = vocode
