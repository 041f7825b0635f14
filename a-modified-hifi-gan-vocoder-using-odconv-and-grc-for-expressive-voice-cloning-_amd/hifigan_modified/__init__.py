"""Drop-in ``hifigan_modified`` package backed by hand-written HIP kernels for MI355X (gfx950).

Export names follow the reference's ``hifigan_modified/__init__.py:5-14`` plus the classes its
``conditioned_hifigan.py`` imports.
"""
from .odconv import ODConv1d, ODConvTranspose1d
from .grc_lora import GRC_LoRA_Block, FiLMLayer, MultiReceptiveFieldBlock
from .generator import (ModifiedHiFiGANGenerator, HiFiGANGenerator, GroupedResidualConv1D,
                        FeatureWiseLinearModulation)
from .complete_vocoder import ModifiedHiFiGANVocoder, VocoderTrainer
from .conditioned_hifigan import ConditionedHiFiGAN, HiFiGANTrainer
from .discriminators import (HiFiGANDiscriminators, MultiPeriodDiscriminator, MultiScaleDiscriminator,
                             Discriminator1D, Discriminator2D)
from .streaming import ChunkedVocoder
from .data import MelFrontEnd, ClipSampler
from .plain_hifigan import PlainHiFiGANGenerator
from .embedding_extractors import ECAPA_TDNN, SE_Res2Block, SE_Module, Emotion2Vec, EmbeddingExtractor

__all__ = [
    "ODConv1d", "ODConvTranspose1d", "GRC_LoRA_Block", "FiLMLayer", "MultiReceptiveFieldBlock",
    "ModifiedHiFiGANGenerator", "HiFiGANGenerator", "GroupedResidualConv1D", "FeatureWiseLinearModulation",
    "HiFiGANDiscriminators", "MultiPeriodDiscriminator", "MultiScaleDiscriminator", "Discriminator1D",
    "Discriminator2D", "ModifiedHiFiGANVocoder", "VocoderTrainer", "ConditionedHiFiGAN", "HiFiGANTrainer",
    "ChunkedVocoder", "MelFrontEnd", "ClipSampler", "PlainHiFiGANGenerator",
    "ECAPA_TDNN", "SE_Res2Block", "SE_Module", "Emotion2Vec", "EmbeddingExtractor",
]
