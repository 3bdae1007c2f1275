"""MI355X-native implementation of the CNN multimodal sentiment model's hot path.

Public surface mirrors the reference's flat modules:
  ``mmsa_amd.model``    <-> ``CNN implementation/multimodal_model.py``
  ``mmsa_amd.frontend`` <-> ``CNN implementation/preprocess.py`` (AudioTransform, set_seed)
plus :class:`mmsa_amd.trainer.TrainStep` (fused training step / data-parallel driver) and
:mod:`mmsa_amd.data` (synthetic utterances).  All arithmetic is in ``libmmsa_hip.so``.
"""
from .model import (Attention, AudioModel, FocalLoss, ModalityFusion, MultifactorFeatureNormalizer,  # noqa: F401
                    MultimodalSentimentModel, ResidualBlock, SEBlock, TextModel)

__all__ = ["Attention", "AudioModel", "FocalLoss", "ModalityFusion", "MultifactorFeatureNormalizer",
           "MultimodalSentimentModel", "ResidualBlock", "SEBlock", "TextModel"]
