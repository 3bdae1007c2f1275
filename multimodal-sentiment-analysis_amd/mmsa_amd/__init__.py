"""MI355X-native implementation of the CNN multimodal sentiment model's hot path.

Public surface mirrors the reference's flat modules:
  ``mmsa_amd.model``    <-> ``CNN implementation/multimodal_model.py``
  ``mmsa_amd.frontend`` <-> ``CNN implementation/preprocess.py`` (AudioTransform, set_seed)
plus :class:`mmsa_amd.trainer.TrainStep` (fused training step / data-parallel driver) and
:mod:`mmsa_amd.data` (synthetic utterances).  All arithmetic is in ``libmmsa_hip.so``.
"""
import os as _os

# HIP runtime setting (read when the runtime starts, i.e. effective if this package is imported before the first GPU
# call): graph kernel nodes are enqueued as ordinary commands instead of from pre-recorded packets.  With pre-recorded
# packets a training step replayed as a chain of several graphs was seen to run its last kernels on corrupted
# arguments; the setting costs nothing measurable here.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from .model import (Attention, AudioModel, FocalLoss, ModalityFusion, MultifactorFeatureNormalizer,  # noqa: F401
                    MultimodalSentimentModel, ResidualBlock, SEBlock, TextModel)

__all__ = ["Attention", "AudioModel", "FocalLoss", "ModalityFusion", "MultifactorFeatureNormalizer",
           "MultimodalSentimentModel", "ResidualBlock", "SEBlock", "TextModel"]
