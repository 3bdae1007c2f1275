"""MI355X-native implementation of the CNN multimodal sentiment model's hot path.

Public surface mirrors the reference's flat modules:
  ``mmsa_amd.model``    <-> ``CNN implementation/multimodal_model.py``
  ``mmsa_amd.frontend`` <-> ``CNN implementation/preprocess.py`` (AudioTransform, set_seed)
plus :class:`mmsa_amd.trainer.TrainStep` (fused training step / data-parallel driver) and
:mod:`mmsa_amd.data` (synthetic utterances).  All arithmetic is in ``libmmsa_hip.so``.
"""
import os as _os

# HIP runtime setting, read once when the runtime initialises (the first HIP call of the process): graph kernel nodes
# are enqueued as ordinary commands instead of from pre-recorded packets.  With pre-recorded packets a training step
# replayed as a CHAIN of several graphs (the multi-rank step: graph | collective | graph | ...) ran the kernels of the
# later graphs on corrupted arguments from the second replay on -- NaN gradients, or a kernel that never ends.  A
# single graph is not affected and the setting costs nothing measurable here.
#
# GRAPH_CHAIN_OK tells the trainer whether a chain of graphs may be replayed in this process: only if the setting
# is known to be "0" -- put there by the caller's environment, or by this module before torch initialised the GPU.
# (A process that touched the GPU through some other library before importing this package should export the
# variable itself.)  MMSA_FORCE_GRAPH_CHAIN=1 overrides the verdict for experiments.


def _gpu_already_initialised() -> bool:
    import sys
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


_pc = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
if _pc is None:
    GRAPH_CHAIN_OK = not _gpu_already_initialised()
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
else:
    GRAPH_CHAIN_OK = _pc.strip() == "0"
if _os.environ.get("MMSA_FORCE_GRAPH_CHAIN") == "1":
    GRAPH_CHAIN_OK = True

from .model import (Attention, AudioModel, FocalLoss, ModalityFusion, MultifactorFeatureNormalizer,  # noqa: F401
                    MultimodalSentimentModel, ResidualBlock, SEBlock, TextModel)

__all__ = ["Attention", "AudioModel", "FocalLoss", "ModalityFusion", "MultifactorFeatureNormalizer",
           "MultimodalSentimentModel", "ResidualBlock", "SEBlock", "TextModel"]
