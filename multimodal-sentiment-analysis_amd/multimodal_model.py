"""Flat-name shim: ``from multimodal_model import MultimodalSentimentModel, FocalLoss`` keeps working when this
directory replaces the reference's ``CNN implementation/`` on ``sys.path`` (the reference's drivers import it that
way: multimodal_train.py:19-21, multimodal_test.py:17-19)."""
from mmsa_amd.model import (Attention, AudioModel, FocalLoss, ModalityFusion, MultifactorFeatureNormalizer,  # noqa: F401
                            MultimodalSentimentModel, ResidualBlock, SEBlock, TextModel)
