"""Flat-name shim for the reference's ``from preprocess import AudioTransform, calculate_class_weights, set_seed``
(multimodal_train.py:20, multimodal_test.py:18, multimodal_dataset.py:9).  Importing it has no side effects
(the reference's module opens a log file in the working directory at import time; this one does not)."""
from mmsa_amd.frontend import (HOP_LENGTH, MAX_AUDIO_LENGTH, N_FFT, N_MELS, SAMPLE_RATE, AudioTransform,  # noqa: F401
                               BatchFrontEnd, LogMelFrontEnd, NoiseReducer, Resampler, WorkerAudioTransform,
                               calculate_class_weights, collate_utterances, set_seed)
