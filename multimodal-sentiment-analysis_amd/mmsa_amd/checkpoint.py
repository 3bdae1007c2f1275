"""Files the reference's training and test scripts exchange, readable and writable from this build.

* ``best_model.pt`` (`CNN implementation/multimodal_train.py:343-357`): a dict with ``epoch``,
  ``model_state_dict``, ``optimizer_state_dict`` (``torch.optim.AdamW`` layout), ``scheduler_state_dict``,
  ``val_loss``, ``val_f1``, ``modality`` and ``config`` -- :func:`save_checkpoint` writes exactly these keys from a
  model and a :class:`~mmsa_amd.trainer.TrainStep`; :func:`load_model` mirrors `multimodal_test.py:123-163`
  (modality from the file, else from the run directory's name) and also accepts a bare state dict.
  Files are read with ``weights_only=True``: nothing in them is executed.
* ``model_info.json`` (`multimodal_train.py:377-386`): :func:`write_model_info`.
* ``<modality>_test_predictions.csv`` (`multimodal_test.py:298-309`): :func:`write_predictions_csv`, byte-identical
  to the ``DataFrame.to_csv(index=False)`` the reference calls (float32 probabilities in shortest round-trip form).

The state-dict keys are the reference's own, so a checkpoint trained there loads here and vice versa.
"""
from __future__ import annotations

import csv
import json
import os
from typing import Optional, Sequence

import numpy as np
import torch

from .model import MultimodalSentimentModel

CLASS_NAMES = ["negative", "neutral", "positive"]


def _config(model) -> dict:
    return {"audio_feature_dim": getattr(model, "audio_feature_dim", 0),
            "text_feature_dim": getattr(model, "text_feature_dim", 0),
            "fusion_dim": getattr(model, "fusion_dim", 0)}


def save_checkpoint(path: str, model: MultimodalSentimentModel, train_step=None, epoch: int = 0,
                    val_loss: float = 0.0, val_f1: float = 0.0, scheduler=None) -> dict:
    """Writes the reference's ``best_model.pt`` dictionary; returns it.  ``epoch`` is 1-based as in the reference."""
    ckpt = {
        "epoch": int(epoch),
        "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "optimizer_state_dict": _cpu(train_step.state_dict()) if train_step is not None else None,
        "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None,
        "val_loss": float(val_loss),
        "val_f1": float(val_f1),
        "modality": model.modality,
        "config": _config(model),
    }
    torch.save(ckpt, path)
    return ckpt


def _cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


def read_checkpoint(path: str, map_location="cpu") -> dict:
    """``torch.load(..., weights_only=True)``.  The reference stores ``val_f1`` as a numpy float64 scalar (it comes
    from sklearn), which the restricted unpickler only admits once numpy's scalar constructors are allow-listed;
    they build plain numbers and execute nothing from the file."""
    allow = []
    try:
        import numpy.core.multiarray as _ma        # noqa: F401  (numpy < 2 pickles name this module)
        allow.append(_ma.scalar)
    except Exception:
        pass
    try:
        import numpy._core.multiarray as _ma2
        allow.append(_ma2.scalar)
    except Exception:
        pass
    allow += [np.dtype, type(np.dtype(np.float64)), type(np.dtype(np.float32)), type(np.dtype(np.int64))]
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location=map_location, weights_only=True)


def infer_modality(checkpoint: dict, model_path: str) -> str:
    """multimodal_test.py:130-141."""
    if isinstance(checkpoint, dict) and "modality" in checkpoint:
        return checkpoint["modality"]
    name = os.path.basename(os.path.dirname(model_path))
    if "audio" in name and "text" not in name:
        return "audio"
    if "text" in name and "audio" not in name:
        return "text"
    return "both"


def load_model(model_path: str, modality: Optional[str] = None, device="cuda", **model_kwargs):
    """(model, modality), as multimodal_test.py:123-163; the model is left in training mode like the reference's.
    The vocabulary size and embedding width are taken from the stored embedding table (the reference always builds
    the default 30 522 x 256 one), other constructor arguments from ``model_kwargs``."""
    ckpt = read_checkpoint(model_path)
    if modality is None:
        modality = infer_modality(ckpt, model_path)
    sd = ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt
    emb = sd.get("text_model.embedding.weight")
    if emb is not None:
        model_kwargs.setdefault("vocab_size", emb.shape[0])
        model_kwargs.setdefault("embedding_dim", emb.shape[1])
    model = MultimodalSentimentModel(modality=modality, num_classes=3, **model_kwargs).to(device)
    model.load_state_dict(sd)
    return model, modality


def resume(model_path: str, model: MultimodalSentimentModel, train_step=None, scheduler=None) -> dict:
    """Restores model, optimiser and scheduler state from a ``best_model.pt``; returns the checkpoint dict."""
    ckpt = read_checkpoint(model_path)
    model.load_state_dict(ckpt["model_state_dict"])
    if train_step is not None and ckpt.get("optimizer_state_dict") is not None:
        train_step.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and ckpt.get("scheduler_state_dict") is not None:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return ckpt


def write_model_info(run_dir: str, model: MultimodalSentimentModel, best_val_f1: float) -> str:
    info = {"modality": model.modality, **_config(model), "best_val_f1": float(best_val_f1)}
    path = os.path.join(run_dir, "model_info.json")
    with open(path, "w") as f:
        json.dump(info, f, indent=4)
    return path


def _fmt(x) -> str:
    """pandas' CSV text of a float32 cell: numpy's shortest round-trip string (positional for 1e-4 <= |x| < 1e16,
    scientific with a two-digit exponent otherwise); NaN is an empty field."""
    x32 = np.float32(x)
    return "" if np.isnan(x32) else str(x32)


def write_predictions_csv(path: str, audio_ids: Sequence, labels: Sequence[int], preds: Sequence[int],
                          probs) -> str:
    """One row per utterance: Audio_ID, true / predicted class name, the three float32 probabilities."""
    probs = np.asarray(probs, dtype=np.float32).reshape(-1, 3)
    if not (len(audio_ids) == len(labels) == len(preds) == probs.shape[0]):
        raise ValueError("audio_ids, labels, preds and probs must have one entry per utterance")
    with open(path, "w", newline="") as f:
        w = csv.writer(f, lineterminator="\n", quoting=csv.QUOTE_MINIMAL)
        w.writerow(["Audio_ID", "True_Sentiment", "Predicted_Sentiment", "Negative_Prob", "Neutral_Prob",
                    "Positive_Prob"])
        for i in range(len(audio_ids)):
            w.writerow([str(audio_ids[i]), CLASS_NAMES[int(labels[i])], CLASS_NAMES[int(preds[i])],
                        _fmt(probs[i, 0]), _fmt(probs[i, 1]), _fmt(probs[i, 2])])
    return path
