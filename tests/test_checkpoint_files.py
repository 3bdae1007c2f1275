"""Files exchanged with the reference's scripts: best_model.pt, model_info.json, <modality>_test_predictions.csv."""
import io
import json
import os

import numpy as np
import pytest
import torch

from conftest import SMALL_VOCAB, batch_tensors, seeded_model

DEV = "cuda"


def test_predictions_csv_is_byte_identical_to_pandas(tmp_path):
    """multimodal_test.py:298-309 builds a DataFrame of float32 probabilities and calls to_csv(index=False)."""
    import pandas as pd
    from mmsa_amd import checkpoint
    rng = np.random.default_rng(0)
    n = 500
    probs = rng.random((n, 3)).astype(np.float32)
    probs[0] = [1e-7, 1 - 1e-7, 0.0]
    probs[1] = [1.0, 0.0, 3e-5]
    probs[2] = [1e-4, 9.99e-5, 1.234e-20]
    probs[3] = [0.5, 0.25, 0.25]
    labels = rng.integers(0, 3, n)
    preds = rng.integers(0, 3, n)
    ids = [f"dia{i}_utt{i % 7}" for i in range(n)]
    ids[5] = 'has,comma'
    ids[6] = 'has"quote'
    names = checkpoint.CLASS_NAMES
    df = pd.DataFrame({
        "Audio_ID": ids,
        "True_Sentiment": [names[i] for i in labels],
        "Predicted_Sentiment": [names[i] for i in preds],
        "Negative_Prob": [p[0] for p in probs],
        "Neutral_Prob": [p[1] for p in probs],
        "Positive_Prob": [p[2] for p in probs],
    })
    want = df.to_csv(index=False)
    path = checkpoint.write_predictions_csv(str(tmp_path / "both_test_predictions.csv"), ids, labels, preds, probs)
    assert open(path, newline="").read() == want
    with pytest.raises(ValueError):
        checkpoint.write_predictions_csv(str(tmp_path / "x.csv"), ids[:3], labels, preds, probs)


def test_model_info_and_checkpoint_round_trip_on_cpu(tmp_path):
    from mmsa_amd import checkpoint
    m = seeded_model("both", vocab=SMALL_VOCAB)
    info = json.load(open(checkpoint.write_model_info(str(tmp_path), m, np.float64(0.625))))
    assert info == {"modality": "both", "audio_feature_dim": 256, "text_feature_dim": 384, "fusion_dim": 256,
                    "best_val_f1": 0.625}
    assert list(info) == ["modality", "audio_feature_dim", "text_feature_dim", "fusion_dim", "best_val_f1"]
    path = str(tmp_path / "best_model.pt")
    ck = checkpoint.save_checkpoint(path, m, None, epoch=3, val_loss=0.9, val_f1=0.5)
    assert list(ck) == ["epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "val_loss",
                        "val_f1", "modality", "config"]
    back = checkpoint.read_checkpoint(path)
    assert back["epoch"] == 3 and back["modality"] == "both" and back["config"] == ck["config"]
    for k, v in m.state_dict().items():
        assert torch.equal(back["model_state_dict"][k], v), k


def test_reference_style_checkpoint_with_numpy_scalars_loads_safely(tmp_path):
    """The reference's val_f1 is a numpy float64 (sklearn's f1_score) inside the pickled dict: it must load through
    the restricted unpickler; modality falls back to the run directory's name (multimodal_test.py:130-141)."""
    from mmsa_amd import checkpoint
    m = seeded_model("text", vocab=SMALL_VOCAB)
    d = tmp_path / "text_20250101_000000"
    d.mkdir()
    path = str(d / "best_model.pt")
    torch.save({"epoch": 1, "model_state_dict": m.state_dict(), "optimizer_state_dict": None,
                "scheduler_state_dict": None, "val_loss": 1.0, "val_f1": np.float64(0.4), "config": {}}, path)
    ck = checkpoint.read_checkpoint(path)
    assert float(ck["val_f1"]) == 0.4
    assert checkpoint.infer_modality(ck, path) == "text"
    bare = str(tmp_path / "audio_run" / "best_model.pt")
    os.makedirs(os.path.dirname(bare))
    torch.save(seeded_model("audio").state_dict(), bare)
    assert checkpoint.infer_modality(checkpoint.read_checkpoint(bare), bare) == "audio"


@pytest.mark.gpu
def test_resume_continues_the_run_and_state_loads_into_torch_adamw(tmp_path):
    """Two steps, save, restore into a fresh model + TrainStep, one more step == three uninterrupted steps;
    the optimiser state is accepted by torch.optim.AdamW over the same parameters (the reference's optimiser)."""
    from mmsa_amd import checkpoint
    from mmsa_amd.trainer import TrainStep

    def inputs(s):
        b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB, device=DEV)
        return dict(audio=b["audio"], input_ids=b["input_ids"], labels=b["label"])

    def fresh():
        m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        return m

    ma = fresh()
    ta = TrainStep(ma)
    for s in range(2):
        ta.step(**inputs(s))
    path = str(tmp_path / "best_model.pt")
    checkpoint.save_checkpoint(path, ma, ta, epoch=1, val_loss=1.0, val_f1=0.3)
    la = float(ta.step(**inputs(2)))

    mb = fresh()
    tb = TrainStep(mb, lr=5e-4)
    ck = checkpoint.resume(path, mb, tb)
    assert ck["epoch"] == 1 and tb.lr == 1e-3 and float(tb.state[0]) == 2.0
    lb = float(tb.step(**inputs(2)))
    # the step is bit-reproducible and the checkpoint holds every bit of its state
    assert la == lb
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), k

    opt = torch.optim.AdamW(mb.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.load_state_dict(ta.state_dict())
    assert len(opt.state) == len(list(mb.parameters()))
    tc = TrainStep(fresh())
    tc.load_state_dict(opt.state_dict())
    assert torch.equal(tc.m, ta.m) and torch.equal(tc.v, ta.v) and float(tc.state[0]) == 3.0

    model, modality = checkpoint.load_model(path, device=DEV)
    assert modality == "both"
    model.eval()
    ma2 = fresh().eval()
    ma2.load_state_dict(checkpoint.read_checkpoint(path)["model_state_dict"])
    b = batch_tensors(0, 4, vocab=SMALL_VOCAB, device=DEV)
    with torch.no_grad():
        o1 = model(audio=b["audio"], input_ids=b["input_ids"], attention_mask=b["attention_mask"])
        o2 = ma2(audio=b["audio"], input_ids=b["input_ids"], attention_mask=b["attention_mask"])
    assert torch.equal(o1["logits"], o2["logits"])


@pytest.mark.gpu
def test_restoring_weights_into_a_model_with_a_captured_step():
    """In-place restore (model.load_state_dict + TrainStep.load_state_dict, the 'back to the best epoch' pattern) while
    a captured launch sequence exists: the replays that follow start from the restored bits, parameter norms included."""
    import copy
    from mmsa_amd.trainer import TrainStep

    def inputs(s):
        b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB, device=DEV)
        return dict(audio=b["audio"], input_ids=b["input_ids"], labels=b["label"])

    m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    m.fusion_layer.draw_modality_dropout = lambda: (False, False)
    ts = TrainStep(m, use_graph=True)
    for s in range(4):
        ts.step(**inputs(s))
    assert ts._graph is not None
    model_sd = copy.deepcopy(m.state_dict())
    opt_sd = copy.deepcopy(ts.state_dict())
    first = [float(ts.step(**inputs(s))) for s in (4, 5)]
    after = m._arena.flat.clone()
    m.load_state_dict(model_sd)
    ts.load_state_dict(opt_sd)
    again = [float(ts.step(**inputs(s))) for s in (4, 5)]
    assert first == again
    assert torch.equal(after, m._arena.flat)
