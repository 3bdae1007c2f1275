"""The front end where the reference calls it: inside DataLoader worker processes (multimodal_train.py:435-442,
multimodal_dataset.py:172).  No GPU needed: what is under test is that the GPU-backed transform refuses to run there
(instead of answering every utterance with an all-zero map) and that the CPU-only half it offers for workers yields
batches the training process can hand to the GPU half."""
import pytest
import torch
from torch.utils.data import DataLoader, Dataset

from conftest import PKG  # noqa: F401  (puts the package on sys.path)


class ToyUtterances(Dataset):
    """Four stereo utterances of different lengths at 16 kHz, shaped like multimodal_dataset.py:229-260's items."""

    def __init__(self, transform):
        self.transform = transform
        self.lengths = [16000, 40000, 52345, 8000]

    def __len__(self):
        return len(self.lengths)

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        audio = torch.randn(2, self.lengths[i], generator=g)
        return {"audio": self.transform(audio, 16000), "input_ids": torch.full((128,), i, dtype=torch.int64),
                "label": torch.tensor(i % 3)}


def test_gpu_transform_refuses_to_run_in_a_loader_worker():
    from mmsa_amd._lib import MmsaError
    from mmsa_amd.frontend import AudioTransform
    for kw in ({}, {"strict": True}, {"waveform_only": True}, {"noise_reduce": False}):
        loader = DataLoader(ToyUtterances(AudioTransform(**kw)), batch_size=2, num_workers=2)
        with pytest.raises(MmsaError, match="DataLoader worker"):
            next(iter(loader))


def test_waveform_only_without_gpu_work_is_allowed_in_a_worker():
    """22 050 Hz input, no noise gate, waveform only: nothing for the GPU to do, so workers may run it."""
    from mmsa_amd.frontend import TARGET_LENGTH, AudioTransform

    class At22k(ToyUtterances):
        def __getitem__(self, i):
            return self.transform(torch.ones(2, self.lengths[i]), 22050)

    t = AudioTransform(waveform_only=True, noise_reduce=False)
    batch = next(iter(DataLoader(At22k(t), batch_size=4, num_workers=2)))
    assert batch.shape == (4, 1, TARGET_LENGTH)
    assert float(batch[3, 0, 7999]) == 1.0 and float(batch[3, 0, 8000]) == 0.0


def test_unusable_gpu_raises_instead_of_returning_zero_maps():
    """In the main process of a machine without a GPU the default transform must not return zeros either."""
    if torch.cuda.is_available():
        pytest.skip("needs a process that cannot use a GPU")
    from mmsa_amd._lib import MmsaError
    from mmsa_amd.frontend import AudioTransform
    with pytest.raises(MmsaError):
        AudioTransform()(torch.randn(1, 22050), 22050)
    with pytest.raises(MmsaError):
        AudioTransform(noise_reduce=False)(torch.randn(1, 16000), 16000)


@pytest.mark.parametrize("max_samples", [None, 48000])
def test_worker_half_yields_batches_for_the_gpu_half(max_samples):
    from mmsa_amd.frontend import AudioTransform, WorkerAudioTransform, collate_utterances
    wt = AudioTransform(augment=True).for_workers(max_samples=max_samples)
    assert isinstance(wt, WorkerAudioTransform)
    import pickle
    pickle.loads(pickle.dumps(wt))                                  # travels to spawned workers
    ds = ToyUtterances(wt)
    loader = DataLoader(ds, batch_size=4, num_workers=2, collate_fn=None if max_samples else collate_utterances)
    batch = next(iter(loader))
    a = batch["audio"]
    want_len = [min(n, max_samples) if max_samples else n for n in ds.lengths]
    assert a["pcm"].shape == (4, max_samples or max(ds.lengths)) and a["pcm"].dtype == torch.float32
    assert a["length"].tolist() == want_len and a["sr"].tolist() == [16000] * 4
    assert batch["input_ids"].shape == (4, 128) and batch["label"].tolist() == [0, 1, 2, 0]
    for i, n in enumerate(want_len):
        g = torch.Generator().manual_seed(i)
        mono = torch.randn(2, ds.lengths[i], generator=g).mean(dim=0)          # preprocess.py:127-129
        assert torch.equal(a["pcm"][i, :n], mono[:n])
        assert float(a["pcm"][i, n:].abs().sum()) == 0.0


def test_worker_half_never_raises_and_the_datasets_fallback_map_collates():
    from mmsa_amd.frontend import SAMPLE_RATE, TARGET_LENGTH, WorkerAudioTransform, collate_utterances
    wt = WorkerAudioTransform()
    bad = wt("not audio", 16000)                                    # preprocess.py:172-174: logged, answered with silence
    assert bad["pcm"].shape == (TARGET_LENGTH,) and int(bad["sr"]) == SAMPLE_RATE and float(bad["pcm"].abs().sum()) == 0.0
    one = wt(torch.ones(5), 8000)                                   # a bare [T] vector is one channel
    assert int(one["length"]) == 5 and int(one["sr"]) == 8000
    # multimodal_dataset.py:175-178 returns a zero MAP when its own loading code fails: it joins the batch as silence
    out = collate_utterances([{"audio": one, "label": torch.tensor(1)},
                              {"audio": torch.zeros(1, 64, 129), "label": torch.tensor(2)}])
    assert out["audio"]["pcm"].shape == (2, TARGET_LENGTH) and out["audio"]["length"].tolist() == [5, TARGET_LENGTH]
    assert out["audio"]["sr"].tolist() == [8000, SAMPLE_RATE] and out["label"].tolist() == [1, 2]
    with pytest.raises(ValueError):
        WorkerAudioTransform(max_samples=0)
