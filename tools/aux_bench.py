#!/usr/bin/env python3
"""Secondary measurements on one MI355X: the bandwidth-bound kernels against the HBM roofline and the inference
configuration of the reference's test script (--modality both, batch 1024, eval).  Prints one JSON object.

    python tools/aux_bench.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-sentiment-analysis_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from mmsa_amd import data, ops  # noqa: E402
from mmsa_amd.frontend import LogMelFrontEnd  # noqa: E402
from mmsa_amd.model import MultimodalSentimentModel  # noqa: E402
from mmsa_amd.trainer import TrainStep  # noqa: E402

DEV = "cuda"
HBM_PEAK = 8000.0  # GB/s


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    out = {}
    # front end: 264 600 B in + 33 280 B out per utterance
    fe = LogMelFrontEnd(DEV)
    for B in (32, 1024, 4096):
        wave = torch.randn(B, 66150, device=DEV) * 0.1
        spec = torch.empty(B, 1, 64, 130, device=DEV)
        t = timeit(lambda: fe(wave, out=spec))
        gbs = B * 297880 / t / 1e9
        out[f"frontend_B{B}"] = {"us": round(t * 1e6, 1), "utt_per_s": round(B / t), "GB_per_s": round(gbs, 1),
                                 "frac_of_hbm_peak": round(gbs / HBM_PEAK, 4)}
    # resampler 16 kHz -> 22.05 kHz: 48 000 samples in, 66 150 out per utterance (4 B each), 334 taps per output
    from mmsa_amd.frontend import AudioTransform, Resampler
    rs = Resampler(16000, 22050, DEV)
    for B in (32, 1024):
        pcm = torch.randn(B, 1, 48000, device=DEV) * 0.1
        res = torch.empty(B, 66150, device=DEV)
        t = timeit(lambda: rs(pcm, out=res))
        gbs = B * (48000 + 66150) * 4 / t / 1e9
        out[f"resample16k_B{B}"] = {"us": round(t * 1e6, 1), "utt_per_s": round(B / t), "GB_per_s": round(gbs, 1),
                                    "frac_of_hbm_peak": round(gbs / HBM_PEAK, 4),
                                    "tflops": round(B * 66150 * 334 * 2 / t / 1e12, 2)}
    # augmented training transform for a batch (waveform augmentation + mel stage with masks)
    for gate in (False, True):
        tf = AudioTransform(augment=True, device=DEV, noise_reduce=gate)
        for B in (32, 1024):
            wave = torch.randn(B, 66150, device=DEV) * 0.1
            drawn = [tf.draw_augmentation(66150) for _ in range(B)]
            rows, masks = [d[0] for d in drawn], [d[1] for d in drawn]
            t = timeit(lambda: tf.augment_batch(wave, rows, masks, seed=1), reps=10)
            out[f"augment_batch{'_gated' if gate else ''}_B{B}"] = {"us": round(t * 1e6, 1), "utt_per_s": round(B / t)}
    # spectral-gating noise reduction of 3-second utterances.  Bytes moved per utterance with nseg STFT frames of the
    # padded chunk and nfr frames reaching the output (rows of 516 elements): spectrum written once and read three
    # times (8 B), smoothed magnitude / mask written twice and read twice (4 B), segments written and read (4 KiB
    # per frame), waveform in and out.
    from mmsa_amd.frontend import NoiseReducer
    nr = NoiseReducer(22050, DEV)
    nseg, nfr = 1 + (66150 + 60000) // 256, 262
    per_utt = nseg * 516 * (8 * 3 + 4 * 4) + nfr * (516 * 8 + 4096 * 2) + 66150 * 8
    for B in (32, 256):
        wave = torch.randn(B, 66150, device=DEV) * 0.1
        res = torch.empty(B, 66150, device=DEV)
        t = timeit(lambda: nr(wave, out=res), reps=10)
        gbs = B * per_utt / t / 1e9
        out[f"noise_gate_B{B}"] = {"us": round(t * 1e6, 1), "utt_per_s": round(B / t), "GB_per_s": round(gbs, 1),
                                   "frac_of_hbm_peak": round(gbs / HBM_PEAK, 4), "bytes_per_utt": per_utt}
    # embedding gather: 1 KiB read + 1 KiB written per token
    table = torch.randn(30522, 256, device=DEV)
    for B in (32, 4096):
        ids = torch.randint(0, 30522, (B, 128), device=DEV)
        emb = torch.empty(B, 128, 256, device=DEV)
        t = timeit(lambda: ops.embedding_gather(ids, table, emb, None))
        gbs = B * 262144 / t / 1e9
        out[f"gather_B{B}"] = {"us": round(t * 1e6, 1), "GB_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK, 4)}
    # fused clip + AdamW over the 'both' parameter set: 28 B/param (+ 8 B/param for the norm pass)
    torch.manual_seed(42)
    model = MultimodalSentimentModel("both", audio_dropout=0.0, text_dropout=0.0, fusion_dropout=0.0).to(DEV).train()
    ts = TrainStep(model)
    nparam = sum(p.numel() for p in model.parameters())
    ts.g.normal_(0, 1e-3)
    ops.param_norms(ts.arena.flat, ts.arena.table, ts.arena.tnorm, None, False)
    t = timeit(lambda: ts._update(True))
    gbs = nparam * 36 / t / 1e9
    out["clip_adamw"] = {"us": round(t * 1e6, 1), "GB_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK, 4),
                         "bytes_per_param": 36}
    # the whole chain from decoded 16 kHz PCM: mono / resample (full length) -> noise gate -> training step (mel stage,
    # forward, backward, clip, AdamW), 32 utterances of 4 s
    model.train()
    ts2 = TrainStep(model, frontend=fe, use_graph=True)
    b32 = data.make_batch(0, 32, with_wave=True)
    ids32 = torch.from_numpy(b32["input_ids"]).to(DEV)
    lab32 = torch.from_numpy(b32["label"]).to(DEV)
    pcm = torch.randn(32, 1, 64000, device=DEV) * 0.1
    n_res = rs.resampled_length(64000)
    res = torch.empty(32, n_res, device=DEV)
    den = torch.empty(32, n_res, device=DEV)

    def chain():
        rs(pcm, n_out=n_res, out=res)
        nr(res, out=den)
        ts2.step(wave=den[:, :66150], input_ids=ids32, labels=lab32)
    t = timeit(chain, reps=20, warm=5)
    out["pcm16k_to_train_step_B32"] = {"ms": round(t * 1e3, 3), "utt_per_s": round(32 / t)}
    # BASELINE config 2 / config 1 shapes as graph-replayed training steps with the reference's default dropouts
    # (multimodal_train.py:71-76): audio-only from decoded 16 kHz PCM (mono / resample -> noise gate -> mel -> step),
    # text-only from token ids
    torch.manual_seed(42)
    m_audio = MultimodalSentimentModel("audio").to(DEV).train()
    ts_a = TrainStep(m_audio, frontend=fe, use_graph=True)

    def audio_chain():
        rs(pcm, n_out=n_res, out=res)
        nr(res, out=den)
        ts_a.step(wave=den[:, :66150], labels=lab32)
    t = timeit(audio_chain, reps=20, warm=5)
    out["audio_train_B32_from_pcm16k"] = {"ms": round(t * 1e3, 3), "utt_per_s": round(32 / t), "graph": ts_a._graph is not None}
    t = timeit(lambda: ts_a.step(wave=den[:, :66150], labels=lab32), reps=20, warm=5)
    out["audio_train_B32_from_wave22k"] = {"ms": round(t * 1e3, 3), "utt_per_s": round(32 / t)}
    m_text = MultimodalSentimentModel("text").to(DEV).train()
    ts_t = TrainStep(m_text, use_graph=True)
    t = timeit(lambda: ts_t.step(input_ids=ids32, labels=lab32), reps=20, warm=5)
    out["text_train_B32"] = {"ms": round(t * 1e3, 3), "utt_per_s": round(32 / t), "graph": ts_t._graph is not None}
    # the reference's own loop body (multimodal_train.py:175-202) unchanged on the drop-in model, beside TrainStep:
    # zero_grad -> model(**inputs, labels) -> backward -> clip_grad_norm_ -> torch.optim.AdamW.step -> loss.item()
    torch.manual_seed(42)
    m_ref = MultimodalSentimentModel("both").to(DEV).train()
    opt = torch.optim.AdamW(m_ref.parameters(), lr=1e-3, weight_decay=1e-4)
    spec32 = fe(torch.from_numpy(b32["wave"]).to(DEV))
    mask32 = (ids32 != 0).long()

    def reference_loop_step():
        opt.zero_grad()
        o = m_ref(audio=spec32, input_ids=ids32, attention_mask=mask32, labels=lab32)
        o["loss"].backward()
        torch.nn.utils.clip_grad_norm_(m_ref.parameters(), max_norm=1.0)
        opt.step()
        _, preds = torch.max(o["logits"], 1)
        preds.cpu()
        return o["loss"].item()
    t = timeit(reference_loop_step, reps=20, warm=5)
    out["reference_loop_on_dropin_model_B32"] = {"ms": round(t * 1e3, 3), "utt_per_s": round(32 / t)}
    for graph in (False, True):
        torch.manual_seed(42)
        m2 = MultimodalSentimentModel("both").to(DEV).train()
        ts3 = TrainStep(m2, use_graph=graph)
        t = timeit(lambda: ts3.step(audio=spec32, input_ids=ids32, labels=lab32), reps=30, warm=5)
        out[f"trainstep_{'graph' if graph else 'eager'}_from_spectrograms_B32"] = {"ms": round(t * 1e3, 3),
                                                                                   "utt_per_s": round(32 / t)}
    # config 4: the attention-fusion head on pooled features (text [8,128], audio [8,768]), the reference's loop
    # (early_fusion.py:716-733: loss / 2, two micro-batches per update, clip 1.0 over the head, AdamW 1e-4 / 0.005)
    from mmsa_amd.fusion_head import HybridFusionHead
    torch.manual_seed(42)
    fh = HybridFusionHead(128, 768, 512, 3, 0.3).to(DEV).train()
    fopt = torch.optim.AdamW(fh.parameters(), lr=1e-4, weight_decay=0.005)
    tf8, af8 = torch.randn(8, 128, device=DEV), torch.randn(8, 768, device=DEV)
    y8 = torch.randint(0, 3, (8,), device=DEV)

    def head_update():
        fopt.zero_grad()
        for _ in range(2):
            (fh(tf8, af8, labels=y8)["loss"] / 2).backward()
        torch.nn.utils.clip_grad_norm_(fh.parameters(), max_norm=1.0)
        fopt.step()
    t = timeit(head_update, reps=30, warm=5)
    out["fusion_head_B8_accum2_update"] = {"ms": round(t * 1e3, 3), "utt_per_s": round(16 / t)}
    # the same head assembled from stock torch layers on the same GPU (what the reference runs), same loop
    nn = torch.nn

    class StockHead(nn.Module):
        def __init__(self):
            super().__init__()
            self.tp, self.ap = nn.Linear(128, 512), nn.Linear(768, 512)
            self.tn, self.an = nn.LayerNorm(512), nn.LayerNorm(512)
            self.att = nn.Sequential(nn.Linear(512, 256), nn.Tanh(), nn.Dropout(0.3), nn.Linear(256, 1))
            self.gate = nn.Sequential(nn.Linear(1024, 512), nn.Sigmoid())
            self.cls = nn.Sequential(nn.Linear(512, 256), nn.LayerNorm(256), nn.GELU(), nn.Dropout(0.3),
                                     nn.Linear(256, 64), nn.LayerNorm(64), nn.GELU(), nn.Dropout(0.3), nn.Linear(64, 3))

        def forward(self, t, a, labels):
            tp, ap = self.tn(self.tp(t)), self.an(self.ap(a))
            mod = torch.stack([tp, ap], dim=1)
            w = torch.softmax(self.att(mod).squeeze(-1), dim=1)
            wt, wa = tp * w[:, 0:1], ap * w[:, 1:2]
            g = self.gate(torch.cat([wt, wa], dim=1))
            return nn.functional.cross_entropy(self.cls(wt * g + wa * (1 - g)), labels)
    sh = StockHead().to(DEV).train()
    sopt = torch.optim.AdamW(sh.parameters(), lr=1e-4, weight_decay=0.005)

    def stock_update():
        sopt.zero_grad()
        for _ in range(2):
            (sh(tf8, af8, y8) / 2).backward()
        torch.nn.utils.clip_grad_norm_(sh.parameters(), max_norm=1.0)
        sopt.step()
    t = timeit(stock_update, reps=30, warm=5)
    out["fusion_head_B8_accum2_update_stock_torch_layers"] = {"ms": round(t * 1e3, 3), "utt_per_s": round(16 / t)}
    # inference, config 5: eval forward of 1024 utterances (front end included), predictions stay on the device
    model.eval()
    b = data.make_batch(0, 64, with_wave=True)
    wave = torch.from_numpy(b["wave"]).to(DEV).repeat(16, 1)
    ids = torch.from_numpy(b["input_ids"]).to(DEV).repeat(16, 1)

    def infer():
        with torch.no_grad():
            spec = fe(wave)
            return model(audio=spec, input_ids=ids)["logits"].argmax(1)
    t = timeit(infer, reps=5, warm=2)
    out["inference_both_B1024"] = {"ms": round(t * 1e3, 2), "utt_per_s": round(1024 / t),
                                   "tflops": round(1024 * 811711872 / t / 1e12, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
