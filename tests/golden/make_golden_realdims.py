"""Golden vectors AT THE REAL DIMENSIONS, produced by executing the reference's own model code on CPU / fp32 ("real-dim
single-layer cases (H=3584, dh=256; SigLIP d=72, N=729)").  Everything else in tests/golden/ pins the tiny config; this file pins the
shapes the kernels are tuned for — hidden 3584, 16 q / 8 kv heads x 256, GeGLU 14 336, SigLIP 1152 / 16 heads x 72 / 4 304 over 729
tokens of a 384-px frame, Whisper 1 280 / 20 x 64 / 5 120 over 1 500 rows — with the DEPTH cut (2 decoder layers: layer 0's stream
update feeds layer 1's K/V; 2 executed SigLIP layers of 3; 2 Whisper layers) and a 1 024-word vocabulary, so that the reference runs in
about a minute on CPU and the weights (0.66 G parameters) can be regenerated from the seed by the test.

    python tests/golden/make_golden_realdims.py           (build container only: needs /root/reference)

What runs: `DattnGemma2ForCausalLM.forward(images=, audios=)` of Vidi1.5_9B/vidi/model/lmm/dattn/gemma.py — i.e.
`prepare_inputs_labels_for_multimodal` -> `encode_video_images / encode_video_audios` (the reference's SiglipVisionTower / WhisperAudioTower
wrapping HF's encoder layers, Conv2DPool, projector MLPs, norms, LearnablePosEmbd) -> `DattnGemma2Model.forward` ->
`DattnGemma2DecoderLayer.forward` x 2 — with only the absent third-party packages replaced (tests/golden/ref_harness.py).
Inputs: 2 frames of 384 x 384 (2 x 196 = 392 video tokens), one 30-s window cut at audio_size 1 000 (100 audio tokens), a 39-token
prompt.  Stored (fp16 unless noted, ~6 MB): the video / audio token embeddings (the decoder's inputs, all rows), sampled rows of the
SigLIP / Whisper tower outputs, K / V cache rows of both layers for sampled tokens, the text hidden states of the last layer and the
fp32 logits of every prompt position.  Frames / mel / weights are NOT stored: the test regenerates them from the seeds below."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(HERE, "reference_realdims.npz")

WEIGHT_SEED, INPUT_SEED = 11, 20260930
N_FRAMES, AUDIO_SIZE, PROMPT = 2, 1000, 39


def realdims_config():
    """Vidi1.5-9B's external dims (vidi_amd.config.vidi15_9b) with the depth and the vocabulary cut"""
    from vidi_amd.config import vidi15_9b
    cfg = vidi15_9b()
    cfg.num_hidden_layers, cfg.vis_num_layers, cfg.aud_num_layers, cfg.vocab_size = 2, 3, 2, 1024
    cfg.eos_token_id = 107
    return cfg


def make_inputs(cfg):
    g = torch.Generator().manual_seed(INPUT_SEED)
    S = cfg.vis_image_size
    px = (torch.randn((1, N_FRAMES, 3, S, S), generator=g) * 0.5).clamp(-1, 1)
    mel = torch.randn((1, 1, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), generator=g) * 0.3
    ids = torch.randint(10, cfg.vocab_size, (1, PROMPT + 1), generator=g)
    ids[0, 0], ids[0, 4] = cfg.bos_token_id, -200
    return px, mel, ids


def sample_rows(n, k, seed):
    rs = np.random.RandomState(seed)
    return np.array(sorted(set([0, 1, n // 2, n - 1] + rs.randint(0, n, k).tolist())), dtype=np.int64)


def main():
    import make_golden_dattn as MG
    from vidi_amd.weights import init_random_weights
    cfg = realdims_config()
    model, G = MG.build_reference_model(cfg)
    w = init_random_weights(cfg, seed=WEIGHT_SEED, dtype=torch.float32, device="cpu")
    MG.load_weights(model, w)
    del w
    px, mel, ids = make_inputs(cfg)
    am = torch.ones_like(ids, dtype=torch.bool)
    torch.set_num_threads(8)
    with torch.no_grad():
        # the towers on their own: the reference's wrappers return hidden_states[-2] (siglip.py:29-34) / the encoder output (whisper.py:26-27)
        _, vis = model.model.mm_vis(px[0])                                                             # (cls, hidden_states[-2] [2, 729, 1152])
        aud_t = model.model.mm_aud(mel[0])                                                             # [1, 1500, 1280]
        r = model(input_ids=ids, attention_mask=am, images=px, audios=mel, audio_sizes=[AUDIO_SIZE], use_cache=True,
                  output_hidden_states=True, return_dict=True)
        (_, _, _, _, _, _, img, imask, aud, amask) = model.prepare_inputs_labels_for_multimodal(ids, None, am, None, None, px, None, mel, [AUDIO_SIZE])
    h16 = lambda t: t.detach().to(torch.float16).numpy()        # noqa: E731
    res = {"weight_seed": np.array([WEIGHT_SEED]), "input_seed": np.array([INPUT_SEED]), "input_ids": ids.numpy(), "audio_sizes": np.array([AUDIO_SIZE]),
           "n_frames": np.array([N_FRAMES])}
    Nv, Na = img.shape[1], aud.shape[1]
    assert (Nv, Na) == (N_FRAMES * 196, 100) and bool(imask.all()) and bool(amask.all())
    res["image_embeds"], res["audio_embeds"] = h16(img[0]), h16(aud[0])
    res["vis_rows"] = sample_rows(729, 60, 1)
    res["vis_tower_rows"] = h16(vis[:, torch.from_numpy(res["vis_rows"])])
    res["aud_rows"] = sample_rows(1500, 60, 2)
    res["aud_tower_rows"] = h16(aud_t[:, torch.from_numpy(res["aud_rows"])])
    res["img_tok"], res["aud_tok"] = sample_rows(Nv, 92, 3), sample_rows(Na, 28, 4)
    for li in range(cfg.num_hidden_layers):
        for name, cache, rows in (("img", r.past_image_key_values, res["img_tok"]), ("aud", r.past_audio_key_values, res["aud_tok"])):
            k, v = cache[li]                                                                           # [1, N, nkv * hd] (gemma.py:59-65)
            assert k.shape[1] == (Nv if name == "img" else Na)
            idx = torch.from_numpy(rows)
            res[f"{name}_k_{li}"], res[f"{name}_v_{li}"] = h16(k[0][idx]), h16(v[0][idx])
    res["prefill_hidden_last"] = h16(r.hidden_states[-1][0])
    res["prefill_logits_all"] = r.logits[0].float().numpy()
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, f"{os.path.getsize(OUT) / 1e6:.2f} MB;", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
