"""Vidi-7B golden vectors AT THE REAL DIMENSIONS, produced by executing the reference's own model code (`DattnMistralForCausalLM.forward`,
Vidi_7B/model/lmm/dattn/{mistral,multimodal,xattn}.py, the learned Conv2DPool of Vidi_7B/model/mm_vision/pool.py) on CPU / fp32 with only its absent
third-party dependencies replaced (tests/golden/ref_harness.py:install_7b).  The twin of make_golden_realdims.py for SURVEY 8 row a20:

    python tests/golden/make_golden_realdims_7b.py        (build container only)  ->  tests/golden/reference_realdims_7b.npz

Dims: Mistral-7B's hidden 4096 / 32 q, 8 kv heads x 128 / SiLU-GLU 14 336, no softcaps, untied lm_head; SigLIP 1152 / 16 x 72 / 4304 over 729 tokens; the
learned pool's 14 x 14 x 1152 -> 1152 convolution (27 -> 14 -> bilinear 2 x 2: 4 tokens per frame); Whisper 1280 / 20 x 64 / 5120; depth cut to
2 decoder / 2 + 2 tower layers, vocabulary 1 024.  Inputs: 3 frames, one 30-s window cut at 1 000 mel frames, a 39-token prompt.  Stored: token
embeddings, K / V rows of both layers (all 12 video tokens, sampled audio tokens), last hidden states, logits of the last position (Vidi-7B returns
fp32 logits).  Frames / mel / weights are regenerated from the seeds by the test."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(HERE, "reference_realdims_7b.npz")
WEIGHT_SEED, INPUT_SEED = 12, 20261001
N_FRAMES, AUDIO_SIZE, PROMPT = 3, 1000, 39


def realdims_config():
    from vidi_amd.config import vidi_7b
    cfg = vidi_7b()
    cfg.num_hidden_layers, cfg.vis_num_layers, cfg.aud_num_layers, cfg.vocab_size = 2, 3, 2, 1024
    return cfg


def make_inputs(cfg):
    g = torch.Generator().manual_seed(INPUT_SEED)
    S = cfg.vis_image_size
    px = (torch.randn((1, N_FRAMES, 3, S, S), generator=g) * 0.5).clamp(-1, 1)
    mel = torch.randn((1, 1, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), generator=g) * 0.3
    ids = torch.randint(10, cfg.vocab_size, (1, PROMPT + 1), generator=g)
    ids[0, 0], ids[0, 4] = cfg.bos_token_id, -200
    return px, mel, ids


def main():
    import make_golden_dattn_7b as M7G
    from make_golden_realdims import sample_rows
    from vidi_amd.weights import init_random_weights
    cfg = realdims_config()
    model, M7 = M7G.build_reference_model(cfg)
    w = init_random_weights(cfg, seed=WEIGHT_SEED, dtype=torch.float32, device="cpu")
    M7G.load_weights(model, w)
    del w
    px, mel, ids = make_inputs(cfg)
    am = torch.ones_like(ids, dtype=torch.bool)
    torch.set_num_threads(8)
    with torch.no_grad():
        r = model(input_ids=ids, attention_mask=am, images=px, audios=mel, audio_sizes=[AUDIO_SIZE], use_cache=True,
                  output_hidden_states=True, return_dict=True)
        (_, _, _, _, _, _, img, imask, aud, amask) = model.prepare_inputs_labels_for_multimodal(ids, None, am, None, None, px, None, mel, [AUDIO_SIZE])
    h16 = lambda t: t.detach().to(torch.float16).numpy()        # noqa: E731
    Nv, Na = img.shape[1], aud.shape[1]
    assert Nv == N_FRAMES * cfg.mm_image_pool_size ** 2 and Na == 100 and bool(imask.all()) and bool(amask.all())
    res = {"weight_seed": np.array([WEIGHT_SEED]), "input_seed": np.array([INPUT_SEED]), "input_ids": ids.numpy(), "audio_sizes": np.array([AUDIO_SIZE]),
           "image_embeds": img[0].float().numpy(), "audio_embeds": h16(aud[0]), "aud_tok": sample_rows(Na, 28, 4)}
    for li in range(cfg.num_hidden_layers):
        k, v = r.past_image_key_values[li]
        res[f"img_k_{li}"], res[f"img_v_{li}"] = k[0].float().numpy(), v[0].float().numpy()                  # all 12 video tokens, fp32
        if r.past_audio_key_values is not None and r.past_audio_key_values[li] is not None:
            k, v = r.past_audio_key_values[li]
            idx = torch.from_numpy(res["aud_tok"])
            res[f"aud_k_{li}"], res[f"aud_v_{li}"] = h16(k[0][idx]), h16(v[0][idx])
    res["prefill_hidden_last"] = h16(r.hidden_states[-1][0])
    res["prefill_logits"] = r.logits[:, -1].float().numpy()
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, f"{os.path.getsize(OUT) / 1e6:.2f} MB;", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
