"""Golden `forward()` outputs produced by EXECUTING THE REFERENCE'S OWN `DattnGemma2ForCausalLM.forward` (gemma.py:484-601) on CPU in fp32
with the call shapes a user of the reference can make: `logits_to_keep` 0 / 1 / 3, images only, audios only, a right-padded batch of
two videos, and `labels` (the loss of gemma.py:571-590 over the labels `prepare_inputs_labels_for_multimodal` re-aligns,
multimodal.py:358-437).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_forward.py

writes tests/golden/reference_forward.npz, checked by tests/test_generate_api.py::test_forward_call_shapes_like_the_reference."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(HERE, "reference_forward.npz")

import make_golden_dattn as MG  # noqa: E402

IDS = [[2, 21, 22, 23, -200, 24, 25, 26, 300, 301]]
IDS2 = [[2, 21, 22, 23, -200, 24, 25, 26, 300, 301], [2, 40, -200, 41, 0, 0, 0, 0, 0, 0]]
AM2 = [[1] * 10, [1] * 4 + [0] * 6]


def cases():
    """name -> (input_ids, attention_mask, labels or None, which modalities, logits_to_keep)"""
    ids, ids2, am2 = torch.tensor(IDS), torch.tensor(IDS2), torch.tensor(AM2, dtype=torch.bool)
    ones = torch.ones_like(ids, dtype=torch.bool)
    lab = torch.where(ids == 23, torch.full_like(ids, -100), ids)
    lab2 = torch.where(ids2 < 0, torch.full_like(ids2, -100), ids2)
    return {"keep0": (ids, ones, None, "va", 0), "keep1": (ids, ones, None, "va", 1), "keep3": (ids, ones, None, "va", 3),
            "images_only": (ids, ones, None, "v", 1), "audios_only": (ids, ones, None, "a", 1),
            "batch2_right_pad": (ids2, am2, None, "va", 0), "labels": (ids, ones, lab, "va", 0), "labels_batch2": (ids2, am2, lab2, "va", 0)}


def call_kwargs(case, px, mel):
    ids, am, lab, which, keep = case
    n = ids.shape[0]
    kw = dict(input_ids=ids, attention_mask=am, logits_to_keep=keep)
    if lab is not None:
        kw["labels"] = lab
    if "v" in which:
        kw["images"] = px.repeat(n, 1, 1, 1, 1)
    if "a" in which:
        kw["audios"] = mel.repeat(n, 1, 1, 1)
        kw["audio_sizes"] = [100] * n
    return kw


SEAM_NAMES = ("input_ids", "position_ids", "attention_mask", "past_key_values", "inputs_embeds", "labels", "image_embeds", "image_attention_mask",
              "audio_embeds", "audio_attention_mask")


def seam_inputs(cfg, right_pad=True):
    S, M, Fr = cfg.vis_image_size, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames
    g = torch.Generator().manual_seed(5)
    px = (torch.randn((2, 3, 3, S, S), generator=g) * 0.5).clamp(-1, 1)
    mel = torch.randn((2, 2, M, Fr), generator=g) * 0.3
    px[1, 2] = 0
    if right_pad:
        ids = torch.tensor([[2, 21, -200, 22, 23, 24], [2, 30, 31, -200, 0, 0]])
        am = torch.tensor([[1] * 6, [1] * 4 + [0] * 2], dtype=torch.bool)
    else:
        ids = torch.tensor([[1, 21, -200, 22, 23, 24], [1, 30, 31, -200, 24, 25]])
        am = torch.ones((2, 6), dtype=torch.bool)
    return dict(images=px, audios=mel, audio_sizes=[130, 60], input_ids=ids, attention_mask=am, labels=torch.where(ids < 0, torch.full_like(ids, -100), ids),
                position_ids=torch.arange(6)[None].repeat(2, 1))


def main():
    from vidi_amd.weights import init_random_weights
    import make_golden_dattn_7b as MG7
    res = {}
    # Vidi-7B (mistral.py:503-627: the same loss code, no `logits_to_keep` argument — transformers 4.44; single rows: its attention rejects right padding)
    for tag, mod, npz, only in (("", MG, "reference_dattn.npz", None), ("7b_", MG7, "reference_dattn_7b.npz", ("keep0", "labels"))):
        cfg = mod.golden_config()
        model, _ = mod.build_reference_model(cfg)
        mod.load_weights(model, init_random_weights(cfg, seed=6, dtype=torch.float32, device="cpu"))
        d = np.load(os.path.join(HERE, npz))
        px, mel = torch.from_numpy(d["A_images"]), torch.from_numpy(d["A_audios"])
        for name, case in cases().items():
            if only is not None and name not in only:
                continue
            kw = call_kwargs(case, px, mel)
            if tag:
                kw.pop("logits_to_keep")
                kw["input_ids"] = torch.where(kw["input_ids"] == 2, torch.ones_like(kw["input_ids"]), kw["input_ids"])      # BOS = 1
            with torch.no_grad():
                o = model.forward(**kw)
            if o.logits is not None:                                   # Vidi-7B returns no logits with labels (mistral.py:587-588)
                res[tag + name + "_logits"] = o.logits.float().numpy()
            if o.loss is not None:
                res[tag + name + "_loss"] = np.array([float(o.loss)])
            print(tag + name, None if o.logits is None else tuple(o.logits.shape), None if o.loss is None else float(o.loss))
    # ---- the inner seams: encode_videos (multimodal.py:254-265) and prepare_inputs_labels_for_multimodal (:339-451) on a
    # batch of two videos of different audio lengths, one with an all-zero frame, a right-padded prompt batch, labels and position ids
    for tag, mod in (("", MG), ("7b_", MG7)):
        cfg = mod.golden_config()
        model, _ = mod.build_reference_model(cfg)
        mod.load_weights(model, init_random_weights(cfg, seed=6, dtype=torch.float32, device="cpu"))
        s = seam_inputs(cfg, right_pad=not tag)               # (Vidi-7B's attention rejects right-padded batches: full rows there)
        with torch.no_grad():
            ev = model.encode_videos(s["images"], s["audios"], s["audio_sizes"])
            pr = model.prepare_inputs_labels_for_multimodal(s["input_ids"], s["position_ids"], s["attention_mask"], None, s["labels"], s["images"], None,
                                                            s["audios"], s["audio_sizes"])
        for n, t in zip(("img", "imask", "aud", "amask"), ev):
            res[tag + "seam_encode_" + n] = t.float().numpy() if t.dtype != torch.bool else t.numpy()
        for n, t in zip(SEAM_NAMES, pr):
            if t is not None:
                res[tag + "seam_prepare_" + n] = t.detach().float().numpy() if t.is_floating_point() else t.numpy()
    print("seams:", [k for k in res if k.startswith("seam_")])
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, f"{os.path.getsize(OUT) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
