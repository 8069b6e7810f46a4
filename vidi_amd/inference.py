"""Temporal-retrieval CLI of the MI355X path — the same behaviour as the reference's `inference.py`
(Vidi1.5_9B/vidi/eval/inference.py:18-91, Vidi_7B/inference.py:19-91), for users without the reference checkout:

    python -m vidi_amd.inference --video-path v.mp4 --query "a dog running" --model-path /ckpt [--arch vidi7b]

`ask()` has the reference's signature and returns the same string for the same model output: prompt construction, greedy
`model.generate(...)` call (same keyword arguments), `HH:MM:SS-HH:MM:SS` post-processing.  (The reference script itself also
runs unchanged against this package: put vidi_amd/compat or vidi_amd/compat_7b on PYTHONPATH — INTEGRATION.md §1.)"""
from __future__ import annotations

import argparse
import os
import re

import torch

from .processors import (DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX, get_media_length, load_audio, load_video, preprocess_chat,
                         preprocess_chat_mistral, process_audio, process_images, tokenizer_image_token)


def format_time_ranges(text: str, length: float, arch: str = "vidi15") -> str:
    """inference.py:52-66 (Vidi1.5: `(\\d\\.\\d+)-(\\d\\.\\d+)` -> HH:MM:SS) / Vidi_7B/inference.py:51-64 (`([\\d|\\.]+)-([\\d|\\.]+)`,
    seconds printed with `{:.2f}`)."""
    outs = []
    if arch == "vidi7b":
        for a, b in re.findall(r"([\d|\.]+)-([\d|\.]+)", text.strip()):
            t0, t1 = float(a) * length, float(b) * length
            outs.append("{:02d}:{:02d}:{:.2f}-{:02d}:{:02d}:{:.2f}".format(int(t0 / 3600), (int(t0) % 3600) // 60, int(t0) % 60,
                                                                       int(t1 / 3600), (int(t1) % 3600) // 60, int(t1) % 60))
    else:
        for a, b in re.findall(r"(\d\.\d+)-(\d\.\d+)", text.strip()):
            t0, t1 = float(a) * length, float(b) * length
            outs.append("{:02d}:{:02d}:{:02d}-{:02d}:{:02d}:{:02d}".format(int(t0 / 3600), (int(t0) % 3600) // 60, int(t0) % 60,
                                                                       int(t1 / 3600), (int(t1) % 3600) // 60, int(t1) % 60))
    return ", ".join(outs)


def build_prompt(question: str, length: float, tokenizer, arch: str = "vidi15") -> torch.Tensor:
    """inference.py:34-38 / Vidi_7B/inference.py:35-38: question template -> chat text -> ids with the -200 placeholder, [1, L]."""
    q = question[:-1] if question.endswith(".") else question
    if arch == "vidi7b":
        text = ("Given the frames from a video, answer the time range in percentage that corresponds to query text split by comma. "
                "Video length is: {:.2f} and text query is: {}.".format(length, q))
        prompt = preprocess_chat_mistral([{"from": "human", "value": DEFAULT_IMAGE_TOKEN + "\n" + text}], tokenizer)
    else:
        text = "During which time segments in the video can we see {}?".format(q)
        prompt = preprocess_chat([{"from": "human", "value": DEFAULT_IMAGE_TOKEN + "\n" + text}], tokenizer)
    return tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)


def ask(question, vid_path, model, tokenizer, image_processor, audio_processor, arch: str = "vidi15", device: str = "cuda"):
    if not os.path.exists(vid_path):
        print("Video not found.")
        raise FileNotFoundError(vid_path)
    dtype = getattr(model, "dtype", torch.float16)
    video = process_images(load_video(vid_path), image_processor, model.config).unsqueeze(0).to(dtype).to(device)
    audio_tensor, audio_size = process_audio(load_audio(vid_path, audio_processor.sampling_rate), audio_processor)
    audio = audio_tensor.unsqueeze(0).to(dtype).to(device)
    length = get_media_length(vid_path)
    input_ids = build_prompt(question, length, tokenizer, arch).to(device)
    kw = {} if arch == "vidi7b" else {"disable_compile": True}
    with torch.inference_mode():
        output_ids = model.generate(input_ids, images=video, audios=audio, audio_sizes=[audio_size], do_sample=False,
                                    max_new_tokens=1024, use_cache=True, pad_token_id=tokenizer.pad_token_id, **kw)
    outputs = tokenizer.batch_decode(output_ids, skip_special_tokens=True)[0].strip()
    return format_time_ranges(outputs, length, arch)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--video-path", type=str, required=True)
    ap.add_argument("--query", type=str, required=True)
    ap.add_argument("--model-path", type=str, required=True)
    ap.add_argument("--arch", choices=["vidi15", "vidi7b"], default="vidi15")
    a = ap.parse_args(argv)
    from .model import load_pretrained_model
    model, tokenizer, image_processor, audio_processor = load_pretrained_model(a.model_path)
    model.config.mm_splits = 32                             # inference.py:87 (a tiling hint; results do not depend on it here)
    print(ask(a.query, a.video_path, model, tokenizer, image_processor, audio_processor, arch=a.arch))


if __name__ == "__main__":
    main()
