"""Host-side processor API of the reference, kept name-for-name: these run on the CPU in
Python exactly as in bytedance/vidi — decoding, resizing, mel features, chat templating and `<image>`
splicing are not on the GPU hot path.  Implementations are our own; each cites the reference lines whose
behaviour it reproduces (`Vidi1.5_9B/vidi/dataset/{img,txt,vid}_utils.py`)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"


# ------------------------------------------------------------------------------------------------
# text  (txt_utils.py:15-34, 64-96, 149-155)
# ------------------------------------------------------------------------------------------------
def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise the text around every `<image>` and join the pieces with `image_token_index`.
    If the tokenizer prepends BOS, it is kept once at the front and dropped from the later pieces."""
    pieces = [tokenizer(chunk).input_ids for chunk in prompt.split(DEFAULT_IMAGE_TOKEN)]
    has_bos = bool(pieces) and len(pieces[0]) > 0 and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids: List[int] = [pieces[0][0]] if has_bos else []
    for i, piece in enumerate(pieces):
        if i > 0:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def chat_template(source: Sequence[Dict[str, str]], tokenizer, roles_chat=("user", "assistant"), roles_data=("human", "gpt")) -> str:
    messages = []
    for i, turn in enumerate(source):
        assert turn["from"] == roles_data[i % 2]
        messages.append({"role": roles_chat[i % 2], "content": turn["value"]})
    text = tokenizer.apply_chat_template(messages, tokenize=False)
    if tokenizer.bos_token:
        text = text.replace(tokenizer.bos_token, "")
    return text


def preprocess_chat(source: Sequence[Dict[str, str]], tokenizer) -> str:
    """Gemma2 chat text for generation: user/model roles + the open model turn (txt_utils.py:85-96,149-155)."""
    return chat_template(source, tokenizer, roles_chat=("user", "model"), roles_data=("human", "gpt")) + "<start_of_turn>model\n"


def preprocess_chat_mistral(source: Sequence[Dict[str, str]], tokenizer) -> str:
    """Vidi-7B chat text: the tokenizer's Mistral [INST] template with user/assistant roles and no generation suffix
    (Vidi_7B/model/txt_utils.py:78-86, 135-140)."""
    return chat_template(source, tokenizer, roles_chat=("user", "assistant"), roles_data=("human", "gpt"))


# ------------------------------------------------------------------------------------------------
# images  (img_utils.py:173-198) — 'resize' is the released video configuration
# ------------------------------------------------------------------------------------------------
def _expand_to_square(img, fill):
    from PIL import Image
    w, h = img.size
    if w == h:
        return img
    side = max(w, h)
    canvas = Image.new(img.mode, (side, side), fill)
    canvas.paste(img, ((side - w) // 2, (side - h) // 2))
    return canvas


def process_images(images, image_processor, model_cfg):
    from PIL import Image
    mode = getattr(model_cfg, "mm_image_aspect_ratio", None)
    out = []
    if mode == "resize":
        size = image_processor.output_size
        for im in images:
            im = im.resize((size, size), resample=Image.BICUBIC)
            out.append(image_processor.preprocess(im, return_tensors="pt")["pixel_values"][0])
    elif mode == "pad":
        fill = tuple(int(x * 255) for x in image_processor.image_mean)
        for im in images:
            out.append(image_processor.preprocess(_expand_to_square(im, fill), return_tensors="pt")["pixel_values"][0])
    elif mode == "crop":
        return image_processor(images, return_tensors="pt")["pixel_values"]
    else:
        raise NotImplementedError(f"Unsupported image aspect ratio: {mode} (anyres is image-mode only, out of scope)")
    if all(x.shape == out[0].shape for x in out):
        return torch.stack(out, dim=0)
    return out


def process_images_gpu(images, image_processor, model_cfg, dtype=torch.bfloat16, device="cuda"):
    """`process_images` for the video configuration ('resize') with the per-pixel work on the GPU (vidi_amd/preproc.py):
    same arguments (PIL images / uint8 arrays of one size, the SigLIP image processor, the model config), same values —
    bit-exact with PIL's BICUBIC resize + `image_processor.preprocess` + `.to(dtype)` — returned on `device`."""
    if getattr(model_cfg, "mm_image_aspect_ratio", None) != "resize":
        raise NotImplementedError("GPU frame preprocessing covers mm_image_aspect_ratio == 'resize' (the video configuration)")
    from .preproc import FramePreprocessor
    pre = getattr(image_processor, "_vidi_gpu_pre", None)
    if pre is None or pre.dtype != dtype or str(pre.dev) != str(torch.device(device)):
        pre = FramePreprocessor.from_image_processor(image_processor, dtype=dtype, device=device)
        image_processor._vidi_gpu_pre = pre
    if isinstance(images, torch.Tensor):
        frames = images
    else:
        frames = np.stack([np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.uint8) for im in images])
    return pre(frames)


# ------------------------------------------------------------------------------------------------
# video / audio  (vid_utils.py:10-64) — decord + ffmpeg on the host, like the reference
# ------------------------------------------------------------------------------------------------
def load_video(file, fps: float = 1.0, time_range=None, num_threads: int = 0, clamp_last: bool = True):
    """Vidi1.5_9B/vidi/dataset/vid_utils.py:10-23: every `round(avg_fps / fps)`-th frame, or `round(seconds * fps)` frames spread evenly over
    `time_range` (seconds) with the last index clamped to the clip.  `clamp_last=False` is Vidi-7B's variant (Vidi_7B/model/vid_utils.py:7-19:
    no clamp — a range that ends past the clip makes decord raise, as there)."""
    from decord import VideoReader, cpu
    from PIL import Image
    vr = VideoReader(str(file), ctx=cpu(0), num_threads=num_threads)
    if time_range is None:
        step = round(vr.get_avg_fps() / fps)
        idx = list(range(0, len(vr), step))
    else:
        first = round(time_range[0] * vr.get_avg_fps())
        last = round(time_range[1] * vr.get_avg_fps())
        if clamp_last:
            last = min(last, len(vr) - 1)
        idx = np.linspace(first, last, round((time_range[1] - time_range[0]) * fps), dtype=int)
    frames = vr.get_batch(idx).asnumpy()
    return [Image.fromarray(f).convert("RGB") for f in frames]


def load_video_7b(file, fps: float = 1.0, time_range=None, num_threads: int = 0):
    """Vidi_7B/model/vid_utils.py:7-19 (what `model.vid_utils.load_video` resolves to through vidi_amd/compat_7b)"""
    return load_video(file, fps, time_range, num_threads, clamp_last=False)


def load_audio(file, sample_rate: int = 16000, time_range=None):
    from subprocess import run
    cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", str(file)]
    if time_range is not None:
        cmd += ["-ss", f"{time_range[0]:.2f}", "-t", f"{time_range[1] - time_range[0]:.2f}"]
    cmd += ["-f", "s16le", "-ac", "1", "-acodec", "pcm_s16le", "-ar", str(sample_rate), "-"]
    raw = run(cmd, capture_output=True, check=True).stdout
    return np.frombuffer(raw, np.int16).flatten().astype(np.float32) / 32768.0


def audio_num_frames(n_samples: int, window: int, hop: int) -> int:
    """mel frames the reference counts for an audio of n_samples: sum over 30-s windows of len(window)//hop
    (vid_utils.py:53-64 with transformers-4.50's `num_frames`; 5.x dropped `return_token_timestamps`)."""
    return sum(min(window, n_samples - s) // hop for s in range(0, n_samples, window))


def process_audio(audio: np.ndarray, audio_processor) -> Tuple[torch.Tensor, int]:
    n = audio_processor.n_samples
    pieces = [audio[i: i + n] for i in range(0, len(audio), n)]
    feats = audio_processor(pieces, sampling_rate=audio_processor.sampling_rate, return_tensors="pt")
    length = audio_num_frames(len(audio), n, audio_processor.hop_length)
    return feats.input_features, int(length)


def process_audio_gpu(audio: np.ndarray, audio_processor, dtype=torch.bfloat16, device="cuda") -> Tuple[torch.Tensor, int]:
    """`process_audio` with the log-mel computed on the GPU (vidi_amd/preproc.py:LogMelExtractor): same arguments and return
    values (`input_features` [C, n_mels, 3000] on `device` in `dtype`, `length` bit-exact)."""
    from .preproc import LogMelExtractor
    ext = getattr(audio_processor, "_vidi_gpu_ext", None)
    if ext is None or ext.dtype != dtype or str(ext.dev) != str(torch.device(device)):
        ext = LogMelExtractor.from_feature_extractor(audio_processor, dtype=dtype, device=device)
        audio_processor._vidi_gpu_ext = ext
    return ext(audio)


def get_media_length(file) -> float:
    """Vidi1.5_9B/vidi/dataset/vid_utils.py:67-80 (ffprobe, container duration in seconds)"""
    from subprocess import run
    cmd = ["ffprobe", "-i", str(file), "-show_entries", "format=duration", "-v", "quiet", "-of", "csv=p=0"]
    return float(run(cmd, capture_output=True, check=True).stdout.strip())


def build_processors(model_path: str, cfg):
    """tokenizer + SigLIP image processor + Whisper feature extractor, as `DattnMMModel.__init__` gathers them
    (multimodal.py:44-61, gemma.py:457-464): the tokenizer comes from the checkpoint directory, the two processors from the TOWER
    repositories (`mm_vision_tower` / `mm_audio_tower`).  Offline, a tower resolves to a local directory (weights.resolve_tower_dir);
    when it does not, the processor is built from the config with the towers' published preprocessing (SigLIP: bicubic resize to
    the tower's image size, 1/255 rescale, mean = std = 0.5; Whisper: 16 kHz, 25 ms window, hop 160, 30-s chunks)."""
    from transformers import AutoTokenizer, SiglipImageProcessor, WhisperFeatureExtractor
    from .weights import resolve_tower_dir
    import os
    tok = AutoTokenizer.from_pretrained(model_path, model_max_length=4096, padding_side="right")
    vdir, adir = resolve_tower_dir(cfg.mm_vision_tower, model_path), resolve_tower_dir(cfg.mm_audio_tower, model_path)
    if vdir and os.path.exists(os.path.join(vdir, "preprocessor_config.json")):
        img = SiglipImageProcessor.from_pretrained(vdir)
    else:
        S = cfg.vis_image_size
        img = SiglipImageProcessor(size={"height": S, "width": S}, resample=3, rescale_factor=1 / 255, image_mean=[0.5, 0.5, 0.5],
                                   image_std=[0.5, 0.5, 0.5])
    img.output_size = img.size["height"]
    if adir and os.path.exists(os.path.join(adir, "preprocessor_config.json")):
        aud = WhisperFeatureExtractor.from_pretrained(adir)
    else:
        aud = WhisperFeatureExtractor(feature_size=cfg.aud_num_mel_bins, sampling_rate=cfg.aud_sampling_rate, hop_length=cfg.aud_hop_length,
                                      chunk_length=cfg.aud_nb_max_frames * cfg.aud_hop_length // cfg.aud_sampling_rate, n_fft=400)
    return tok, img, aud
