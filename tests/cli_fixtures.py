"""Fixtures for the end-to-end CLI tests: a real (character-level) HF tokenizer written to disk, synthetic media, a checkpoint dir."""
import os

import numpy as np


def write_tokenizer(path, vocab_size=512, mistral=False):
    """PreTrainedTokenizerFast with BOS prepended (like Gemma / Mistral), a chat template, and a vocabulary in which every id the tiny
    model can emit decodes to text: printable ASCII characters one by one, and every other id to one complete '0.ab-0.cd, ' range."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors, decoders
    from transformers import PreTrainedTokenizerFast
    vocab = {"<pad>": 0, "<unk>": 1, "<bos>": 2, "<eos>": 7}
    nxt = 10
    for c in [chr(i) for i in range(32, 127)] + ["\n"]:
        vocab[c] = nxt
        nxt += 1
    for i in range(vocab_size):
        if i in vocab.values():
            continue
        a, b = (i * 7) % 100, (i * 13 + 5) % 100
        lo, hi = min(a, b), max(a, b)
        vocab[f"0.{lo:02d}-0.{hi:02d}, " + "​" * (i % 5) + f"‌{i}‌"] = i          # unique strings; the suffix has no digits adjacent to the range
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Split(pattern="", behavior="isolated")
    tk.post_processor = processors.TemplateProcessing(single="<bos> $A", special_tokens=[("<bos>", 2)])
    tk.decoder = decoders.Fuse()
    tmpl = ("{{ bos_token }}{% for m in messages %}[INST] {{ m['content'] }} [/INST]{% endfor %}" if mistral else
            "{{ bos_token }}{% for m in messages %}<start_of_turn>{{ m['role'] }}\n{{ m['content'] }}<end_of_turn>\n{% endfor %}")
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<bos>", eos_token="<eos>", pad_token="<pad>", unk_token="<unk>", chat_template=tmpl)
    tok.save_pretrained(path)
    return tok


def media(seed, n_frames=3, seconds=1.6):
    from PIL import Image
    rng = np.random.default_rng(seed)
    frames = [Image.fromarray(rng.integers(0, 256, size=(60, 80, 3), dtype=np.uint8)) for _ in range(n_frames)]
    audio = (rng.standard_normal(int(16000 * seconds)) * 0.1).astype(np.float32)
    return frames, audio
