"""Golden values for the host-side integer/string logic, produced by EXECUTING the reference's own functions
(Vidi1.5_9B/vidi/dataset/txt_utils.py `tokenizer_image_token`, `preprocess_chat`; eval/inference.py `ask()` driven end to end
with a stub model/tokenizer so that its prompt construction and timestamp post-processing run unmodified).  Build container
only (needs /root/reference):   python tests/golden/make_golden_host.py  ->  tests/golden/reference_host.json
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import ref_harness as RH  # noqa: E402
from test_processors import FakeTok  # noqa: E402  (deterministic character-level tokenizer with a Gemma-style chat template)


def main():
    RH.install()
    import vidi.dataset.txt_utils as TU
    import vidi.eval.inference as INF
    tok = FakeTok()
    out = {"tokenizer_image_token": [], "preprocess_chat": [], "ask": []}
    for prompt in ["ab<image>cd", "<image>\nq", "no image here", "<image>", "x<image>y<image>z", ""]:
        out["tokenizer_image_token"].append({"prompt": prompt, "ids": TU.tokenizer_image_token(prompt, tok, -200)})
    for src in ([{"from": "human", "value": "<image>\nDuring which time segments in the video can we see a dog?"}],
                [{"from": "human", "value": "hi"}, {"from": "gpt", "value": "hello"}, {"from": "human", "value": "<image>\nand now?"}]):
        out["preprocess_chat"].append({"source": src, "text": TU.preprocess_chat(src, tok)})

    # ---- ask(): stub everything around the two pieces of reference logic (prompt building, post-processing) ----
    captured = {}

    class Model:
        config = None

        def generate(self, input_ids, **kw):
            captured["input_ids"] = input_ids.cpu().tolist()
            captured["kwargs"] = sorted(k for k in kw)
            return torch.tensor([[1, 2, 3]])

    class Tok(FakeTok):
        pad_token_id = 0

        def batch_decode(self, ids, skip_special_tokens=True):
            return [captured["answer"]]

    class AP:
        sampling_rate = 16000

    INF.os.path.exists = lambda p: True
    INF.load_video = lambda p: [None]
    INF.process_images = lambda v, ip, cfg: torch.zeros(1, 3, 4, 4)
    INF.load_audio = lambda p, sr: None
    INF.process_audio = lambda a, ap: (torch.zeros(1, 8, 10), 10)
    torch.Tensor.cuda = lambda self, *a, **k: self
    for question, answer, length in [("a dog running.", "0.10-0.25, 0.50-0.75", 3600.0), ("x", "garbage", 10.0),
                                     ("y", "0.999-1.000", 3661.5), ("z", " 0.00-0.01,0.333-0.667 and 0.5-0.50001 ", 86399.9),
                                     ("w", "1.5-2.5", 100.0), ("v", "0.12345-0.54321", 24.6)]:
        INF.get_length = lambda p, _l=length: _l
        captured["answer"] = answer
        res = INF.ask(question, "video.mp4", Model(), Tok(), None, AP())
        out["ask"].append({"question": question, "answer": answer, "length": length, "result": res,
                           "input_ids": captured["input_ids"], "generate_kwargs": captured["kwargs"]})
    # ---- process_images (dataset/img_utils.py:173-198, 'resize' mode) executed with a real SiglipImageProcessor at a small size
    import numpy as np
    from PIL import Image
    from transformers import SiglipImageProcessor
    import vidi.dataset.img_utils as IU
    proc = SiglipImageProcessor(size={"height": 98, "width": 98}, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    proc.output_size = 98
    rng = np.random.default_rng(42)
    frames = [rng.integers(0, 256, size=(120, 213, 3), dtype=np.uint8), rng.integers(0, 256, size=(120, 213, 3), dtype=np.uint8)]
    frames[1][:40, :100] = 255
    cfgobj = type("C", (), {"mm_image_aspect_ratio": "resize"})()
    pv = IU.process_images([Image.fromarray(f) for f in frames], proc, cfgobj)
    np.savez_compressed(os.path.join(HERE, "reference_process_images.npz"), frames=np.stack(frames), pixel_values=pv.numpy())
    with open(os.path.join(HERE, "reference_host.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote reference_host.json:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
