"""Host processor API (names/signatures of the reference) — CPU only."""
import os
import sys

import numpy as np
import torch

from vidi_amd import processors as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeTok:
    bos_token_id = 2
    bos_token = "<bos>"

    def __call__(self, text):
        class R:
            pass
        r = R()
        r.input_ids = [2] + [10 + (ord(c) % 50) for c in text]
        return r

    def apply_chat_template(self, messages, tokenize=False):
        return "<bos>" + "".join(f"<start_of_turn>{m['role']}\n{m['content']}<end_of_turn>\n" for m in messages)


def test_tokenizer_image_token_splice():
    tok = FakeTok()
    ids = P.tokenizer_image_token("ab<image>cd", tok)
    a, c = tok("ab").input_ids, tok("cd").input_ids
    assert ids == [2] + a[1:] + [P.IMAGE_TOKEN_INDEX] + c[1:]
    t = P.tokenizer_image_token("<image>\nq", tok, return_tensors="pt")
    assert t.dtype == torch.long and int((t == -200).sum()) == 1 and int(t[0]) == 2


def test_preprocess_chat_gemma2():
    s = P.preprocess_chat([{"from": "human", "value": "<image>\nDuring which time segments can we see x?"}], FakeTok())
    assert s.startswith("<start_of_turn>user\n<image>\n") and s.endswith("<end_of_turn>\n<start_of_turn>model\n") and "<bos>" not in s


def test_audio_num_frames():
    # 24.6 s @16 kHz -> one window of 393600 samples -> 2460 mel frames (Na = 246)
    assert P.audio_num_frames(393600, 480000, 160) == 2460
    assert P.audio_num_frames(480000 * 2 + 1600, 480000, 160) == 3000 * 2 + 10


def test_process_images_resize():
    from PIL import Image

    class Proc:
        output_size = 16
        image_mean = [0.5, 0.5, 0.5]

        def preprocess(self, im, return_tensors="pt"):
            a = torch.from_numpy(np.asarray(im).astype(np.float32) / 255.0).permute(2, 0, 1)
            return {"pixel_values": [(a - 0.5) / 0.5]}

    class Cfg:
        mm_image_aspect_ratio = "resize"

    ims = [Image.fromarray((np.random.rand(20, 30, 3) * 255).astype(np.uint8)) for _ in range(3)]
    out = P.process_images(ims, Proc(), Cfg())
    assert out.shape == (3, 3, 16, 16)


def test_compat_import_paths():
    sys.path.insert(0, os.path.join(ROOT, "vidi_amd", "compat"))
    try:
        for k in [k for k in sys.modules if k == "vidi" or k.startswith("vidi.")]:
            del sys.modules[k]
        from vidi.constants import IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN
        from vidi.model.builder import load_pretrained_model
        from vidi.dataset.img_utils import process_images
        from vidi.dataset.txt_utils import tokenizer_image_token, preprocess_chat
        from vidi.dataset.vid_utils import load_video, load_audio, process_audio
        assert IMAGE_TOKEN_INDEX == -200 and DEFAULT_IMAGE_TOKEN == "<image>" and callable(load_pretrained_model)
    finally:
        sys.path.pop(0)
        for k in [k for k in sys.modules if k == "vidi" or k.startswith("vidi.")]:
            del sys.modules[k]


def test_compat_import_paths_7b():
    """Vidi_7B/inference.py:8-12 import paths resolve through vidi_amd/compat_7b."""
    sys.path.insert(0, os.path.join(ROOT, "vidi_amd", "compat_7b"))
    try:
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]
        from model.constants import IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN
        from model.builder import load_pretrained_model
        from model.img_utils import process_images
        from model.txt_utils import tokenizer_image_token, preprocess_chat
        from model.vid_utils import load_video, load_audio, process_audio
        assert IMAGE_TOKEN_INDEX == -200 and DEFAULT_IMAGE_TOKEN == "<image>" and callable(load_pretrained_model)

        class Tok:
            bos_token = "<s>"

            def apply_chat_template(self, messages, tokenize=False):
                return "<s>" + "".join(f"[INST] {m['content']} [/INST]" if m["role"] == "user" else m["content"] for m in messages)
        assert preprocess_chat([{"from": "human", "value": "hi"}], Tok()) == "[INST] hi [/INST]"
    finally:
        sys.path.pop(0)
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]


# ---- host logic vs values produced by EXECUTING the reference's functions (tests/golden/make_golden_host.py) --------
def _host_golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_host.json")) as f:
        return json.load(f)


def test_tokenizer_image_token_matches_reference_execution():
    for case in _host_golden()["tokenizer_image_token"]:
        assert P.tokenizer_image_token(case["prompt"], FakeTok(), -200) == case["ids"], case["prompt"]


def test_preprocess_chat_matches_reference_execution():
    for case in _host_golden()["preprocess_chat"]:
        assert P.preprocess_chat(case["source"], FakeTok()) == case["text"]


def test_ask_prompt_and_postprocessing_match_reference_execution():
    """eval/inference.py:18-66 run end to end with a stub model: the ids it hands to generate() and the HH:MM:SS string"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vidi_oracle as O
    for case in _host_golden()["ask"]:
        q = case["question"]
        qs = P.DEFAULT_IMAGE_TOKEN + "\n" + "During which time segments in the video can we see {}?".format(q[:-1] if q.endswith(".") else q)
        prompt = P.preprocess_chat([{"from": "human", "value": qs}], FakeTok())
        ids = P.tokenizer_image_token(prompt, FakeTok(), P.IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)
        assert ids.tolist() == case["input_ids"]
        assert O.format_time_ranges(case["answer"], case["length"]) == case["result"]
        # every keyword the reference passes to generate() is accepted by ours
        import inspect
        from vidi_amd.model import VidiForCausalLM
        sig = inspect.signature(VidiForCausalLM.generate)
        assert all(k in sig.parameters or any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values()) for k in case["generate_kwargs"])


def test_process_images_matches_reference_execution():
    """dataset/img_utils.py:process_images('resize') executed by the reference on two frames -> ours (host path) and the
    preprocessing oracle reproduce it bit for bit"""
    from PIL import Image
    from transformers import SiglipImageProcessor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import preproc_oracle as PO
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_process_images.npz"))
    proc = SiglipImageProcessor(size={"height": 98, "width": 98}, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    proc.output_size = 98
    cfgobj = type("C", (), {"mm_image_aspect_ratio": "resize"})()
    ours = P.process_images([Image.fromarray(f) for f in g["frames"]], proc, cfgobj).numpy()
    assert np.array_equal(ours, g["pixel_values"])
    orc = np.stack([PO.siglip_rescale_normalize(PO.pil_resize_bicubic_u8(f, 98, 98)) for f in g["frames"]])
    assert np.array_equal(orc, g["pixel_values"])
