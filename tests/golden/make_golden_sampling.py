"""Golden `do_sample=True` outputs produced by EXECUTING THE REFERENCE'S OWN `generate()` (gemma.py:603-655 / mistral.py:629-681 -> HF
`GenerationMixin._sample`: temperature / top-k / top-p warpers, one `torch.multinomial` draw per step) on CPU in fp32 under
`torch.manual_seed`, on the tiny models of make_golden_dattn*.py.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_sampling.py

writes tests/golden/reference_sampling.json; tests/test_generate_api.py::test_sampling_* replays the same seeds through the product class
over the CPU oracle engine: same draws, token for token — which pins the warpers' order and thresholds AND the defaults HF applies when a
knob is not passed (`top_k = 50`)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(HERE, "reference_sampling.json")

import make_golden_dattn as MG  # noqa: E402
import make_golden_dattn_7b as MG7  # noqa: E402

KW = [dict(), dict(temperature=0.7, top_k=20), dict(temperature=1.3, top_p=0.8), dict(top_k=5, top_p=0.9, temperature=0.9), dict(top_k=0, top_p=0.6),
      dict(temperature=2.0, top_k=None), dict(num_return_sequences=3, top_k=10),
      dict(num_beams=3, top_k=10), dict(num_beams=2, num_return_sequences=2, temperature=1.4, top_p=0.9)]       # beam-search multinomial sampling


def main():
    from vidi_amd.weights import init_random_weights
    out = []
    for arch, mod, npz, bos in (("vidi15", MG, "reference_dattn.npz", 2), ("vidi7b", MG7, "reference_dattn_7b.npz", 1)):
        cfg = mod.golden_config()
        model, _ = mod.build_reference_model(cfg)
        mod.load_weights(model, init_random_weights(cfg, seed=6, dtype=torch.float32, device="cpu"))
        model.generation_config.eos_token_id = cfg.eos_token_id
        d = np.load(os.path.join(HERE, npz))
        px, mel = torch.from_numpy(d["A_images"]), torch.from_numpy(d["A_audios"])
        prompt = [[bos, 21, 22, 23, -200, 24, 25, 26, 300, 301]]
        for i, kw in enumerate(KW):
            seed = 1000 + i
            torch.manual_seed(seed)
            with torch.no_grad():
                g = model.generate(torch.tensor(prompt), images=px, audios=mel, audio_sizes=[100], do_sample=True, max_new_tokens=10, use_cache=True,
                                   pad_token_id=0, **kw)
            out.append(dict(arch=arch, seed=6, torch_seed=seed, input_ids=prompt, kwargs=kw, tokens=g.tolist()))
            print(arch, kw, g.tolist())
    import transformers
    with open(OUT, "w") as f:
        json.dump(dict(transformers=transformers.__version__, torch=torch.__version__, cases=out), f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
