"""Golden vectors for the Vidi-7B specific module that is importable by file path: the learned Conv2DPool
(Vidi_7B/model/mm_vision/pool.py).  EXECUTES the reference module on seeded weights/inputs; run here (needs
/root/reference); the .npz is committed and checked by tests/test_oracle_golden.py.

    python tests/golden/make_golden_7b.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/Vidi_7B"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_modules_7b.npz")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    pool = load("ref7_pool", f"{REF}/model/mm_vision/pool.py")
    g = torch.Generator().manual_seed(4321)
    out = {}
    for tag, (d_in, d_out, s_in, s_out) in {"a": (3, 4, 27, 2), "b": (3, 4, 27, 3), "c": (5, 2, 7, 2), "d": (2, 2, 7, 7)}.items():
        torch.manual_seed(100 + len(tag) + s_out)
        m = pool.Conv2DPool(d_in, d_out, s_in, s_out).eval()
        x = torch.randn((2, d_in, s_in, s_in), generator=g)
        out[f"{tag}_cfg"] = np.array([d_in, d_out, s_in, s_out])
        out[f"{tag}_w"] = m.conv.weight.detach().numpy()
        out[f"{tag}_x"] = x.numpy()
        out[f"{tag}_y"] = m(x).detach().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
