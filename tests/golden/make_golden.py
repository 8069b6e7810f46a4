"""Generate golden vectors by EXECUTING the reference's own modules (the ones importable by file path
in the build container — SURVEY.md §8c) on seeded inputs.  Run here (needs /root/reference); the
.npz it writes is committed and is what tests/test_oracle_golden.py checks the oracle against.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/Vidi1.5_9B"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_modules.npz")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    sys.path.insert(0, REF)
    import vidi.utils as vutils                      # space_to_depth, resize_by_tokens (importable normally)
    norm = load("ref_norm", f"{REF}/vidi/model/mm_layer/norm.py")
    mlp = load("ref_mlp", f"{REF}/vidi/model/mm_layer/mlp.py")
    # pos.py does `from vidi.model.mm_layer import Linear`: pre-seed that package with the by-path module
    pkg_model = types.ModuleType("vidi.model"); pkg_model.__path__ = []
    sys.modules["vidi.model"] = pkg_model
    sys.modules["vidi.model.mm_layer"] = mlp
    pool = load("ref_pool", f"{REF}/vidi/model/mm_vision/pool.py")
    pos = load("ref_pos", f"{REF}/vidi/model/mm_vision/pos.py")

    g = torch.Generator().manual_seed(1234)
    out = {}
    # --- rms_norm / RMSNorm (fp32 and bf16 inputs)
    x = torch.randn((5, 64), generator=g) * 3
    w = torch.randn((64,), generator=g)
    out["norm_x"] = x.numpy(); out["norm_w"] = w.numpy()
    out["rms_norm_f32"] = norm.rms_norm(x).numpy()
    m = norm.RMSNorm(64, std=0.02898)
    out["RMSNorm_std_f32"] = m(x).detach().numpy()
    with torch.no_grad():
        m.weight.copy_(w)
    out["RMSNorm_w_f32"] = m(x).detach().numpy()
    out["RMSNorm_w_bf16"] = m.to(torch.bfloat16)(x.to(torch.bfloat16)).detach().float().numpy()
    # --- space_to_depth / Conv2DPool (no-resize sentinel and resize paths) / resize_by_tokens table
    f = torch.randn((2, 3, 27, 27), generator=g)
    out["pool_x"] = f.numpy()
    cp = pool.Conv2DPool(3, 3, 27, 2, 1, 2)
    out["pool_28"] = cp(f, (28, 28)).numpy()
    out["pool_10"] = cp(f, (10, 10)).numpy()
    out["pool_26"] = cp(f, (26, 26)).numpy()
    f7 = torch.randn((2, 3, 7, 7), generator=g)
    out["pool7_x"] = f7.numpy()
    out["pool7_28"] = cp(f7, (28, 28)).numpy()
    out["pool7_10"] = cp(f7, (10, 10)).numpy()
    out["s2d"] = vutils.space_to_depth(torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6), 2).numpy()
    Ts = [25, 300, 306, 307, 400, 600, 1200, 3600, 7200]
    tab = []
    for T in Ts:
        fake = torch.empty((T, 1, 27, 27))
        n_tokens = T * 28 * 28
        max_tokens = 60000 * 2 * 2
        hw = vutils.resize_by_tokens(fake, max_tokens) if n_tokens > max_tokens else (28, 28)     # multimodal.py:175-180
        tab.append([T, hw[0], hw[1]])
    out["budget_table"] = np.array(tab)
    # --- FractionalSinusoidalEmbedding / LearnablePosEmbd (eval mode, fp32 MLP)
    fs = pos.FractionalSinusoidalEmbedding(32)
    p = torch.arange(9, dtype=torch.float) / 8 * 99
    out["sin_p"] = p.numpy(); out["sin_pe"] = fs(p).numpy()
    torch.manual_seed(77)
    lp = pos.LearnablePosEmbd(32, 100).eval()
    sd = {k: v.detach().numpy() for k, v in lp.state_dict().items()}
    for k, v in sd.items():
        out["pos_" + k] = v
    xx = torch.zeros((7, 4, 5, 32), dtype=torch.bfloat16)
    out["pos_dim0_bf16"] = lp(xx, dim=0).detach().float().numpy()       # [7,1,1,32]
    out["pos_dim2_bf16"] = lp(xx, dim=2).detach().float().numpy()       # [1,1,5,32]
    # --- mm_layer.Linear fp32-cast forward + MLP('mlp2x_gelu')
    torch.manual_seed(78)
    mm = mlp.MLP("mlp2x_gelu", 16, 24).eval()
    xi = torch.randn((3, 16), generator=g)
    out["mlp_x"] = xi.numpy(); out["mlp_y"] = mm(xi).detach().numpy()
    for k, v in mm.state_dict().items():
        out["mlp_" + k] = v.detach().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
