"""Same-box A/B of the decode step (Vidi1.5-9B decoder, 60-min caches: 90 000 image + 36 000 audio keys, random data).

  python tools/ab_decode.py gemv lib_a.so lib_b.so ...   the decoder's five skinny projections on every library (same ABI, loaded side by
                                                       side with ctypes), weights cycled through > 256 MB so every call streams HBM
  python tools/ab_decode.py cross lib_a.so lib_b.so ...  the T2V + T2A launch of one layer on every library
  python tools/ab_decode.py one [tokens] [rounds]        the same loop with the engine's default switches only (for rocprofv3)
  python tools/ab_decode.py e2e [tokens] [rounds]        one engine (the library VIDI_HIP_LIB names, default the product), ms per greedy
                                                       token with the decode switches flipped between timed runs; greedy tokens compared
"""
import ctypes
import dataclasses
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gemv_ab(libs):
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [("qkv", "gemv", 1, 8192, 3584, 6), ("o_proj x3", "gemv", 3, 3584, 4096, 10), ("gate/up", "glu", 1, 14336, 3584, 3),
              ("down", "gemv", 1, 3584, 14336, 4), ("lm_head", "gemv", 1, 256000, 3584, 1)]
    st = lambda: torch.cuda.current_stream().cuda_stream
    for name, kind, M, N, K, copies in shapes:
        rows = 2 * N if kind == "glu" else N
        ws = [(torch.randn((rows, K), generator=g, device="cuda") * 0.02).to(dt) for _ in range(copies)]
        x = torch.randn((M, K), generator=g, device="cuda").to(dt)
        res, outs = {}, {}
        for path in libs:
            lib = ctypes.CDLL(path)
            if kind == "glu":
                f = lib.vidi_gemv_glu
                f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
            else:
                f = lib.vidi_gemv
                f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
            f.restype = ctypes.c_int
            y = torch.zeros((M, N), dtype=dt, device="cuda")

            def run(w, f=f, y=y):
                if kind == "glu":
                    rc = f(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, 1, 0, st())
                else:
                    rc = f(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, 0, st())
                assert rc == 0, rc
            run(ws[0]); torch.cuda.synchronize()
            outs[path] = y.clone()
            res[path] = [run, 0.0]
        rounds, reps = 5, 4
        for _ in range(rounds):
            for path in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    for w in ws:
                        res[path][0](w)
                e1.record(); torch.cuda.synchronize()
                res[path][1] += e0.elapsed_time(e1) / (reps * len(ws))
        for path in libs:
            us = res[path][1] / rounds * 1e3
            print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "lib": os.path.basename(path), "us": round(us, 2),
                              "TB/s": round(rows * K * 2 / us / 1e6, 3),
                              "bit_identical_to_first": bool(torch.equal(outs[path], outs[libs[0]]))}), flush=True)
        del ws


def cross_ab(libs):
    """vidi_attn_cross2 at the decode shape (1 token, 16 heads over 8 kv heads, 90 000 + 36 000 keys) and at the prompt shape (39 tokens),
    three layers' caches cycled (3 GB); the merged outputs of every library compared with the first one's"""
    dt = torch.bfloat16
    nkv, G, HD, Nv, Na = 8, 2, 256, 90000, 36000
    aud_start = (Nv + 63) // 64 * 64
    ntile = (aud_start + Na + 63) // 64
    g = torch.Generator(device="cuda").manual_seed(1)
    caches = [((torch.randn((nkv, ntile, 64, HD), generator=g, device="cuda")).to(dt),
               (torch.randn((nkv, 2 * ntile, HD, 32), generator=g, device="cuda")).to(dt)) for _ in range(3)]
    from vidi_amd import hip
    st = lambda: torch.cuda.current_stream().cuda_stream
    for Lq in (1, 39):
        R = Lq * G
        Rpad = (R + 31) // 32 * 32
        Z = 256 // (nkv * (Rpad // 32))
        za = min(max(1, round(Z * ((Nv + 31) // 32) / ((Nv + 31) // 32 + (Na + 31) // 32))), Z - 1)
        zb = Z - za
        q = torch.randn((Lq, nkv * G * HD), generator=g, device="cuda").to(dt)
        wa = hip.attn_cross_workspace(za, nkv, Rpad, HD, "cuda"); wb = hip.attn_cross_workspace(zb, nkv, Rpad, HD, "cuda")
        runs, outs = {}, {}
        for path in libs:
            lib = ctypes.CDLL(path)
            f = lib.vidi_attn_cross2
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p] * 3 + ([ctypes.c_void_p] * 3 + [ctypes.c_int] * 3) * 2 + [ctypes.c_int] * 7 + \
                         [ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]

            def run(kc, vtc, f=f):
                rc = f(q.data_ptr(), kc.data_ptr(), vtc.data_ptr(), None, wa[0].data_ptr(), wa[1].data_ptr(), 0, Nv, za,
                       None, wb[0].data_ptr(), wb[1].data_ptr(), aud_start, Na, zb, R, Rpad, G, nkv, HD, q.stride(0), ntile,
                       HD ** -0.5, 50.0, 0, st())
                assert rc == 0, rc
            run(*caches[0])
            oa = torch.zeros((Lq, nkv * G * HD), dtype=dt, device="cuda"); ob = torch.zeros_like(oa)
            hip.attn_merge2(wa[0], wa[1], oa, za, False, wb[0], wb[1], ob, zb, False, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
            torch.cuda.synchronize()
            outs[path] = torch.cat([oa, ob]).float()
            runs[path] = [run, 0.0]
        rounds, reps = 5, 4
        names = list(runs)
        for _ in range(rounds):
            for path in names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    for kc, vtc in caches:
                        runs[path][0](kc, vtc)
                e1.record(); torch.cuda.synchronize()
                runs[path][1] += e0.elapsed_time(e1) / (reps * len(caches))
        nbytes = (Nv + Na) * 2 * nkv * HD * 2
        for path in names:
            us = runs[path][1] / rounds * 1e3
            print(json.dumps({"shape": f"cross2 Lq={Lq}", "za": za, "zb": zb, "lib": os.path.basename(path), "us": round(us, 1),
                              "TB/s": round(nbytes / us / 1e6, 3),
                              "max_abs_diff_vs_first": float((outs[path] - outs[libs[0]]).abs().max()),
                              "out_std": float(outs[path].std())}), flush=True)


def e2e(tokens, rounds, only_default=False):
    from vidi_amd import config as C
    from vidi_amd.engine import VidiEngine
    from vidi_amd.weights import init_random_weights
    dt = torch.bfloat16
    cfg = dataclasses.replace(C.vidi15_9b(), vis_num_layers=1, aud_num_layers=1)
    eng = VidiEngine(cfg, init_random_weights(cfg, seed=3, dtype=dt, device="cuda"), dtype=dt, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    img = (torch.randn((90000, cfg.hidden_size), generator=g, device="cuda") * cfg.mm_std * eng.normalizer).to(dt)
    aud = (torch.randn((36000, cfg.hidden_size), generator=g, device="cuda") * cfg.mm_std * eng.normalizer).to(dt)
    mm = eng.mm_stream_prefill(img, None, aud, None, pre_normalized=True, check_masks=False)
    mm.img_any_valid = mm.aud_any_valid = True
    del img, aud
    L = 39
    ids = torch.randint(0, cfg.vocab_size, (1, L), generator=g, device="cuda")

    def run(n):
        ts = eng.new_text_state(1, L + n + 2)
        hn = eng.text_forward(eng.embed_tokens(ids), torch.arange(L, device="cuda"), ts, mm, Lq=L)
        ts.n_valid = torch.tensor([L], device="cuda")
        _, nxt = eng.logits_argmax(hn.view(1, L, -1)[:, -1])
        torch.cuda.synchronize()
        toks = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            emb = eng.embed_tokens(nxt)
            posn = ts.n_valid.clone(); ts.n_valid += 1
            hn = eng.text_forward(emb, posn, ts, mm, Lq=1)
            _, nxt = eng.logits_argmax(hn)
            toks.append(int(nxt[0]))
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, toks

    arms = [("two-launch T2T, cross per modality, norm pairs as launches", False, False, False, False), ("fused T2T", True, False, False, False),
            ("dual cross", False, True, False, False), ("norm pairs inside the projections", False, False, True, False),
            ("all three", True, True, True, False), ("all three + T2T and merge as one launch", True, True, True, True)]
    if only_default:                                     # the engine's own switches (env), e.g. under rocprofv3
        arms = [("engine defaults", eng.decode_attn, eng.cross_dual, eng.decode_norm_gemv, eng.decode_tail)]
    tot = {a[0]: 0.0 for a in arms}
    toks = {}
    for r in range(rounds + 1):
        for name, da, cd, ng, dtl in arms:
            eng.decode_attn, eng.cross_dual, eng.decode_norm_gemv, eng.decode_tail = da, cd, ng, dtl
            ms, t = run(tokens)
            if r == 0:
                toks[name] = t
            else:
                tot[name] += ms
    for name, *_ in arms:
        print(json.dumps({"lib": os.path.basename(os.environ.get("VIDI_HIP_LIB", "libvidi_hip.so")), "arm": name,
                          "ms_per_token": round(tot[name] / rounds, 3), "same_greedy_tokens_as_first_arm": toks[name] == toks[arms[0][0]]}),
              flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gemv":
        gemv_ab([a for a in sys.argv[2:] if a.endswith(".so")])
    elif len(sys.argv) > 1 and sys.argv[1] == "cross":
        cross_ab([a for a in sys.argv[2:] if a.endswith(".so")])
    else:
        a = [x for x in sys.argv[2:]]
        e2e(int(a[0]) if a else 24, int(a[1]) if len(a) > 1 else 2, only_default=sys.argv[1] == "one")
