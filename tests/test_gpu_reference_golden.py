"""HIP path vs golden vectors produced by EXECUTING the reference's own model code on CPU/fp32
(tests/golden/make_golden_dattn.py, make_golden_dattn_7b.py — third-party stand-ins only).  No oracle in between:
`VidiForCausalLM.forward/generate` on the GPU against what `DattnGemma2ForCausalLM` / `DattnMistralForCausalLM` returned.
Tolerances: the GPU model computes in bf16/fp16 with the reference's rounding points, the goldens are fp32.  Activations: 3 % of
the tensor's rms + 2 % relative for bf16 (0.6 % / 0.4 % fp16).  Logits: max |err| <= LOGIT_TOL x std(logits) with no relative part —
5 % (bf16) / 1 % (fp16) (3.3 % typical, 4.7 % worst observed for bf16; audited per call through VIDI_TEST_REPORT, tests/util.py);
`test_a_two_percent_kernel_error_is_caught` plants a 2 % error in one projection and shows which of these checks see it.  Masks bit-exact.  Greedy tokens: |err| <= tol on every logit implies the same argmax wherever the
reference's top-2 margin exceeds 2 x tol, so free-running generate() must reproduce the reference's tokens up to the first step below
that margin (and at least 4 of them), and a teacher-forced decode (the reference's tokens fed back) must reproduce EVERY step's scores."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from util import logit_tol, report

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def tol(dt, k=1.0):
    """golden activations: 3 % of the spread + 2 % relative for bf16 (0.6 % + 0.4 % fp16) — the audit's worst use of round 4's 5 % + 3 % was 0.45"""
    return (3e-2 * k, 2e-2) if dt == torch.bfloat16 else (6e-3 * k, 4e-3)


def check_free_running(got, ref_tok, step_logits, atol, min_agree, what):
    """free-running greedy tokens against the reference's, up to the first step whose reference margin does not exceed 2 x atol"""
    agreed = 0
    for i in range(min(got.shape[1], ref_tok.shape[1])):
        top2 = torch.topk(step_logits[i].float(), 2).values
        if float(top2[0] - top2[1]) <= 2 * atol:
            break                                                                # a tie within tolerance: later tokens may diverge
        assert int(got[0, i]) == int(ref_tok[0, i]), f"{what}: token {i}: {int(got[0, i])} != reference {int(ref_tok[0, i])}"
        agreed += 1
    assert agreed >= min_agree, f"{what}: only {agreed} comparable greedy steps (need {min_agree})"
    return agreed


def teacher_forced_scores(model, ids, px, mel, sizes, ref_tok):
    """prefill + decode with the REFERENCE's tokens fed back: returns our logits of every step [n_steps, V]"""
    from vidi_amd.model import strip_image_token
    eng = model.engine
    mm = model.encode_mm_state(px, mel, sizes)
    idt, mask, pos = strip_image_token(ids)
    n = ref_tok.shape[1]
    ts, last = model._prefill(idt, mask, pos, mm, n + 1)
    out = [eng.logits_argmax(last)[0].float().cpu()[0]]
    for i in range(n - 1):
        emb = eng.embed_tokens(torch.tensor([int(ref_tok[0, i])], dtype=torch.int64).cuda())
        posn = ts.n_valid.clone(); ts.n_valid += 1
        hn = eng.text_forward(emb, posn, ts, mm, Lq=1)
        out.append(eng.logits_argmax(hn)[0].float().cpu()[0])
    return torch.stack(out)


def build(cfg, dt, seed=3):
    from vidi_amd.engine import VidiEngine
    from vidi_amd.model import VidiForCausalLM
    from vidi_amd.weights import init_random_weights
    w = init_random_weights(cfg, seed=seed, dtype=torch.float32, device="cpu")     # the weights the goldens were made with
    eng = VidiEngine(cfg, {k: v.to(dt) if not k.count(".mm_rand_pos_") else v for k, v in w.items()}, dtype=dt, device="cuda")
    model = VidiForCausalLM.__new__(VidiForCausalLM)
    model.config, model.dtype, model.device, model.engine = cfg, dt, torch.device("cuda"), eng
    model.generation_config = SimpleNamespace(eos_token_id=cfg.eos_token_id, pad_token_id=0)
    model.model = None
    return model


def check_case(model, D, case, dt, n_new, with_mask=False, min_agree=0):
    cfg = model.config
    t = lambda n: torch.from_numpy(D[f"{case}_{n}"])                                 # noqa: E731
    ids = t("input_ids")
    am = t("attention_mask") if with_mask else None
    px, mel = t("images").to(dt).cuda(), t("audios").to(dt).cuda()
    sizes = D[f"{case}_audio_sizes"].tolist()
    kw = {} if am is None else {"attention_mask": am}
    # the golden inputs are fp32; the GPU model sees them rounded to its dtype (as inference.py does with `.to(dtype)`)
    ref = t("prefill_logits")
    ltol = logit_tol(dt, ref)
    if with_mask:                                    # padded batch: each row's last VALID position, all positions checked too
        out = model.forward(ids, images=px, audios=mel, audio_sizes=sizes, logits_to_keep=0, **kw)
        tm = t("text_mask").bool()
        lens = tm.sum(-1)
        got = out.logits[torch.arange(ids.shape[0]), (lens - 1).cuda()]
        report(f"{case} all valid positions' logits", out.logits.cpu()[tm], t("prefill_logits_all")[tm], ltol, 0.0)
    else:
        got = model.forward(ids, images=px, audios=mel, audio_sizes=sizes, logits_to_keep=1, **kw).logits[:, -1]
    report(f"{case} prefill logits vs reference execution", got, ref, ltol, 0.0)
    if n_new:
        got = model.generate(ids, images=px, audios=mel, audio_sizes=sizes, max_new_tokens=n_new, do_sample=False).cpu()
        ref_tok = D[f"{case}_tokens"]
        logits = [ref[0]] + [x for x in torch.from_numpy(D[f"{case}_step_logits"])[0]]
        check_free_running(got, ref_tok, logits, ltol, min_agree, f"case {case}")
        # every decode step's logits with the reference's own tokens fed back (served from the three caches like gemma.py:646-687)
        tf = teacher_forced_scores(model, ids, px, mel, sizes, torch.from_numpy(ref_tok))
        report(f"{case} teacher-forced step logits vs reference execution", tf, torch.stack([x.float() for x in logits[: tf.shape[0]]]), ltol, 0.0)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_vidi15_against_reference_execution(dt):
    from vidi_amd.config import tiny
    D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
    model = build(tiny(sliding_window=64), dt)
    # encode pipeline: masks bit-exact, features within tolerance
    eng = model.engine
    px = torch.from_numpy(D["A_images"])[0].to(dt).cuda()
    feats, mask = eng.encode_video_images(px)
    assert torch.equal(mask.bool().cpu(), torch.from_numpy(D["A_image_mask"])[0])
    ref = torch.from_numpy(D["A_image_embeds"])[0]
    report("A image_embeds vs reference execution", feats, ref, *tol(dt, ref.std().item()))
    check_case(model, D, "A", dt, n_new=6, min_agree=5)          # step 0 margin 0.19 std, steps 1-5 0.44-0.76 std: all comparable
    check_case(model, D, "B", dt, n_new=0, with_mask=True)
    check_case(model, D, "C", dt, n_new=2)


@pytest.mark.parametrize("which", ["vidi15", "vidi7b"])
def test_exact_rounding_arms_against_reference_execution(which, monkeypatch):
    """The default engine re-rounds some weights once at load time (LayerNorm gains folded into the tower projections, the repeat_kv column
    blocks of the stream's o_proj summed) and gathers the patch / pool windows in the GEMM loader; each of those has a switch whose OFF arm
    keeps the reference's own rounding points and data path.  The goldens hold for that arm too (bf16)."""
    from vidi_amd.config import tiny, tiny_7b
    for env in ("VIDI_LN_FOLD", "VIDI_FOLD_REPKV", "VIDI_PATCH_LOADER", "VIDI_POOL_LOADER", "VIDI_STREAM_NORM2"):
        monkeypatch.setenv(env, "0")
    dt = torch.bfloat16
    if which == "vidi15":
        D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
        model = build(tiny(sliding_window=64), dt)
        assert not model.engine.ln_fold and not model.engine.fold_repkv and not model.engine.patch_loader
        check_case(model, D, "A", dt, n_new=6, min_agree=5)
        check_case(model, D, "B", dt, n_new=0, with_mask=True)
    else:
        D = np.load(os.path.join(GOLD, "reference_dattn_7b.npz"))
        model = build(tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64), dt)
        assert not model.engine.ln_fold and not model.engine.fold_repkv and not model.engine.pool_loader
        check_case(model, D, "A", dt, n_new=5)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_vidi7b_against_reference_execution(dt):
    from vidi_amd.config import tiny_7b
    D = np.load(os.path.join(GOLD, "reference_dattn_7b.npz"))
    model = build(tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64), dt)
    eng = model.engine
    mel = torch.from_numpy(D["A_audios"])[0].to(dt).cuda()
    feats, mask = eng.encode_video_audios(mel, int(D["A_audio_sizes"][0]))
    assert torch.equal(mask.bool().cpu(), torch.from_numpy(D["A_audio_mask"])[0])
    ref = torch.from_numpy(D["A_audio_embeds"])[0]
    report("7B audio_embeds vs reference execution", feats, ref, *tol(dt, ref.std().item()))
    check_case(model, D, "A", dt, n_new=5)


def test_vidi15_token_budget_branch_against_reference_execution():
    """case D of the goldens: 3 760 frames cross the token budget inside the reference's own encode_video_images ((10, 10) via
    resize_by_tokens, bilinear up-sampling in Conv2DPool, 25 tokens/frame).  HIP path: same token count (bit-exact integer rule),
    sampled embeddings and prefill logits within the bf16 tolerances."""
    from vidi_amd.config import tiny
    dt = torch.bfloat16
    D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
    cfg = tiny(sliding_window=64)
    model = build(cfg, dt)
    T = int(D["D_n_frames"][0])
    S, M, Fr = cfg.vis_image_size, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames
    gd = torch.Generator().manual_seed(777)
    px = (torch.randn((1, T, 3, S, S), generator=gd) * 0.5).clamp(-1, 1)
    mel = torch.randn((1, 1, M, Fr), generator=gd) * 0.3
    feats, mask = model.engine.encode_video_images(px[0].to(dt).cuda())
    assert feats.shape[0] == int(D["D_n_tokens"][0]) == 25 * T and int(mask.sum()) == int(D["D_mask_sum"][0])
    for name, sl in (("head", slice(0, 50)), ("mid", slice(47000, 47050)), ("tail", slice(-50, None))):
        ref = torch.from_numpy(D[f"D_embeds_{name}"])
        report(f"D embeds {name} vs reference execution", feats[sl], ref, *tol(dt, ref.std().item()))
    out = model.forward(torch.from_numpy(D["D_input_ids"]), images=px.to(dt).cuda(), audios=mel.to(dt).cuda(),
                        audio_sizes=D["D_audio_sizes"].tolist(), logits_to_keep=1)
    ref = torch.from_numpy(D["D_prefill_logits"])
    report("D prefill logits vs reference execution", out.logits[:, -1], ref, logit_tol(dt, ref), 0.0)


def test_vidi15_generate_against_reference_generate():
    """cases E/F: the reference's own generate() (HF greedy loop threaded by gemma.py:657-687).  Free-running, ours must return the same
    NEW tokens up to the first step whose top-2 margin is inside the tolerance (steps 0-4: margins 0.19-0.85 std; step 5, where the
    reference switches 301 -> 214, has margin 0.007 std) and stop at EOS like it does; teacher-forced, ALL 8 steps' scores must match,
    including the steps after the token change."""
    from vidi_amd.config import tiny
    dt = torch.bfloat16
    D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
    cfg = tiny(sliding_window=64)
    px = torch.from_numpy(D["A_images"]).to(dt).cuda(); mel = torch.from_numpy(D["A_audios"]).to(dt).cuda()
    m6 = build(cfg, dt, seed=6)
    ids = torch.from_numpy(D["E_input_ids"])
    got = m6.generate(ids, images=px, audios=mel, audio_sizes=[100], max_new_tokens=8, do_sample=False, use_cache=True, pad_token_id=0).cpu()
    scores = torch.from_numpy(D["E_scores"])[0]
    ltol = logit_tol(dt, scores)
    assert check_free_running(got, D["E_tokens"], scores, ltol, 5, "E generate()") >= 5
    tf = teacher_forced_scores(m6, ids, px, mel, [100], torch.from_numpy(D["E_tokens"]))
    report("E teacher-forced scores vs reference generate()", tf, scores, ltol, 0.0)
    assert len(set(D["E_tokens"][0].tolist())) > 1                                   # the golden sequence is not degenerate
    for i in range(scores.shape[0]):                                                 # same argmax wherever the margin allows: 7 of 8 steps
        top2 = torch.topk(scores[i], 2).values
        if float(top2[0] - top2[1]) > 2 * ltol:
            assert int(tf[i].argmax()) == int(D["E_tokens"][0, i])
    m3 = build(cfg, dt, seed=3)
    got = m3.generate(torch.from_numpy(D["F_input_ids"]), images=px, audios=mel, audio_sizes=[100], max_new_tokens=8, do_sample=False).cpu()
    assert got.tolist() == D["F_tokens"].tolist()                       # [[eos]]: one new token, then stop


def test_vidi7b_generate_against_reference_generate():
    """Vidi-7B case E: six DIFFERENT greedy tokens from the reference's own generate().  Step 1's margin (0.01 std) is inside any
    tolerance, so the free-running comparison ends there; the teacher-forced decode holds all six steps' scores to the tolerance and the
    argmax of the five steps whose margin allows it."""
    from vidi_amd.config import tiny_7b
    dt = torch.bfloat16
    D = np.load(os.path.join(GOLD, "reference_dattn_7b.npz"))
    cfg = tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64)
    m6 = build(cfg, dt, seed=6)
    px = torch.from_numpy(D["A_images"]).to(dt).cuda(); mel = torch.from_numpy(D["A_audios"]).to(dt).cuda()
    ids = torch.from_numpy(D["A_input_ids"])
    got = m6.generate(ids, images=px, audios=mel, audio_sizes=[100], max_new_tokens=6, do_sample=False).cpu()
    scores = torch.from_numpy(D["E_scores"])[0]
    ltol = logit_tol(dt, scores)
    check_free_running(got, D["E_tokens"], scores, ltol, 1, "7B E generate()")
    tf = teacher_forced_scores(m6, ids, px, mel, [100], torch.from_numpy(D["E_tokens"]))
    report("7B E teacher-forced scores vs reference generate()", tf, scores, ltol, 0.0)
    agreed = 0
    for i in range(scores.shape[0]):
        top2 = torch.topk(scores[i], 2).values
        if float(top2[0] - top2[1]) > 2 * ltol:
            assert int(tf[i].argmax()) == int(D["E_tokens"][0, i])
            agreed += 1
    assert agreed >= 4 and len(set(D["E_tokens"][0].tolist())) >= 4


def test_a_two_percent_kernel_error_is_caught():
    """Negative control for the bounds above (round-4 verdict: "the model-level tests would not catch a 3 % kernel error").  A kernel error is
    emulated where it would live — every call of one projection is off by 2 %: the output features of `o_proj` alternately x 1.02 and x 0.98 in
    every decoder layer (a UNIFORM factor would prove nothing: Gemma2's post-attention RMSNorm, gemma.py:237, divides it out exactly — on the
    CPU oracle a uniform x 1.02 moves the logits by 1e-6 of their spread, the alternating one by 11.5 %) — and case A is re-run on the damaged model:
      * the teacher-forced step logits / prefill logits against the reference execution must FAIL (the 5 % bound is meant to see this);
      * for the record, the damage in units of the bound is printed (and logged through VIDI_TEST_REPORT) for 1 %, 2 % and 4 %.
    What the model-level bound cannot see (a 1 % error) is the per-kernel tests' job: 1 % + 1 % per launch, teacher-forced per layer."""
    from vidi_amd.config import tiny
    import vidi_amd.weights as W
    dt = torch.bfloat16
    D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
    cfg = tiny(sliding_window=64)
    ref = torch.from_numpy(D["A_prefill_logits"])
    ltol = logit_tol(dt, ref)
    used = {}
    real_init = W.init_random_weights
    for gain in (1.0, 1.01, 1.02, 1.04):
        def damaged(*a, _g=gain, **kw):
            w = real_init(*a, **kw)
            for k in w:
                if k.endswith("self_attn.o_proj.weight"):
                    f = torch.ones((w[k].shape[0], 1), dtype=w[k].dtype)
                    f[::2] = _g
                    f[1::2] = 2.0 - _g
                    w[k] = w[k] * f
            return w
        W.init_random_weights = damaged
        try:
            model = build(cfg, dt)
        finally:
            W.init_random_weights = real_init
        ids = torch.from_numpy(D["A_input_ids"])
        px, mel = torch.from_numpy(D["A_images"]).to(dt).cuda(), torch.from_numpy(D["A_audios"]).to(dt).cuda()
        got = model.forward(ids, images=px, audios=mel, audio_sizes=D["A_audio_sizes"].tolist(), logits_to_keep=1).logits[:, -1].float().cpu()
        used[gain] = float((got - ref).abs().max()) / ltol
    print("share of the 5 % logit bound in use, by planted o_proj error:", {k: round(v, 2) for k, v in used.items()})
    log = os.environ.get("VIDI_TEST_REPORT")
    if log:
        import json
        with open(log, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "name": "negative control: planted o_proj error vs share of the logit bound", "used": used}) + "\n")
    assert used[1.0] < 1.0, "the undamaged model must pass"
    assert used[1.02] > 1.0, f"a 2 % error in every o_proj call stays inside the logit bound ({used[1.02]:.2f} of it): the bound is too loose to see it"
    assert used[1.04] > used[1.02] > used[1.0]
