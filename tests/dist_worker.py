"""Worker for tests/test_gpu_dist.py: runs a (possibly sharded) tiny prefill + one decode step on the GPU and
saves rank 0's hidden states.  Launched as a plain process (world 1) or under torch.distributed.run (world 2,
gloo transport, both ranks on cuda:0)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import torch.distributed as dist
    from util import seeded
    from vidi_amd.config import tiny
    from vidi_amd.model import VidiForCausalLM, strip_image_token
    from vidi_amd.weights import init_random_weights
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dt = torch.bfloat16
    cfg = tiny()
    backend = os.environ.get("VIDI_DIST_BACKEND", "gloo")
    grouped = world > 1 or os.environ.get("VIDI_FORCE_SHARDED", "0") == "1"       # one rank + forced shards: the RCCL branch on a one-GPU box
    if grouped:
        torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=rank, world_size=world)
    model = VidiForCausalLM(cfg, init_random_weights(cfg, seed=3, dtype=dt), dtype=dt, device="cuda:0")
    eng = model.engine
    if grouped:
        eng.set_dist(None)
    T, C, audio_size = 4, 2, 173
    px = seeded((T, 3, cfg.vis_image_size, cfg.vis_image_size), 200, 0.5).clamp(-1, 1).to(dt)
    mel = seeded((C, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 201, 0.3).to(dt)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24, 25]], dtype=torch.int64)
    # the PRODUCT entry point shards the video (every rank is handed all of it, vidi_amd/model.py:encode_mm_state)
    mm = model.encode_mm_state([px], [mel], [audio_size])
    # the inner seam (multimodal.py:254-265): under set_dist (either mode) the ranks' shards are all-gathered into the reference-order tensors
    g0 = getattr(eng, "n_token_gathers", 0)
    enc = [None if t is None else t.cpu() for t in model.encode_videos([px], [mel], [audio_size])]
    token_gathers = getattr(eng, "n_token_gathers", 0) - g0
    idt, mask, pos = strip_image_token(ids)
    ts = eng.new_text_state(1, 16)
    n0 = eng.n_collectives
    hn = eng.text_forward(eng.embed_tokens(idt.cuda()), pos.reshape(-1).cuda(), ts, mm, Lq=idt.shape[1], new_mask=mask.cuda())
    per_forward = eng.n_collectives - n0
    # what the per-layer exchange itself costs, BEFORE any amplification by the layers above: layer 0's merged T2V / T2A outputs for a
    # fixed query block (5 tokens), straight from the cross-attention entry points the text stream uses
    nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    Lq, G = 5, nq // nkv
    qx = seeded((Lq, nq * hd), 77).to(dt).cuda()
    xo = torch.zeros((2 * Lq, nq * hd), dtype=dt, device="cuda")
    if eng.sharded:
        eng._cross_sharded(qx, 0, mm, {"img": xo[:Lq], "aud": xo[Lq:]}, R=Lq * G)
    elif not eng._cross_dual(qx, 0, mm, xo[:Lq], xo[Lq:], R=Lq * G):
        eng._cross(qx, 0, mm, "img", xo[:Lq], R=Lq * G)
        eng._cross(qx, 0, mm, "aud", xo[Lq:], R=Lq * G)
    eng.n_collectives = n0 + per_forward
    nxt = torch.tensor([30], dtype=torch.int64).cuda()
    hn2 = eng.text_forward(eng.embed_tokens(nxt), torch.tensor([idt.shape[1]]).cuda(), ts, mm, Lq=1)
    toks = model.generate(ids, images=[px], audios=[mel], audio_sizes=[audio_size], max_new_tokens=6, do_sample=False)
    toks_cached = model.generate(ids, mm_state=mm, max_new_tokens=6, do_sample=False)
    # BASELINE configs[4] shape: a batch of 8 ragged prompts (right-padded, attention mask) against the one sharded video
    g = torch.Generator().manual_seed(5)
    lens = [5, 6, 7, 8, 9, 10, 11, 12]
    bids = torch.randint(20, cfg.vocab_size - 1, (8, max(lens)), generator=g)
    bids[:, 0] = 2
    bids[:, 3] = -200
    bmask = torch.arange(max(lens))[None, :] < torch.tensor(lens)[:, None]
    n1 = eng.n_collectives
    toks8 = model.generate(bids, mm_state=mm, attention_mask=bmask, max_new_tokens=4, do_sample=False)
    coll8 = eng.n_collectives - n1
    # sampled decoding under shards: every rank must leave the loop with the same tokens (rank 0's draw is broadcast)
    gs = torch.Generator(device="cuda:0").manual_seed(100 + rank)               # DIFFERENT RNG state per rank on purpose
    toks_s = model.generate(ids, mm_state=mm, max_new_tokens=5, do_sample=True, temperature=1.0, top_k=8, generator=gs)
    if grouped:
        allt = [None] * world
        dist.all_gather_object(allt, toks_s.cpu().tolist())
        assert all(t == allt[0] for t in allt), allt
    # graph-captured decode over shards (RCCL launches inside the captured step); the gloo transport is host-driven
    toks_graph = None
    if not grouped or backend == "nccl":
        os.environ["VIDI_DECODE_GRAPH"], os.environ["VIDI_DECODE_GRAPH_MIN"] = "1", "2"
        toks_graph = model.generate(ids, mm_state=mm, max_new_tokens=6, do_sample=False).cpu()
        os.environ["VIDI_DECODE_GRAPH"] = "0"
    if rank == 0:
        torch.save({"enc": enc, "token_gathers": token_gathers, "dist_mode": getattr(eng, "dist_mode", None), "xattn_layer0": xo.float().cpu(), "tokens8": toks8.cpu(), "collectives8": coll8, "tokens_graph": toks_graph, "sharded": bool(eng.sharded),
                    "prefill": hn.float().cpu(), "decode": hn2.float().cpu(), "g_img": int(mm.g_img), "g_aud": int(mm.g_aud),
                    "n_img_local": int(mm.n_img), "n_aud_local": int(mm.n_aud), "tokens": toks.cpu(), "tokens_cached": toks_cached.cpu(),
                    "collectives_per_forward": per_forward, "layers": cfg.num_hidden_layers}, out_path)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
