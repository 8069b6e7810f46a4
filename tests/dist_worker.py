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
    if world > 1:
        dist.init_process_group("gloo")
    model = VidiForCausalLM(cfg, init_random_weights(cfg, seed=3, dtype=dt), dtype=dt, device="cuda:0")
    eng = model.engine
    if world > 1:
        eng.set_dist(None)
    T, C, audio_size = 4, 2, 173
    px = seeded((T, 3, cfg.vis_image_size, cfg.vis_image_size), 200, 0.5).clamp(-1, 1).to(dt)
    mel = seeded((C, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 201, 0.3).to(dt)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24, 25]], dtype=torch.int64)
    # the PRODUCT entry point shards the video (every rank is handed all of it, vidi_amd/model.py:encode_mm_state)
    mm = model.encode_mm_state([px], [mel], [audio_size])
    idt, mask, pos = strip_image_token(ids)
    ts = eng.new_text_state(1, 16)
    n0 = eng.n_collectives
    hn = eng.text_forward(eng.embed_tokens(idt.cuda()), pos.reshape(-1).cuda(), ts, mm, Lq=idt.shape[1], new_mask=mask.cuda())
    per_forward = eng.n_collectives - n0
    nxt = torch.tensor([30], dtype=torch.int64).cuda()
    hn2 = eng.text_forward(eng.embed_tokens(nxt), torch.tensor([idt.shape[1]]).cuda(), ts, mm, Lq=1)
    toks = model.generate(ids, images=[px], audios=[mel], audio_sizes=[audio_size], max_new_tokens=6, do_sample=False)
    toks_cached = model.generate(ids, mm_state=mm, max_new_tokens=6, do_sample=False)
    if rank == 0:
        torch.save({"prefill": hn.float().cpu(), "decode": hn2.float().cpu(), "g_img": int(mm.g_img), "g_aud": int(mm.g_aud),
                    "n_img_local": int(mm.n_img), "n_aud_local": int(mm.n_aud), "tokens": toks.cpu(), "tokens_cached": toks_cached.cpu(),
                    "collectives_per_forward": per_forward, "layers": cfg.num_hidden_layers}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
