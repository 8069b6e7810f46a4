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
    from vidi_amd.engine import VidiEngine
    from vidi_amd.model import strip_image_token
    from vidi_amd.weights import init_random_weights
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dt = torch.bfloat16
    cfg = tiny()
    if world > 1:
        dist.init_process_group("gloo")
    eng = VidiEngine(cfg, init_random_weights(cfg, seed=3, dtype=dt), dtype=dt, device="cuda:0")
    if world > 1:
        eng.set_dist(None)
    T, C, audio_size = 4, 2, 173
    px = seeded((T, 3, cfg.vis_image_size, cfg.vis_image_size), 200, 0.5).clamp(-1, 1).to(dt)
    mel = seeded((C, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 201, 0.3).to(dt)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24, 25]], dtype=torch.int64)
    f0, f1 = rank * T // world, (rank + 1) * T // world
    c0, c1 = rank * C // world, (rank + 1) * C // world
    nz = eng.normalizer
    fi, mi = eng.encode_video_images(px[f0:f1].cuda(), frame_offset=f0, total_frames=T, normalizer=nz)
    fa, ma = eng.encode_video_audios(mel[c0:c1].cuda(), audio_size, normalizer=nz, chunk_offset=c0)
    mm = eng.mm_stream_prefill(fi, mi, fa, ma, pre_normalized=True)
    idt, mask, pos = strip_image_token(ids)
    ts = eng.new_text_state(1, 16)
    hn = eng.text_forward(eng.embed_tokens(idt.cuda()), pos.reshape(-1).cuda(), ts, mm, Lq=idt.shape[1], new_mask=mask.cuda())
    nxt = torch.tensor([30], dtype=torch.int64).cuda()
    hn2 = eng.text_forward(eng.embed_tokens(nxt), torch.tensor([idt.shape[1]]).cuda(), ts, mm, Lq=1)
    if rank == 0:
        torch.save({"prefill": hn.float().cpu(), "decode": hn2.float().cpu(), "g_img": int(mm.g_img), "g_aud": int(mm.g_aud),
                    "n_img_local": int(mm.n_img), "n_aud_local": int(mm.n_aud)}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
