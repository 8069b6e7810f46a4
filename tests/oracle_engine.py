"""Test infrastructure: an object with VidiEngine's interface whose numerics are the CPU oracle (oracle/vidi_oracle.py).

`vidi_amd.model.VidiForCausalLM(..., engine=OracleEngine(...))` lets the host logic of the product class — `generate()`'s greedy
loop, EOS / padding handling, batching, `forward`, `encode_videos`, `prepare_inputs_labels_for_multimodal` — run in a container
without a GPU, e.g. under the reference's own CLI (`tests/test_reference_cli.py`).  It is never used by the product: the product
engine is `vidi_amd.engine.VidiEngine` (HIP kernels only, no CPU path)."""
import dataclasses
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F

import vidi_oracle as O


def oracle_config(cfg) -> "O.OracleConfig":
    names = {f.name for f in dataclasses.fields(O.OracleConfig)}
    d = {k: v for k, v in cfg.to_dict().items() if k in names}
    d["arch"] = cfg.arch                                             # a property of VidiConfig (from model_type), a field of OracleConfig
    return O.OracleConfig(**d, vis_select_layer=cfg.mm_vision_select_layer)


class OracleEngine:
    """embed_tokens / encode_* return UN-normalised rows; `O.model_forward` applies the sqrt(H) normalizer itself (gemma.py:353-356).
    model.py treats these tensors as opaque, so the division of labour between the methods is the engine's business."""

    def __init__(self, cfg, weights, dtype=torch.float32):
        self.cfg, self.ocfg = cfg, oracle_config(cfg)
        self.w = {k: v.float() for k, v in weights.items()}
        self.dtype, self.dev = torch.float32, torch.device("cpu")
        self.mistral = cfg.arch == "mistral"
        self.normalizer = 1.0
        self.world, self.rank, self.pg = 1, 0, None
        self.dist_mode, self.shard_encode = "sharded_stream", False
        # per_unit: frames / audio windows are encoded ONE AT A TIME (every matmul sees the same operand shapes whatever the shard cut), so
        # that "the all-gathered shards equal the single-rank encode" can be asserted BIT FOR BIT on the CPU as well (a batched fp32 GEMM
        # may round a row differently when the row count changes: a property of the BLAS blocking, not of the path under test)
        self.per_unit = False

    # ---- multi-rank test mode (tests/test_shard.py): the product class (vidi_amd/model.py) shards the video; this engine checks
    #      that the shards tile it with the right global offsets, reassembles them over the process group (the north-star's literal
    #      "all-gather of visual tokens") and runs the oracle on the whole — so every rank must reproduce the single-rank answer ----
    def set_dist(self, group=None, mode=None):
        import os
        import torch.distributed as dist
        self.pg, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        self.dist_mode = mode or os.environ.get("VIDI_DIST_MODE", "sharded_stream")
        self.shard_encode = self.world > 1

    def sample_flag(self, x):
        return torch.tensor([int(bool((x != 0).any()))], dtype=torch.int32)

    def _gather_shards(self, rec):
        import torch.distributed as dist
        allr = [None] * self.world
        dist.all_gather_object(allr, rec, group=self.pg)
        return allr

    # ---- multimodal encode ----
    def _local(self):
        return self.per_unit or (self.world > 1 and self.dist_mode == "gather_tokens")

    def _images_local(self, pixel, frame_offset, total_frames, flag, budget_frames):
        """the rows of frames [frame_offset, frame_offset + T) of a `total_frames`-frame video, frame by frame, with the GLOBAL
        positions / budget (multimodal.py:156-208 restated through the oracle's pieces for ONE frame at a time)"""
        w, cfg, m = self.w, self.ocfg, "model."
        T = int(pixel.shape[0])
        Ttot = T if total_frames is None else int(total_frames)
        d, side, pool = cfg.hidden_size, cfg.vis_side, cfg.mm_image_pool_size
        hw = None if cfg.arch == "mistral" else O.token_budget_hw(Ttot if budget_frames is None else int(budget_frames), side, pool, cfg.mm_max_tokens_base)
        pt_all = O.mm_rms_norm(O.learnable_pos_embd(Ttot, cfg.mm_time_interval, d, w, m + "mm_rand_pos_t.", torch.float32))
        rows = []
        for t in range(T):
            f = O.siglip_forward(pixel[t: t + 1].float(), w, cfg)
            f = f.reshape(1, side, side, -1).permute(0, 3, 1, 2)
            f = O.learned_conv2d_pool(f, w[m + "mm_rand_img_pool.conv.weight"], pool) if cfg.arch == "mistral" else O.conv2d_pool(f, hw, pool)
            f = f.permute(0, 2, 3, 1)
            f = O.mm_RMSNorm(O.projector_mlp(f, w, m + "mm_rand_img_projector."), w[m + "mm_rand_img_norm.weight"])
            ph = O.learnable_pos_embd(f.shape[1], pool, d, w, m + "mm_rand_pos_h.", f.dtype)
            f = f + O.mm_rms_norm(ph.reshape(1, -1, 1, d))
            pw = O.learnable_pos_embd(f.shape[2], pool, d, w, m + "mm_rand_pos_w.", f.dtype)
            f = f + O.mm_rms_norm(pw.reshape(1, 1, -1, d))
            f = f + pt_all[frame_offset + t].reshape(1, 1, 1, d)
            rows.append(f.flatten(0, 2))
        if not rows:
            return torch.empty((0, d)), torch.empty((0,), dtype=torch.uint8)
        feats = torch.cat(rows, dim=0)
        mask = (torch.sum(torch.abs(feats), dim=-1) != 0) & bool(flag)
        feats = O.mm_RMSNorm(feats, w[m + "mm_rand_llm_norm.weight"]) * mask.unsqueeze(-1)
        return feats, mask.to(torch.uint8)

    def _audios_local(self, mel, audio_size, chunk_offset, flag):
        """the audio tokens of 30-s windows [chunk_offset, chunk_offset + C), window by window, clipped by the GLOBAL floors
        (multimodal.py:210-252 for one window at a time; the Conv1d has kernel == stride == pool, so it never straddles windows)"""
        import torch.nn.functional as F
        from vidi_amd.shard import audio_shard_tokens
        w, cfg, m = self.w, self.ocfg, "model."
        d, pool = cfg.hidden_size, cfg.mm_audio_pool_size
        s1, s2 = O.audio_token_counts([int(audio_size)], cfg)
        s1, s2_total = int(s1[0]), int(s2[0])
        assert s2_total > 1
        pt_all = O.mm_rms_norm(O.learnable_pos_embd(s2_total, cfg.mm_time_interval, d, w, m + "mm_rand_pos_t.", torch.float32))
        rows_per = cfg.aud_max_source_positions
        rows = []
        for c in range(int(mel.shape[0])):
            tok0, n = audio_shard_tokens(chunk_offset + c, 1, rows_per, pool, s2_total)
            if n <= 0:
                continue
            f = O.whisper_encoder_forward(mel[c: c + 1].float(), w, cfg)[0]               # [rows_per, Da]
            f = F.conv1d(f[: n * pool].t()[None], w[m + "mm_rand_aud_pool.weight"], None, stride=pool)[0].t()
            f = O.mm_RMSNorm(O.projector_mlp(f, w, m + "mm_rand_aud_projector."), w[m + "mm_rand_aud_norm.weight"])
            rows.append(f + pt_all[tok0: tok0 + n])
        if not rows:
            return torch.empty((0, d)), torch.empty((0,), dtype=torch.uint8)
        feats = torch.cat(rows, dim=0)
        mask = (torch.sum(torch.abs(feats), dim=-1) != 0) & bool(flag)
        feats = O.mm_RMSNorm(feats, w[m + "mm_rand_llm_norm.weight"]) * mask.unsqueeze(-1)
        return feats, mask.to(torch.uint8)

    def encode_video_images(self, pixel, normalizer=None, frame_offset=0, total_frames=None, sample_flag=None, **kw):
        if self._local():
            flag = int(sample_flag[0]) if sample_flag is not None else int(bool((pixel != 0).any()))
            self.last_shard = dict(kind="img", local=int(pixel.shape[0]), off=int(frame_offset), total=int(pixel.shape[0] if total_frames is None else total_frames))
            return self._images_local(pixel.float().cpu(), int(frame_offset), total_frames, flag, kw.get("budget_frames"))
        if self.world > 1:
            allr = self._gather_shards(dict(x=pixel.float().cpu(), off=int(frame_offset), tot=total_frames, flag=int(sample_flag[0])))
            full = torch.cat([r["x"] for r in allr], dim=0)
            off = 0
            for r in allr:                                            # contiguous, ordered, complete; every rank knows the global T
                assert r["off"] == off and r["tot"] == full.shape[0], (r["off"], off, r["tot"], full.shape[0])
                assert r["flag"] == int(bool((full != 0).any()))      # the "sample is not all zeros" flag is the WHOLE sample's
                off += r["x"].shape[0]
            f, m = O.encode_video_images([full], self.w, self.ocfg)
            self.last_shard = dict(kind="img", local=int(pixel.shape[0]), off=int(frame_offset), total=int(full.shape[0]))
            return f[0], m[0].to(torch.uint8)
        f, m = O.encode_video_images([pixel.float().cpu()], self.w, self.ocfg, budget_frames=kw.get("budget_frames"))
        return f[0], m[0].to(torch.uint8)

    def encode_video_audios(self, mel, audio_size, normalizer=None, chunk_offset=0, sample_flag=None, **kw):
        if self._local():
            flag = int(sample_flag[0]) if sample_flag is not None else int(bool((mel != 0).any()))
            return self._audios_local(mel.float().cpu(), int(audio_size), int(chunk_offset), flag)
        if self.world > 1:
            allr = self._gather_shards(dict(x=mel.float().cpu(), off=int(chunk_offset), size=int(audio_size), flag=int(sample_flag[0])))
            full = torch.cat([r["x"] for r in allr], dim=0)
            off = 0
            for r in allr:
                assert r["off"] == off and r["size"] == int(audio_size)   # audio_size stays the GLOBAL mel-frame count
                assert r["flag"] == int(bool((full != 0).any()))
                off += r["x"].shape[0]
            f, m = O.encode_video_audios([full], [int(audio_size)], self.w, self.ocfg)
            return f[0], m[0].to(torch.uint8)
        f, m = O.encode_video_audios([mel.float().cpu()], [int(audio_size)], self.w, self.ocfg)
        return f[0], m[0].to(torch.uint8)

    def mm_stream_prefill(self, img, img_mask, aud, aud_mask, pre_normalized=True, check_masks=True):
        # the oracle runs the multimodal stream inside its first model_forward call (interleaved, like the reference)
        return SimpleNamespace(img=img, imask=None if img_mask is None else img_mask.bool(), aud=aud,
                               amask=None if aud_mask is None else aud_mask.bool(),
                               g_img=0 if img is None else img.shape[0], g_aud=0 if aud is None else aud.shape[0])

    # ---- text stream ----
    def new_text_state(self, B, Lmax):
        return SimpleNamespace(B=B, Lmax=Lmax, caches=O.OracleCaches(), mask=torch.zeros((B, 0), dtype=torch.bool), past_len=0, n_valid=None)

    def reorder_text_state(self, ts, parents):
        idx = parents.long().cpu()
        c = ts.caches
        for name in ("text", "image", "audio"):
            setattr(c, name, [(k.index_select(0, idx), v.index_select(0, idx)) for k, v in getattr(c, name)])
        ts.mask = ts.mask.index_select(0, idx)
        if ts.n_valid is not None:
            ts.n_valid = ts.n_valid.index_select(0, idx)

    def embed_tokens(self, ids, normalize=True):
        ids = ids.reshape(-1).long().cpu()
        e = F.embedding(ids.clamp(min=0), self.w["model.embed_tokens.weight"])
        return e * (ids >= 0)[:, None]

    def text_forward(self, hidden, positions, ts, mm, Lq, new_mask=None, dyn=False):
        B = ts.B
        emb = hidden.view(B, Lq, -1)
        pos = positions.view(B, Lq).long().cpu()
        nm = torch.ones((B, Lq), dtype=torch.bool) if new_mask is None else new_mask.bool().cpu()
        ts.mask = torch.cat([ts.mask, nm], dim=1)
        ex = lambda t: None if t is None else t[None].expand(B, *t.shape)          # noqa: E731  (queries of a batch share the video)
        img, imask, aud, amask = (None,) * 4 if mm is None else (ex(mm.img), ex(mm.imask), ex(mm.aud), ex(mm.amask))
        h = O.model_forward(emb, pos, ts.mask, img, imask, aud, amask, self.w, self.ocfg, ts.caches, ts.past_len)
        ts.past_len += Lq
        return h.reshape(B * Lq, -1)

    def logits_argmax(self, hn_last):
        logits = O.lm_logits(hn_last[:, None, :], self.w, self.ocfg)[:, 0]
        return logits, torch.argmax(logits.float(), dim=-1)
