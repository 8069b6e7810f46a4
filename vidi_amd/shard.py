"""Frame/time-axis sharding of one video over the ranks of a node.

The reference shards its sequence-parallel *training* path the same way (`vidi/model/lmm/dattn/sequence_parallel/split.py:73-93`
cuts the token axis into contiguous per-rank ranges); inference there is single-GPU.  Here every rank takes a contiguous range of
frames and of 30-s audio windows, encodes them with the GLOBAL positions (`frame_offset`/`total_frames`, `chunk_offset`/
`audio_size`), runs the diagonal multimodal stream on its own tokens and keeps its cross-attention K/V shard resident.

Pure host integer logic: importable without a GPU (tests/test_shard.py drives it under gloo)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple


def token_budget_hw(T: int, side: int, pool: int, base: int) -> Tuple[int, int]:
    """Token-budget rule (multimodal.py:175-180 + vidi/utils.py:152-171), integer/float host math.
    Returns the `hw` the reference hands to Conv2DPool; (28,28) is its "no resize" sentinel."""
    n_tokens = T * (side + 1) * (side + 1)
    max_tokens = base * pool * pool
    if n_tokens > max_tokens:
        H = W = side + 1
        ratio = math.sqrt(max_tokens / (T * H * W))
        th, tw = int(H * ratio), int(W * ratio)
        return max(10, th - th % 2), max(10, tw - tw % 2)
    return 28, 28


def audio_token_counts(audio_size: int, cfg) -> Tuple[int, int]:
    """floor(size*1500/3000), then floor(/pool) — multimodal.py:226-227, 234-235 (same numpy float64 ops)."""
    import numpy as np
    pool_ratio = cfg.aud_max_source_positions / cfg.aud_nb_max_frames
    s1 = int(np.floor(np.array([audio_size]) * pool_ratio).astype(int)[0])
    s2 = int(np.floor(np.array([s1]) / cfg.mm_audio_pool_size).astype(int)[0])
    return s1, s2


def shard(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, end) of rank's share of n units; sizes differ by at most one, earlier ranks take the extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(max(0, n), world)
    s = rank * base + min(rank, extra)
    return s, s + base + (1 if rank < extra else 0)


@dataclass(frozen=True)
class VideoShard:
    """What one rank encodes of a T-frame video with C audio windows."""
    f0: int
    f1: int
    c0: int
    c1: int
    total_frames: int
    total_windows: int

    @property
    def frames(self) -> int:
        return self.f1 - self.f0

    @property
    def windows(self) -> int:
        return self.c1 - self.c0


def video_shard(total_frames: int, total_windows: int, world: int, rank: int) -> VideoShard:
    f0, f1 = shard(total_frames, world, rank)
    c0, c1 = shard(total_windows, world, rank)
    return VideoShard(f0, f1, c0, c1, total_frames, total_windows)


def audio_shard_tokens(chunk_offset: int, windows_local: int, rows_per_window: int, pool: int, tokens_total: int) -> Tuple[int, int]:
    """(first global audio token, token count) of a rank that holds `windows_local` 30-s windows starting at window `chunk_offset`.
    A window gives rows_per_window encoder rows = rows_per_window // pool pooled tokens (the Conv1d has kernel == stride == pool,
    multimodal.py:231-233, so pooling never straddles a window when pool divides rows_per_window); `tokens_total` is the GLOBAL
    count floor(floor(audio_size * 1500 / 3000) / pool) (multimodal.py:226-235) and clips the last windows."""
    if rows_per_window % pool:
        raise ValueError("encoder rows per window must be a multiple of the audio pool size to shard by window")
    per = rows_per_window // pool
    tok0 = chunk_offset * per
    return tok0, max(0, min(tokens_total - tok0, windows_local * per))


def image_tokens_per_frame(cfg, budget_frames: int) -> int:
    """video tokens one frame contributes: (h / pool) * (w / pool) under the token-budget rule (Vidi1.5, `budget_frames` = the frames the
    rule counts: the whole batch's, multimodal.py:157-158, 175-180), pool * pool for Vidi-7B's learned Conv2DPool."""
    pool = cfg.mm_image_pool_size
    if cfg.arch == "mistral":
        return pool * pool
    hw = token_budget_hw(budget_frames, cfg.vis_side, pool, cfg.mm_max_tokens_base)
    h, w = hw if hw[0] != 28 else (cfg.vis_side + 1, cfg.vis_side + 1)
    return (h // pool) * (w // pool)


def gather_counts(cfg, total_frames: int, total_windows: int, audio_size: Optional[int], world: int,
                  budget_frames: Optional[int] = None) -> Tuple[List[int], List[int]]:
    """Rows every rank contributes to the all-gather of visual / audio tokens (`dist_mode = "gather_tokens"`), from host integers alone —
    no rank has to ask another how much it encoded: -> (video tokens per rank, audio tokens per rank), both in rank (= reference) order."""
    per = image_tokens_per_frame(cfg, total_frames if budget_frames is None else budget_frames) if total_frames else 0
    img = [(shard(total_frames, world, r)[1] - shard(total_frames, world, r)[0]) * per for r in range(world)]
    aud = [0] * world
    if total_windows and audio_size is not None:
        s2_total = audio_token_counts(int(audio_size), cfg)[1]
        rows = cfg.aud_max_source_positions                     # encoder rows per 30-s window
        for r in range(world):
            c0, c1 = shard(total_windows, world, r)
            aud[r] = audio_shard_tokens(c0, c1 - c0, rows, cfg.mm_audio_pool_size, s2_total)[1]
    return img, aud


def packed_partial_floats(n_modalities: int, nkv: int, rows: int, hd: int) -> int:
    """fp32 words one rank contributes to the per-layer all-gather: per modality a numerator [nkv, rows, hd] and (m, l) [nkv, rows, 2]."""
    return n_modalities * nkv * rows * (hd + 2)


def packed_offsets(slot: int, nkv: int, rows: int, hd: int) -> Tuple[int, int]:
    """(numerator offset, (m, l) offset) in fp32 words of modality slot `slot` inside one rank's packed partial."""
    base = slot * nkv * rows * (hd + 2)
    return base, base + nkv * rows * hd


def split_key_slices(slices: int, subtiles_a: int, subtiles_b: int, min_subtiles: int = 8):
    """Share `slices` split-KV key slices of ONE cross-attention launch between two modalities in proportion to their 32-key sub-tiles
    (the launch is as long as its slowest slice): -> (za, zb), each >= 1, za + zb <= slices, and no slice shorter than `min_subtiles`
    sub-tiles unless the modality itself is shorter (a slice costs a partial and a merge step whatever its length)."""
    if slices < 2 or subtiles_a <= 0 or subtiles_b <= 0:
        raise ValueError("two non-empty modalities and at least two slices")
    za = min(max(1, round(slices * subtiles_a / (subtiles_a + subtiles_b))), slices - 1)
    zb = slices - za
    cap = lambda z, n: max(1, min(z, (n + min_subtiles - 1) // min_subtiles))
    return cap(za, subtiles_a), cap(zb, subtiles_b)

