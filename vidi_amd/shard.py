"""Frame/time-axis sharding of one video over the ranks of a node.

The reference shards its sequence-parallel *training* path the same way (`vidi/model/lmm/dattn/sequence_parallel/split.py:73-93`
cuts the token axis into contiguous per-rank ranges); inference there is single-GPU.  Here every rank takes a contiguous range of
frames and of 30-s audio windows, encodes them with the GLOBAL positions (`frame_offset`/`total_frames`, `chunk_offset`/
`audio_size`), runs the diagonal multimodal stream on its own tokens and keeps its cross-attention K/V shard resident.

Pure host integer logic: importable without a GPU (tests/test_shard.py drives it under gloo)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple


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

