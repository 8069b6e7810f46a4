"""Dry run of the multi-GPU partition (no GPU, no process group): for a world of N ranks on a BASELINE configuration, what every rank
encodes and keeps and what it exchanges — frames / 30-s windows, global token ranges, resident K/V bytes, the per-layer collective's
bytes at prefill and per decode step — computed with the product's own host logic (vidi_amd/shard.py, engine.token_budget_hw /
audio_token_counts), and checked against the figures DESIGN.md section 6 states.  The first real 8-GPU run (the driver's SCALE step) has
this table to be held against; `--json` prints it for a record under profiles/.

    python tools/dist_plan.py [--world 8] [--frames 3600] [--fps 1] [--prompt 39] [--queries 1] [--json]"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidi_amd import config as C  # noqa: E402
from vidi_amd.engine import audio_token_counts, token_budget_hw  # noqa: E402
from vidi_amd.shard import audio_shard_tokens, gather_counts, packed_partial_floats, video_shard  # noqa: E402


def plan(world: int, frames: int, fps: float, prompt: int, queries: int, preset: str = "vidi15_9b") -> dict:
    cfg = getattr(C, preset)()
    secs = frames / fps
    windows = math.ceil(secs / 30)
    audio_size = int(round(secs * 100))
    hw = token_budget_hw(frames, cfg.vis_side, cfg.mm_image_pool_size, cfg.mm_max_tokens_base)
    h, w = hw if hw[0] != 28 else (cfg.vis_side + 1, cfg.vis_side + 1)
    per_frame = (h // cfg.mm_image_pool_size) * (w // cfg.mm_image_pool_size)
    Nv = frames * per_frame
    enc_rows, Na = audio_token_counts(audio_size, cfg)
    L, nkv, hd, G = cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim, cfg.num_attention_heads // cfg.num_key_value_heads
    H = cfg.hidden_size
    kv_row = 2 * nkv * hd * 2                                     # K + V bytes of one key in one layer
    ranks, tv, ta = [], 0, 0
    for r in range(world):
        sh = video_shard(frames, windows, world, r)
        t0, n_a = audio_shard_tokens(sh.c0, sh.windows, cfg.aud_max_source_positions, cfg.mm_audio_pool_size, Na)
        n_v = sh.frames * per_frame
        assert sh.f0 * per_frame == tv and (n_a == 0 or t0 == ta), "token ranges must tile the sequence in rank order"
        ranks.append({"rank": r, "frames": [sh.f0, sh.f1], "windows": [sh.c0, sh.c1], "video_tokens": [tv, tv + n_v], "audio_tokens": [ta, ta + n_a],
                      "kv_resident_GB": (n_v + n_a) * kv_row * L / 1e9, "embeddings_never_gathered_MB": (n_v + n_a) * H * 2 / 1e6})
        tv += n_v; ta += n_a
    assert tv == Nv and ta == Na, (tv, Nv, ta, Na)
    rows_prefill = queries * prompt * G                             # (token, g) rows per kv head
    rows_decode = queries * G
    per_rank_prefill = packed_partial_floats(2, nkv, rows_prefill, hd) * 4
    per_rank_decode = packed_partial_floats(2, nkv, rows_decode, hd) * 4
    weights_GB = 18.5 if preset == "vidi15_9b" else None
    # dist mode "gather_tokens" (the north-star's literal collective, BASELINE configs[3]): every rank's rows in the token all-gather —
    # the product's own host rule (shard.gather_counts), which must agree with the per-rank ranges above
    g_img, g_aud = gather_counts(cfg, frames, windows, audio_size, world)
    assert g_img == [r["video_tokens"][1] - r["video_tokens"][0] for r in ranks] and g_aud == [r["audio_tokens"][1] - r["audio_tokens"][0] for r in ranks]
    row_bytes = H * 2 + 1                                         # one token: H values of the model dtype + its mask byte
    # what 8 GPUs can buy in each mode, from SURVEY 8(d)'s per-unit work (PFLOP): towers shard in both modes, the stream only when it stays sharded
    work = {"siglip": frames * (640.8e9 + 0.99e9), "whisper": windows * 2.25e12, "stream": (Nv + Na) * 367.0e6 * (L - 1) + (Nv + Na) * 29.36e6}
    amdahl_gather = sum(work.values()) / ((work["siglip"] + work["whisper"]) / world + work["stream"])
    amdahl_stream = float(world)
    return {"world": world, "preset": preset, "frames": frames, "fps": fps, "windows": windows, "audio_size": audio_size, "tokens_per_frame": per_frame,
            "video_tokens": Nv, "audio_tokens": Na, "ranks": ranks,
            "collectives_per_forward": L, "collective": "all_gather_into_tensor of the packed (numerator, m, l) partials of both modalities, fp32",
            "allgather_bytes_per_rank_prefill": per_rank_prefill, "allgather_bytes_total_prefill": per_rank_prefill * world,
            "allgather_bytes_per_rank_decode": per_rank_decode, "allgather_bytes_total_decode": per_rank_decode * world,
            "visual_token_embeddings_MB_not_exchanged": (Nv) * H * 2 / 1e6, "kv_total_GB": (Nv + Na) * kv_row * L / 1e9,
            "gather_tokens": {"collective": "all_gather_into_tensor of the visual / audio token rows + their mask bytes (4 collectives per video; ragged shards padded to the longest)",
                              "rows_per_rank_video": g_img, "rows_per_rank_audio": g_aud,
                              "allgather_MB_video": Nv * row_bytes / 1e6, "allgather_MB_audio": Na * row_bytes / 1e6,
                              "padded_MB_on_the_wire": (max(g_img) + max(g_aud)) * world * row_bytes / 1e6,
                              "ring_ms_at_153GBps_per_link": (max(g_img) + max(g_aud)) * (world - 1) * row_bytes / 153e9 * 1e3,
                              "kv_resident_GB_per_rank": (Nv + Na) * kv_row * L / 1e9,
                              "prefill_speedup_bound_amdahl": amdahl_gather},
            "sharded_stream": {"prefill_speedup_bound": amdahl_stream},
            "decode_step_bytes_per_rank_GB": None if weights_GB is None else weights_GB + max(x["kv_resident_GB"] for x in ranks),
            "decode_step_floor_ms_at_6.29TBps": None if weights_GB is None else (weights_GB + max(x["kv_resident_GB"] for x in ranks)) / 6.29}


def check_design_figures() -> None:
    """DESIGN.md section 6 / SURVEY 8(e) for the 60-min single-query configuration at 8 ranks"""
    p = plan(8, 3600, 1.0, 39, 1)
    assert p["video_tokens"] == 90000 and p["audio_tokens"] == 36000
    assert [r["frames"][1] - r["frames"][0] for r in p["ranks"]] == [450] * 8 and [r["windows"][1] - r["windows"][0] for r in p["ranks"]] == [15] * 8
    assert abs(p["visual_token_embeddings_MB_not_exchanged"] - 645.1) < 0.1                 # "645 MB never gathered"
    assert abs(p["allgather_bytes_per_rank_prefill"] / 1e6 - 1.29) < 0.01                   # "1.29 MB per rank at the 39-token prefill"
    assert abs(p["allgather_bytes_per_rank_decode"] / 1e3 - 33.0) < 0.1                     # "33 KB per decode step"
    assert abs(p["kv_total_GB"] - 43.35) < 0.01 and abs(max(r["kv_resident_GB"] for r in p["ranks"]) - 5.42) < 0.01    # "43 GB -> 5.4 GB per GPU"
    assert p["collectives_per_forward"] == 42
    g = p["gather_tokens"]                                                                  # the north-star's collective, BASELINE configs[3]
    assert g["rows_per_rank_video"] == [11250] * 8 and g["rows_per_rank_audio"] == [4500] * 8
    assert abs(g["allgather_MB_video"] - 645.2) < 0.1 and abs(g["allgather_MB_audio"] - 258.1) < 0.1     # "645 MB at 60 min"
    assert 1.9 < g["prefill_speedup_bound_amdahl"] < 2.2 and abs(g["kv_resident_GB_per_rank"] - 43.35) < 0.01     # "only ~2.0x"
    assert abs(p["decode_step_floor_ms_at_6.29TBps"] - 3.8) < 0.05                          # "the step's floor is ~3.8 ms"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frames", type=int, default=3600)
    ap.add_argument("--fps", type=float, default=1.0)
    ap.add_argument("--prompt", type=int, default=39)
    ap.add_argument("--queries", type=int, default=1)
    ap.add_argument("--preset", default="vidi15_9b")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    check_design_figures()
    p = plan(a.world, a.frames, a.fps, a.prompt, a.queries, a.preset)
    if a.json:
        print(json.dumps(p))
    else:
        print(f"world {p['world']}: {p['frames']} frames @{p['fps']:g} fps -> {p['video_tokens']} video + {p['audio_tokens']} audio tokens ({p['tokens_per_frame']} per frame, {p['windows']} windows)")
        for r in p["ranks"]:
            print(f"  rank {r['rank']}: frames {r['frames']}  windows {r['windows']}  video tokens {r['video_tokens']}  audio tokens {r['audio_tokens']}  "
                  f"K/V resident {r['kv_resident_GB']:.2f} GB  embeddings kept local {r['embeddings_never_gathered_MB']:.1f} MB")
        print(f"  per layer: one all-gather, {p['allgather_bytes_per_rank_prefill'] / 1e6:.3f} MB per rank at prefill ({p['allgather_bytes_total_prefill'] / 1e6:.2f} MB gathered), "
              f"{p['allgather_bytes_per_rank_decode'] / 1e3:.1f} KB per rank per decode step; {p['collectives_per_forward']} per forward")
        g = p["gather_tokens"]
        print(f"  --dist-mode gather_tokens instead: all-gather of {g['allgather_MB_video']:.0f} MB video + {g['allgather_MB_audio']:.0f} MB audio tokens once per video "
              f"(ring at 153 GB/s per link: {g['ring_ms_at_153GBps_per_link']:.1f} ms), decoder replicated: K/V {g['kv_resident_GB_per_rank']:.1f} GB on EVERY rank, "
              f"prefill speed-up bound {g['prefill_speedup_bound_amdahl']:.2f}x (sharded stream: {p['sharded_stream']['prefill_speedup_bound']:.0f}x)")
        print(f"  never exchanged: {p['visual_token_embeddings_MB_not_exchanged']:.0f} MB of visual-token embeddings; K/V {p['kv_total_GB']:.2f} GB total")
        if p["decode_step_floor_ms_at_6.29TBps"]:
            print(f"  decode step floor: {p['decode_step_bytes_per_rank_GB']:.1f} GB per rank -> {p['decode_step_floor_ms_at_6.29TBps']:.2f} ms at 6.29 TB/s")
        print("DESIGN.md section 6 figures: ok")
