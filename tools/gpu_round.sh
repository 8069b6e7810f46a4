#!/bin/bash
# One GPU-box session: GPU parity suite (with the tolerance-slack audit), smoke, bench, rocprofv3 kernel stats, PMC traffic passes.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tests|bench|prof|pmc|smoke|decode|variants|mfma ...]   (default: tests smoke bench prof pmc)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
what=${@:-tests smoke bench prof pmc}
for w in $what; do
case $w in
tests)
  rm -f $OUT/tol_audit.jsonl $OUT/reference_cli.jsonl
  # VIDI_CLI_RECORD: tests/test_gpu_cli.py writes the answer strings of the reference's own inference.py (staged by __graft_entry__.build()
  # under oracle/_ref/) and of vidi_amd/inference.py, both driving the HIP engine
  VIDI_CLI_RECORD=$OUT/reference_cli.jsonl VIDI_TEST_REPORT=$OUT/tol_audit.jsonl timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 --durations=15 > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; tail -5 $OUT/pytest.log ;;
canary)
  # the WHOLE GPU suite once under the guard-zone device allocator (tests/canary/): every tensor its own hipMalloc with poisoned zones
  VIDI_CANARY=1 timeout 3000 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider --deselect tests/test_gpu_canary.py::test_kernel_and_model_suites_under_the_guard_zone_allocator > $OUT/canary.log 2>&1
  echo "canary rc=$?"; tail -5 $OUT/canary.log ;;
xattn)
  # many-row cross-attention micro-benchmark: the single prompt (39 tokens) and the 8-prompt batch (304 tokens), masked (the product's form whenever a
  # key is invalid) and unmasked, bf16 fixed reference / fp16 + uncapped running reference, against the per-tile kernel (VIDI_XATTN_ROWS=0)
  : > $OUT/ab_xattn.jsonl
  for lq in 39 304; do for cfg in "bf16 50" "fp16 50" "bf16 0" "bf16 80"; do set -- $cfg; for m in 0 1; do for rows in 1 0; do
    VIDI_XATTN_ROWS=$rows PYTHONPATH=. timeout 300 python tools/bench_xattn.py --keys 90000 --lq $lq --zsplit 0 --dtype $1 --softcap $2 --masked $m --iters 20 | sed "s/^{/{\"rows_kernel\": $rows, /" >> $OUT/ab_xattn.jsonl
  done; done; done; done
  echo "xattn rc=$?"; python - <<'PY'
import json
for l in open("gpurun_out/ab_xattn.jsonl"):
    d = json.loads(l)
    print("Lq", d["Lq"], d["dtype"], "cap", d["softcap"], "masked", int(d["masked"]), "rows_kernel", d["rows_kernel"], "tiles/block", d["row_tiles_per_block"], "zsplit", d["zsplit"], "ms %.3f" % d["ms"], "TFLOP/s %.0f" % d["TFLOPs"], "GB/s %.0f" % d["GBps"])
PY
  ;;
probe)
  # what the GPU box has of the media decoders BASELINE configs[0] needs (recorded in DESIGN.md section 9)
  { echo "ffmpeg: $(which ffmpeg 2>&1 || echo absent)"; echo "ffprobe: $(which ffprobe 2>&1 || echo absent)"; python -c "import decord; print('decord', decord.__version__)" 2>&1 | tail -1;
    python -c "import cv2; print('cv2', cv2.__version__)" 2>&1 | tail -1; python -c "import av; print('av', av.__version__)" 2>&1 | tail -1; rocm-smi --showproductname 2>/dev/null | grep -i "card series" | head -1; nproc; } > $OUT/probe_media.txt 2>&1
  cat $OUT/probe_media.txt ;;
dist8g)
  # EIGHT ranks sharing the one GPU (gloo transport): BASELINE's 8-way partition through bench.py end to end in both dist modes; 48 frames (6 per rank; 2 audio windows: six ranks without audio) — every rank of the gather mode holds the FULL K/V, eight of them share 288 GB here
  for m in gather_tokens sharded_stream; do
    VIDI_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --dist-mode $m --frames 48 --steps 1 --warmup 1 --no-preproc --no-other-configs --no-kernel-timer --decode-steps 8 > $OUT/bench_dist8_$m.json 2> $OUT/bench_dist8_$m.err; echo "dist8 $m rc=$?"
  done
  timeout 600 python bench.py --frames 48 --steps 1 --warmup 1 --no-preproc --no-cpu-baseline --no-other-configs --no-kernel-timer --decode-steps 8 > $OUT/bench_dist8_ref1.json 2> $OUT/bench_dist8_ref1.err; echo "dist8 ref rc=$?"
  # the gather mode's bit-identity with one rank holds when every GEMM of a rank's shard takes the SAME kernel as on the whole video: vidi_gemm picks the
  # 128 x 128 tile kernel below 192 tiles of 256 x 256 (another MFMA shape sums in another order) — below 14 frames per rank for the N = 1 152 tower
  # GEMMs, below 3 511 token rows per rank for the projector (18 frames at 196 tokens / frame; BASELINE: 450 frames, 11 250 rows per rank).  World 4 x 24 frames:
  VIDI_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 4 --dist-mode gather_tokens --frames 96 --steps 1 --warmup 1 --no-preproc --no-other-configs --no-kernel-timer --decode-steps 4 > $OUT/bench_dist4_gather_tokens_96.json 2> $OUT/bench_dist4_gather_tokens_96.err; echo "dist4 gather 96 rc=$?"
  timeout 600 python bench.py --frames 96 --steps 1 --warmup 1 --no-preproc --no-cpu-baseline --no-other-configs --no-kernel-timer --decode-steps 4 > $OUT/bench_dist4_ref1_96.json 2> $OUT/bench_dist4_ref1_96.err; echo "dist4 ref 96 rc=$?"
  python - <<'PY'
import json
for n in ("dist8_gather_tokens", "dist8_sharded_stream", "dist8_ref1", "dist4_gather_tokens_96", "dist4_ref1_96"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/bench_{n}.json") if l.startswith("{")][-1])
        v = d.get("verify") or {}
        print(n, "n_gpus", d["n_gpus"], "frames", d["config"]["frames"], "value", round(d["value"]), "dist_mode", d.get("dist_mode"), {k: round(x, 1) for k, x in d["stage_ms_per_step"].items()}, "first_token", d["first_token"], "sha", d.get("first_token_logits_sha256", "")[:12], "verify", v.get("ok"), v.get("kv_rows"), v.get("frames_checked"))
    except Exception as e:
        print(n, "failed:", e)
PY
  ;;
dist2g)
  # dist mode gather_tokens (BASELINE configs[3] as worded) next to the sharded stream, two ranks on the one GPU (gloo transport), 10-minute video
  for m in gather_tokens sharded_stream; do
    VIDI_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --dist-mode $m --frames 600 --steps 1 --warmup 1 --no-preproc --no-other-configs > $OUT/bench_dist2_$m.json 2> $OUT/bench_dist2_$m.err; echo "dist2 $m rc=$?"
  done
  timeout 600 python bench.py --frames 600 --steps 1 --warmup 1 --no-preproc --no-cpu-baseline --no-other-configs > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "dist1 rc=$?"
  python - <<'PY'
import json
for n in ("dist2_gather_tokens", "dist2_sharded_stream", "dist1"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/bench_{n}.json") if l.startswith("{")][-1])
        print(n, "value", round(d["value"]), "dist_mode", d.get("dist_mode"), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, "first_token", d["first_token"], "sha", d.get("first_token_logits_sha256", "")[:12], "verify", d["verify"]["ok"])
    except Exception as e:
        print(n, "failed:", e)
PY
  ;;
smoke)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -6 $OUT/smoke.log ;;
bench)
  timeout 1200 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/show_bench.py $OUT/bench.json 2>/dev/null | head -40 ;;
bench20)
  # the driver's command shape
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; echo "bench20 rc=$?"; python tools/show_bench.py $OUT/bench20.json 2>/dev/null | head -40 ;;
prof)
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r6 -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-preproc --no-verify --no-other-configs > $OUT/prof_bench.json 2> $OUT/prof.err)
  echo "prof rc=$?"; find $OUT/prof -name '*kernel_stats.csv' | head -3
  # keep the summary, drop the per-dispatch trace (tens of MB)
  find $OUT/prof -name '*kernel_trace.csv' -delete; find $OUT/prof -name '*.db' -delete ;;
mfma)
  CMD="python $REPO/bench.py --steps 1 --warmup 0 --decode-steps 2 --no-cpu-baseline --no-kernel-timer --no-preproc --no-verify --no-other-configs"
  (cd /tmp && timeout 1200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- $CMD > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err); echo "mfma rc=$?"
  python tools/mfma_busy_from_pmc.py "$(find $OUT/pmc_mfma -name '*counter_collection.csv' | head -1)" $OUT/mfma_busy.json
  rm -rf $OUT/pmc_mfma ;;
verify300)
  # the bench's in-run verification leg on the 5-min config (quick)
  timeout 600 python bench.py --frames 300 --steps 1 --warmup 1 --no-cpu-baseline --no-preproc --no-kernel-timer --decode-steps 4 > $OUT/bench_verify300.json 2> $OUT/bench_verify300.err; echo "verify300 rc=$?"
  python tools/show_bench.py $OUT/bench_verify300.json 2>/dev/null | grep -E "value|first_token|verify" || tail -5 $OUT/bench_verify300.err ;;
epi2)
  bash tools/lab/run_epi2_ab.sh 2>&1 | tail -60 ;;
abw4n)
  # same-box ABAB of the prefill: 288 x 224 tiles for N = 1 152 (default) against the 256-wide kernel everywhere (VIDI_W4N=0), one library
  : > $OUT/ab_w4n.jsonl
  for r in 1 2; do for sw in 0 1; do
    VIDI_W4N=$sw timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-preproc --decode-steps 4 2> $OUT/ab_w4n.err | grep '^{' | sed "s/^{/{\"VIDI_W4N\": $sw, /" >> $OUT/ab_w4n.jsonl; echo "abw4n $sw rc=$?"
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/ab_w4n.jsonl"):
    d = json.loads(l)
    print("VIDI_W4N", d["VIDI_W4N"], round(d["value"]), {k: round(v) for k, v in d["stage_ms_per_step"].items()}, "gemm TFLOP/s", round(d["kernel_families"]["gemm"]["TFLOP/s"]), "frac", round(d["roofline"]["frac"], 4), "verify", d["verify"]["ok"], round(d["verify"]["embeds_frames_max_err"], 4), "first_token", d["first_token"])
PY
  ;;
abqkv)
  # same-box ABAB of the prefill: SigLIP's q | k | v projection on 288 x 224 tiles (default) against the 256-wide kernel (VIDI_W4N_QKV=0), one library
  : > $OUT/ab_qkv.jsonl
  for r in 1 2; do for sw in 0 1; do
    VIDI_W4N_QKV=$sw timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-preproc --no-other-configs --decode-steps 4 2> $OUT/ab_qkv.err | grep '^{' | sed "s/^{/{\"VIDI_W4N_QKV\": $sw, /" >> $OUT/ab_qkv.jsonl; echo "abqkv $sw rc=$?"
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/ab_qkv.jsonl"):
    d = json.loads(l)
    print("VIDI_W4N_QKV", d["VIDI_W4N_QKV"], round(d["value"]), {k: round(v) for k, v in d["stage_ms_per_step"].items()}, "gemm TFLOP/s", round(d["kernel_families"]["gemm"]["TFLOP/s"]), "frac", round(d["roofline"]["frac"], 4), "verify", d["verify"]["ok"], round(d["verify"]["embeds_frames_max_err"], 4), "first_token", d["first_token"], "probe", round(d["box_reference"]["mfma_bf16_16x16x32_TFLOP/s"]))
PY
  ;;
abskinny)
  # same-box ABAB of the 5-min config (text prefill is a visible share there): the prompt's projections on vidi_gemm_skinny (default) against
  # the tile GEMM (VIDI_SKINNY_GEMM=0); `stage_ms_per_step.text_prefill` is the number to read
  : > $OUT/ab_skinny.jsonl
  for r in 1 2; do for sw in 0 1; do
    VIDI_SKINNY_GEMM=$sw timeout 300 python bench.py --frames 300 --steps 5 --warmup 2 --no-cpu-baseline --no-preproc --decode-steps 4 2> $OUT/ab_skinny.err | grep '^{' | sed "s/^{/{\"VIDI_SKINNY_GEMM\": $sw, /" >> $OUT/ab_skinny.jsonl; echo "abskinny $sw rc=$?"
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/ab_skinny.jsonl"):
    d = json.loads(l)
    print("VIDI_SKINNY_GEMM", d["VIDI_SKINNY_GEMM"], round(d["value"]), {k: round(v, 1) for k, v in d["stage_ms_per_step"].items()}, "verify", d["verify"]["ok"], "first_token", d["first_token"])
PY
  ;;
abepi)
  # same-box ABAB of the prefill: the product library (epilogue form 2) against tools/build_epi1.sh's build (form 1 everywhere, built in the container)
  : > $OUT/ab_epi.jsonl
  for r in 1 2; do for lib in libvidi_hip_epi1.so libvidi_hip.so; do
    VIDI_HIP_LIB=$REPO/vidi_amd/$lib timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-preproc --no-verify --decode-steps 4 2> $OUT/ab_epi.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", /" >> $OUT/ab_epi.jsonl; echo "abepi $lib rc=$?"
  done; done
  python - <<'PY'
import json
for l in open("gpurun_out/ab_epi.jsonl"):
    d = json.loads(l)
    print(d["lib"], round(d["value"]), {k: round(v) for k, v in d["stage_ms_per_step"].items()}, "gemm TFLOP/s", round(d["kernel_families"]["gemm"]["TFLOP/s"]), "frac", round(d["roofline"]["frac"], 4), "first_token", d["first_token"])
PY
  ;;
clock)
  bash tools/lab/run_clock.sh 2>&1 | tail -70 ;;
towerbound)
  bash tools/lab/run_tower_bound.sh 2>&1 | tail -90 ;;
dist8)
  # eight ranks sharing the one GPU (gloo transport): the BASELINE 8-way partition of bench.py end to end on a 10-minute video, and the same video on one rank
  VIDI_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --frames 600 --steps 1 --warmup 1 --no-preproc --no-kernel-timer --decode-steps 8 > $OUT/bench_dist8.json 2> $OUT/bench_dist8.err; echo "dist8 rc=$?"
  timeout 600 python bench.py --frames 600 --steps 1 --warmup 1 --no-preproc --no-cpu-baseline --no-kernel-timer --decode-steps 8 > $OUT/bench_dist8_ref1.json 2> $OUT/bench_dist8_ref1.err; echo "dist8 ref rc=$?"
  python tools/show_bench.py $OUT/bench_dist8.json $OUT/bench_dist8_ref1.json 2>/dev/null | grep -E "value|stages|first_token|verify" ;;
attn)
  # encoder-attention schedule variants (tools/build_variant.sh ... in the container), same-box A/B
  timeout 600 python tools/ab_attn.py vidi_amd/libvidi_hip.so $(ls vidi_amd/libvidi_hip_attn*.so) > $OUT/ab_attn.jsonl 2> $OUT/ab_attn.err; echo "attn rc=$?"; cat $OUT/ab_attn.jsonl ;;
variants)
  # regression sweep of the other bench configurations on the current build (BASELINE configs[1], [4]-shape, Vidi-7B, fp16, graph decode)
  # "cfg4": BASELINE configs[4] on one GPU — 30 min @2 fps (3 600 frames, 60 audio windows), 8 ragged prompts sharing the video, 128 decoded tokens
  for v in "--frames 300" "--queries 8" "--preset vidi_7b" "--dtype fp16" "--decode-graph" "--fps 2 --queries 8 --ragged-prompts 24 52 --decode-steps 128"; do
    n=$(echo $v | tr -d ' -' | cut -c1-24); timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-preproc $v > $OUT/bench_$n.json 2> $OUT/bench_$n.err; echo "bench $v rc=$?"
    python tools/show_bench.py $OUT/bench_$n.json 2>/dev/null | grep -E "value|stages"
  done ;;
labres)
  # residual-ring depth A/B in the GEMM lab (binaries built in the container: tools/lab/gemm_lab_rd{3,5,6}), interleaved per shape
  : > $OUT/lab_res.jsonl
  for shape in siglip_o siglip_fc2 siglip_fc1; do for v in w4p w4p_br w4p_bt; do for rd in old 3 5 6 old 3 5 6; do
    LAB_SHAPE=$shape timeout 120 tools/lab/gemm_lab_rd$rd $v 5 | sed "s/^{/{\"rd\": \"$rd\", /" >> $OUT/lab_res.jsonl
  done; done; done
  echo "labres rc=$?"; python - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/lab_res.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); acc[(d["shape"], d["variant"], d["rd"])].append(d["tflops"])
for k in sorted(acc): print(k, [round(x) for x in acc[k]])
PY
  ;;
decode)
  # decode step: per-switch end-to-end A/B on one engine, the cross-attention launch and the skinny projections against lab builds
  timeout 600 python tools/ab_decode.py e2e 32 4 > $OUT/ab_decode_e2e.jsonl 2> $OUT/ab_decode_e2e.err; echo "decode rc=$?"; cat $OUT/ab_decode_e2e.jsonl
  libs=$(ls vidi_amd/libvidi_hip_*.so 2>/dev/null)
  if [ -n "$libs" ]; then
    timeout 300 python tools/ab_decode.py cross vidi_amd/libvidi_hip.so $libs > $OUT/ab_decode_cross.jsonl 2> $OUT/ab_decode_cross.err; cat $OUT/ab_decode_cross.jsonl
    timeout 300 python tools/ab_decode.py gemv vidi_amd/libvidi_hip.so $libs > $OUT/ab_decode_gemv.jsonl 2> $OUT/ab_decode_gemv.err; cat $OUT/ab_decode_gemv.jsonl
  fi ;;
abln)
  timeout 900 python tools/ab_ln_fold.py 1440 3 > $OUT/ab_ln_fold.jsonl 2> $OUT/ab_ln_fold.err; echo "abln rc=$?"; cat $OUT/ab_ln_fold.jsonl ;;
dist2)
  # two ranks sharing the one GPU of the box (gloo transport; RCCL refuses two ranks on one device): the torchrun / sharded code path
  # of bench.py end to end, on a 10-minute video so both ranks fit
  # (bench.py launches its own ranks: no torchrun on the command line)
  VIDI_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --frames 600 --steps 1 --warmup 1 --no-preproc --no-other-configs > $OUT/bench_dist2.json 2> $OUT/bench_dist2.err; echo "dist2 rc=$?"
  VIDI_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --frames 600 --fps 2 --queries 8 --ragged-prompts 24 52 --decode-steps 64 --steps 1 --warmup 1 --no-preproc \
      > $OUT/bench_dist2_q8.json 2> $OUT/bench_dist2_q8.err; echo "dist2 q8 rc=$?"
  timeout 600 python bench.py --frames 600 --steps 1 --warmup 1 --no-preproc --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "dist1 rc=$?"
  python tools/show_bench.py $OUT/bench_dist2.json $OUT/bench_dist1.json 2>/dev/null | grep -E "value|stages" ;;
chunk)
  timeout 900 python tools/bench_vis_chunk.py 3600 > $OUT/vis_chunk.jsonl 2> $OUT/vis_chunk.err; echo "chunk rc=$?"; cat $OUT/vis_chunk.jsonl ;;
pmc)
  CMD="python $REPO/bench.py --steps 1 --warmup 0 --decode-steps 2 --no-cpu-baseline --no-kernel-timer --no-preproc --no-verify --no-other-configs"
  # rocprofv3's counter mode sometimes segfaults within seconds of the first dispatches of this command (round 4: 5 of 7 attempts, either
  # counter, different boxes; the same passes run through when retried): a failed attempt costs ~5 s, so each pass is tried up to 5 times
  for pass in "FETCH_SIZE pmc_fetch f" "WRITE_SIZE pmc_write w"; do
    set -- $pass
    for attempt in 1 2 3 4 5; do
      rm -rf $OUT/$2
      (cd /tmp && timeout 1200 rocprofv3 --pmc $1 --output-format csv -d $OUT/$2 -o $3 -- $CMD > $OUT/$2.json 2> $OUT/$2.err); rc=$?
      echo "$1 attempt $attempt rc=$rc"
      [ $rc -eq 0 ] && break
    done
  done
  F=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
  python tools/traffic_from_pmc.py "$F" "$W" $OUT/traffic.json 3600 vidi15_9b | head -c 4000
  # a bench run later in this call attaches the figure when it finds it under profiles/ (it is committed from gpurun_out/ afterwards)
  [ -n "$F" ] && [ -n "$W" ] && [ -s $OUT/traffic.json ] && cp $OUT/traffic.json $REPO/profiles/traffic.json
  # keep only the small summaries (the raw per-dispatch CSVs are tens of MB)
  rm -rf $OUT/pmc_fetch $OUT/pmc_write ;;
esac
done
