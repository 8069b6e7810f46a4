cd $GRAFT_REPO_ROOT 2>/dev/null || true
: > gpurun_out/chunk_sweep.jsonl
for r in 1 2; do for vc in 720 1200 1800 3600; do
  timeout 600 python bench.py --steps 2 --warmup 1 --vis-chunk $vc --no-cpu-baseline --no-preproc --no-verify --no-other-configs --no-kernel-timer --decode-steps 2 2>/dev/null | grep '^{' | sed "s/^{/{\"vis_chunk\": $vc, /" >> gpurun_out/chunk_sweep.jsonl
done; done
for ac in 60 120; do
  timeout 600 python bench.py --steps 2 --warmup 1 --aud-chunk $ac --no-cpu-baseline --no-preproc --no-verify --no-other-configs --no-kernel-timer --decode-steps 2 2>/dev/null | grep '^{' | sed "s/^{/{\"aud_chunk\": $ac, /" >> gpurun_out/chunk_sweep.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/chunk_sweep.jsonl"):
    d=json.loads(l); print({k:d[k] for k in ("vis_chunk","aud_chunk") if k in d}, round(d["value"]), {k:round(v,1) for k,v in d["stage_ms_per_step"].items()}, d["first_token_logits_sha256"][:10])
PY
