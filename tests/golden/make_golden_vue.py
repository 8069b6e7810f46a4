"""Golden scores for vidi_amd/eval_tr.py produced by EXECUTING the reference's own VUE_TR_V2/qa_eval.py and VUE_TR/qa_eval.py
(metric functions only; plotting untouched) on the result files and ground truth they ship.  Build container only:
    python tests/golden/make_golden_vue.py  ->  tests/golden/reference_vue.json
The ground-truth / result JSON files stay in /root/reference (not copied): the CPU test that uses them is skipped where the
reference tree is absent; a small synthetic subset with reference-computed scores is stored for everywhere else."""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def scores(Q, gt_path, res_path):
    results = Q.load_result(gt_path, res_path)
    out = {}
    for attr in ["ultra-short", "short", "medium", "long", "ultra-long", "keyword", "phrase", "sentence", "vision", "audio",
                 "vision+audio", "overall"]:
        if attr in ["ultra-short", "short", "medium", "long", "ultra-long"]:
            sub = [r for r in results if r["duration_category"] == attr]
        elif attr in ["keyword", "phrase", "sentence"]:
            sub = [r for r in results if r["query_format"] == attr]
        elif attr in ["audio", "vision", "vision+audio"]:
            sub = [r for r in results if r["query_modality"] == attr]
        else:
            sub = results
        _, iou = Q.success_overlap(sub)
        pre, rec = Q.compute_precision_recall(sub)
        out[attr] = {"precision": float(pre), "recall": float(rec), "iou": float(iou), "n": len(sub)}
    return out


def main():
    import matplotlib
    matplotlib.use("Agg")
    out = {}
    for tag, d, gt in (("v2", "/root/reference/VUE_TR_V2", "VUE-TRv2_ground_truth.json"), ("v1", "/root/reference/VUE_TR", "VUE-TR_ground_truth.json")):
        Q = load(f"ref_qa_eval_{tag}", os.path.join(d, "qa_eval.py"))
        for f in sorted(os.listdir(d)):
            if f.startswith("results_") and f.endswith(".json"):
                out[f"{tag}:{f}"] = scores(Q, os.path.join(d, gt), os.path.join(d, f))
                print(tag, f, {k: round(v, 4) for k, v in out[f"{tag}:{f}"]["overall"].items()})
    # synthetic subset that travels with the repo: hand-made spans covering merge / empty / reversed / multi-span cases
    Q = sys.modules["ref_qa_eval_v2"]
    gt = [{"query_id": i, "gt": g, "duration_category": c, "query_format": "phrase", "query_modality": "vision"} for i, (g, c) in enumerate([
        ([[10, 20]], "short"), ([[0, 5], [30, 40]], "short"), ([[100, 200]], "medium"), ([], "medium"), ([[50, 60]], "long"),
        ([[5, 15], [15, 25]], "long"), ([[1000, 1010]], "ultra-long"), ([[7, 9]], "ultra-short")])]
    pred = [{"query_id": 0, "answer": [[12.3, 18.7]]}, {"query_id": 1, "answer": [[2.2, 4.1], [3.0, 33.5]]}, {"query_id": 2, "answer": []},
            {"query_id": 3, "answer": []}, {"query_id": 4, "answer": [[70.5, 80.2]]}, {"query_id": 5, "answer": [[0.1, 30.9]]},
            {"query_id": 6, "answer": [[1005.5, 1001.2]]}, {"query_id": 7, "answer": [[6.9, 9.4], [8.0, 8.5]]}]
    gp, pp = os.path.join(HERE, "vue_synth_gt.json"), os.path.join(HERE, "vue_synth_pred.json")
    json.dump(gt, open(gp, "w")); json.dump(pred, open(pp, "w"))
    out["synthetic"] = scores(Q, gp, pp)
    json.dump(out, open(os.path.join(HERE, "reference_vue.json"), "w"), indent=1)
    print("wrote reference_vue.json:", list(out))


if __name__ == "__main__":
    main()
