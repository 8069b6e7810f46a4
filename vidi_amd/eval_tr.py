"""VUE-TR / VUE-TRv2 temporal-retrieval scoring of `ask()`-style answers: the metric side of the
reference's `VUE_TR_V2/qa_eval.py` (:105-165 IoU curve and AUC, :208-300 merge / intersection / union / precision / recall,
:303-342 result loading) without its plotting.  Pure host code (numpy); the accuracy regression hook for real checkpoints:

    answers = {query_id: model_answer_string}            # "00:15:46-00:15:53, ..." as inference.py:52-66 prints it
    scores  = score_predictions(answers_to_results(answers), "VUE-TRv2_ground_truth.json")
    -> {"overall": {"precision": .., "recall": .., "iou": ..}, "ultra-short": {...}, ...}

Pinned by executing the reference's own qa_eval.py on the result files it ships (tests/golden/make_golden_vue.py)."""
from __future__ import annotations

import copy
import json
import re
from typing import Dict, Iterable, List, Sequence

import numpy as np

ATTRIBUTES = ["ultra-short", "short", "medium", "long", "ultra-long", "keyword", "phrase", "sentence", "vision", "audio",
              "vision+audio", "overall"]
_THRES = np.linspace(0, 1, 101)


def _trapz(y, x):
    f = getattr(np, "trapezoid", None) or np.trapz
    return f(y, x)


def parse_time_ranges(answer: str) -> List[List[float]]:
    """'HH:MM:SS-HH:MM:SS, ...' (what `ask()` returns, inference.py:59-66) -> [[start_s, end_s], ...]"""
    out = []
    for a, b in re.findall(r"(\d+:\d\d:\d\d)-(\d+:\d\d:\d\d)", answer):
        s = [int(x) for x in a.split(":")]
        e = [int(x) for x in b.split(":")]
        out.append([float(s[0] * 3600 + s[1] * 60 + s[2]), float(e[0] * 3600 + e[1] * 60 + e[2])])
    return out


def answers_to_results(answers: Dict[int, str]) -> List[dict]:
    return [{"query_id": int(q), "answer": parse_time_ranges(a)} for q, a in answers.items()]


def merge_time_spans(intervals: np.ndarray) -> np.ndarray:
    """qa_eval.py:208-223: sort by start, merge overlapping or touching spans"""
    if len(intervals) == 0:
        return np.array([])
    intervals = intervals[np.argsort(intervals[:, 0])]
    merged = [intervals[0]]
    for cur in intervals[1:]:
        if cur[0] <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], cur[1])
        else:
            merged.append(cur)
    return np.array(merged)


def overlap_ratio(pred: np.ndarray, gt: np.ndarray) -> float:
    """qa_eval.py:105-137 — IoU of two span sets (pred merged first; intersection summed over all pairs)"""
    if len(gt) == 0 or gt.shape[0] == 0:
        return 1.0 if (len(pred) == 0 or pred.shape[0] == 0) else 0.0
    if len(pred) == 0 or pred.shape[0] == 0:
        return 0.0
    pred = merge_time_spans(pred)
    len_gt = np.sum(gt[:, 1] - gt[:, 0])
    pred = pred[pred[:, 0] <= pred[:, 1]]
    if pred.shape[1] == 0:
        return 0.0
    len_pred = np.sum(pred[:, 1] - pred[:, 0])
    intersect = 0
    for p in pred:
        for g in gt:
            intersect += np.maximum(0.0, np.minimum(p[1], g[1]) - np.maximum(p[0], g[0]))
    union = len_pred + len_gt - intersect
    return float(np.maximum(np.minimum(1.0, intersect / (union + 1e-16)), 0.0))


def success_overlap(results: Sequence[dict]):
    """qa_eval.py:140-153 — fraction of queries with IoU > t over 101 thresholds, and its AUC"""
    n = len(results)
    iou = np.array([overlap_ratio(np.array(r["answer"]), r["gt"]) for r in results]) if n else np.zeros(0)
    success = np.array([np.sum(iou > t) / float(n + 1e-16) for t in _THRES])
    return success, float(_trapz(success, _THRES))


def interval_intersection(a: List[List[float]], b: List[List[float]]):
    """qa_eval.py:226-246 (two-pointer sweep over the lists as given)"""
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        a0, a1 = a[i]
        b0, b1 = b[j]
        if a0 <= b1 and b0 <= a1:
            out.append((max(a0, b0), min(a1, b1)))
        if a1 < b1:
            i += 1
        else:
            j += 1
    return out


def interval_union(a: List[List[float]], b: List[List[float]]):
    """qa_eval.py:249-266"""
    intervals = sorted(a + b)
    out = []
    if intervals:
        cur = intervals[0]
        for it in intervals[1:]:
            if it[0] <= cur[1]:
                cur[1] = max(cur[1], it[1])
            else:
                out.append(cur)
                cur = it
        out.append(cur)
    return out


def compute_precision_recall(results: Sequence[dict], avg: bool = True, version: int = 2):
    """qa_eval.py:269-300.  version 1 = VUE_TR/qa_eval.py, which has no `empty gt and empty prediction -> precision 1` rule"""
    gt_all, pred_all, inter_all = [], [], []
    for item in results:
        gt = [[min(x), max(x)] for x in item["gt"] if len(x) == 2]
        pred = [[min(x), max(x)] for x in item["answer"] if len(x) == 2]
        inter = interval_intersection(copy.deepcopy(gt), copy.deepcopy(pred))
        gt_all.append(sum(x[1] - x[0] for x in gt))
        pred_all.append(sum(x[1] - x[0] for x in pred))
        inter_all.append(sum(x[1] - x[0] for x in inter))
    recall = np.array([i / g for i, g in zip(inter_all, gt_all) if g != 0])
    precision = []
    for i, g, p in zip(inter_all, gt_all, pred_all):
        if version >= 2 and g == 0 and p == 0:
            precision.append(1.0)
        elif p != 0:
            precision.append(i / p)
    precision = np.array(precision)
    if not avg:
        return precision, recall
    pt = np.array([np.mean(precision >= t) for t in _THRES])
    rt = np.array([np.mean(recall >= t) for t in _THRES])
    return float(_trapz(pt, _THRES)), float(_trapz(rt, _THRES))


def load_result(gt_path: str, predictions) -> List[dict]:
    """qa_eval.py:303-342: predictions = path to a .json/.jsonl file or a list of {'query_id'|'id', 'answer'} records;
    answers are widened to whole seconds (floor start, ceil end) and joined with the ground truth by query id."""
    with open(gt_path) as f:
        gts = {g["query_id"]: g for g in json.load(f)}
    if isinstance(predictions, str):
        with open(predictions) as f:
            preds = json.load(f) if predictions.endswith("json") else [json.loads(x) for x in f.readlines()]
    else:
        preds = [dict(p) for p in predictions]
    for p in preds:
        qid = p["query_id"] if "query_id" in p else p["id"]
        if len(p["answer"]) == 0 or (len(p["answer"]) == 1 and len(p["answer"][0]) == 0):
            p["answer"] = np.array([])
        else:
            a = np.array(p["answer"])
            a[:, 0] = np.floor(a[:, 0])
            a[:, 1] = np.ceil(a[:, 1])
            p["answer"] = a
        p.update(gts[qid])
        p["gt"] = np.array(p["gt"])
    return preds


def _subset(results: Sequence[dict], attr: str):
    if attr in ("ultra-short", "short", "medium", "long", "ultra-long"):
        return [r for r in results if r["duration_category"] == attr]
    if attr in ("keyword", "phrase", "sentence"):
        return [r for r in results if r["query_format"] == attr]
    if attr in ("audio", "vision", "vision+audio"):
        return [r for r in results if r["query_modality"] == attr]
    return list(results)


def score_predictions(predictions, gt_path: str, attributes: Iterable[str] = ATTRIBUTES, version: int = 2) -> Dict[str, Dict[str, float]]:
    """overall + per-attribute precision / recall / IoU AUCs (qa_eval.py:168-205, 365-384), as fractions in [0, 1]"""
    results = load_result(gt_path, predictions)
    out = {}
    for attr in attributes:
        sub = _subset(results, attr)
        _, iou = success_overlap(sub)
        pre, rec = compute_precision_recall(sub, version=version)
        out[attr] = {"precision": pre, "recall": rec, "iou": iou, "n": len(sub)}
    return out
