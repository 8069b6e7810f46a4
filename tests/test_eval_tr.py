"""vidi_amd/eval_tr.py (VUE-TR scorer hookup) against scores produced by EXECUTING the reference's own qa_eval.py
(tests/golden/make_golden_vue.py): a synthetic set that travels with the repo, and — where /root/reference is present — every
result file the reference ships (incl. the published Vidi rows)."""
import json
import math
import os
import warnings

import pytest

from vidi_amd import eval_tr as E

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_vue.json")))


def same(a, b):
    return (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 1e-12


def check(got, ref):
    for attr, r in ref.items():
        for k in ("precision", "recall", "iou"):
            assert same(got[attr][k], r[k]), (attr, k, got[attr][k], r[k])
        assert got[attr]["n"] == r["n"]


def test_synthetic_matches_reference_execution():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = E.score_predictions(os.path.join(GOLD, "vue_synth_pred.json"), os.path.join(GOLD, "vue_synth_gt.json"))
    check(got, REF["synthetic"])
    assert 0.0 < got["overall"]["iou"] < 1.0


@pytest.mark.skipif(not os.path.isdir("/root/reference/VUE_TR_V2"), reason="reference tree (ground truth + shipped results) not present")
@pytest.mark.parametrize("key", [k for k in REF if k != "synthetic"])
def test_shipped_results_match_reference_execution(key):
    tag, fname = key.split(":")
    d, gt = ("/root/reference/VUE_TR_V2", "VUE-TRv2_ground_truth.json") if tag == "v2" else ("/root/reference/VUE_TR", "VUE-TR_ground_truth.json")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = E.score_predictions(os.path.join(d, fname), os.path.join(d, gt), version=2 if tag == "v2" else 1)
    check(got, REF[key])


def test_answer_string_round_trip():
    """`ask()` prints HH:MM:SS spans (inference.py:59-66); the hookup turns them back into seconds for the scorer"""
    assert E.parse_time_ranges("00:15:46-00:15:53, 01:00:57-01:01:01") == [[946.0, 953.0], [3657.0, 3661.0]]
    assert E.parse_time_ranges("") == [] and E.parse_time_ranges("garbage 0.1-0.2") == []
    res = E.answers_to_results({0: "00:00:12-00:00:19", 3: ""})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = E.score_predictions(res + [{"query_id": i, "answer": []} for i in (1, 2, 4, 5, 6, 7)], os.path.join(GOLD, "vue_synth_gt.json"))
    assert got["overall"]["n"] == 8 and 0.0 < got["overall"]["iou"] < 1.0
