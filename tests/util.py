"""Helpers shared by the test modules."""
import json
import math
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unj(x):
    """Inverse of make_golden._jsonable for floats encoded as 'nan'/'inf'."""
    if isinstance(x, dict):
        return {k: unj(v) for k, v in x.items()}
    if isinstance(x, list):
        return [unj(v) for v in x]
    if x == "nan":
        return float("nan")
    if x == "inf":
        return float("inf")
    if x == "-inf":
        return float("-inf")
    return x


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return unj(json.load(f))


def close(a, b, rel=1e-6, abs_=0.0):
    """NaN-aware closeness."""
    a = float(a)
    b = float(b)
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    if math.isinf(a) or math.isinf(b):
        return a == b
    return abs(a - b) <= max(rel * max(abs(a), abs(b)), abs_)


def compare_reports(got, exp, tag, rel=1e-6):
    """``got``: workers.report_to_plain of our Report; ``exp``: the reference's report as stored in the golden files."""
    if exp is None:
        assert got is None, tag
        return
    assert got is not None, tag
    for key in ("gpu_relative_perf_scores", "gpu_individual_perf_scores"):
        assert set(map(str, got[key].keys())) == set(exp[key].keys()), (tag, key)
        for r, v in got[key].items():
            assert close(v, exp[key][str(r)], rel=rel), (tag, key, r, v, exp[key][str(r)])
    for key in ("section_relative_perf_scores", "section_individual_perf_scores"):
        assert set(got[key].keys()) == set(exp[key].keys()), (tag, key, got[key].keys(), exp[key].keys())
        for n, per_rank in got[key].items():
            assert set(map(str, per_rank.keys())) == set(exp[key][n].keys()), (tag, key, n)
            for r, v in per_rank.items():
                assert close(v, exp[key][n][str(r)], rel=rel), (tag, key, n, r, v, exp[key][n][str(r)])
    assert {str(k): v for k, v in got["rank_to_node"].items()} == exp["rank_to_node"], tag
    assert got["gather_on_rank0"] == exp["gather_on_rank0"] and got["rank"] == exp["rank"]
    for thr, e in exp.get("stragglers", {}).items():
        assert got["stragglers"][thr] == e, (tag, thr)
