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
