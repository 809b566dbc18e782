"""Pin the CPU oracle (oracle/) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU-only; this is what makes the oracle trustworthy as the checker
for the HIP kernels."""
import hashlib

import numpy as np
import pytest

import synth
from oracle import oracle
from util import close, load_golden

STATS = ("MIN", "MAX", "MED", "AVG", "STD", "NUM")


def _cases():
    return {c["name"]: c["values"] for c in synth.section_stat_cases()}


def test_inputs_regenerate_bit_exactly():
    g = load_golden("section_stats.json")
    cases = _cases()
    for rec in g["cases"]:
        v = np.asarray(cases[rec["name"]], dtype=np.float32)
        assert hashlib.sha256(v.tobytes()).hexdigest() == rec["sha256"], rec["name"]


def test_section_stats_match_reference_detector():
    """oracle_section_stats == reference Detector._get_section_summaries (straggler.py:172-197)."""
    g = load_golden("section_stats.json")
    cases = _cases()
    for rec in g["cases"]:
        kept = synth.retained(cases[rec["name"]], rec["ring_capacity"])
        got = oracle.section_stats(kept.astype(np.float64))
        exp = rec["expected"]
        for i, k in enumerate(STATS):
            if k in ("MIN", "MAX", "MED", "NUM"):
                assert got[i] == exp[k], (rec["name"], k, got[i], exp[k])  # selections are exact
            else:
                assert close(got[i], exp[k], rel=1e-12), (rec["name"], k, got[i], exp[k])


def test_kernel_stats_match_reference_computeStats_golden():
    """oracle_kernel_stats == reference computeStats (CuptiProfiler.cpp:44-74), bit for bit."""
    g = load_golden("native.json")
    cases = _cases()
    for rec in g["compute_stats"]:
        got = oracle.kernel_stats(cases[rec["name"]])
        for i in range(6):
            a, b = got[i], rec["expected"][i]
            assert (np.isnan(a) and np.isnan(b)) or a == b, (rec["name"], i, a, b)


def _numpy_compute_stats(x):
    """CuptiProfiler.cpp:44-74 restated a SECOND time, independently of oracle/straggler_oracle.c and of anything compiled
    from the reference: NumPy f32 scalars, one operation per statement of the source.  ``std::accumulate(..., 0.0f)`` is a
    sequential f32 sum in sorted order; ``/ n`` divides an f32 by a size_t, i.e. in f32; the squared deviations accumulate
    in f32 in the same order; ``std::sqrt`` of an f32."""
    f = np.float32
    x = np.sort(np.asarray(x, dtype=np.float32), kind="stable")
    n = x.size
    if n == 0:
        return [float("nan")] * 5 + [0.0]
    med = (x[n // 2 - 1] + x[n // 2]) / f(2) if n % 2 == 0 else x[n // 2]
    acc = f(0)
    for v in x:
        acc = f(acc + v)
    avg = f(acc / f(n))
    sq = f(0)
    for v in x:
        d = f(v - avg)
        sq = f(sq + f(d * d))
    return [float(x[0]), float(x[-1]), float(med), float(avg), float(np.sqrt(f(sq / f(n)))), float(n)]


def test_kernel_stats_equal_a_numpy_restatement_on_the_golden_rows_and_on_random_rows():
    """The pin of ``oracle_kernel_stats`` that needs NO reference build: (1) a second, NumPy restatement of
    CuptiProfiler.cpp:44-74 agrees with the C restatement bit for bit on every golden row, on lognormal rows of the sizes the
    rings hold and on rows of repeated / negative / huge values; (2) the one value the reference's own unit test holds for
    this routine (tests/straggler/unit/test_cupti_ext.py:98-116: 21 durations into a ring of 7 -> ``num_calls == 7``, over the
    newest 7) comes out of both.  The library compiled
    behind the stand-in cupti.h (``oracle.ref_lib``) is a second witness on top of this, not the pin."""
    cases = _cases()
    rng = np.random.default_rng(11)
    rows = [np.asarray(v, dtype=np.float32) for v in cases.values()]
    rows += [rng.lognormal(2.0, 1.0, n).astype(np.float32) for n in (1, 2, 3, 10, 11, 100, 1000, 8192)]
    rows += [np.full(17, 3.25, np.float32), np.array([-5, 2, -7.5, 1e30, 2, 2], np.float32), np.array([1e-40, 3e-41], np.float32)]
    with np.errstate(over="ignore", invalid="ignore"):
        for x in rows:
            a, b = oracle.kernel_stats(x), _numpy_compute_stats(x)
            assert np.array_equal(a, np.asarray(b, dtype=np.float64), equal_nan=True), (x[:8], a, b)
    # the committed golden rows (made through the stand-in build) say what the NumPy restatement says: the fixture is not
    # taken on the stand-in build's word
    with np.errstate(over="ignore", invalid="ignore"):
        for rec in load_golden("native.json")["compute_stats"]:
            b = _numpy_compute_stats(cases[rec["name"]])
            assert all((np.isnan(u) and np.isnan(v)) or u == v for u, v in zip(b, rec["expected"])), (rec["name"], b, rec["expected"])
    durations = rng.lognormal(2.0, 0.2, 21).astype(np.float32)
    kept = oracle.ring_run(durations, 7)
    assert np.array_equal(kept, durations[-7:])
    assert oracle.kernel_stats(kept)[5] == 7 == _numpy_compute_stats(kept)[5]
    assert np.array_equal(oracle.kernel_stats(kept), np.asarray(_numpy_compute_stats(durations[-7:])))


def test_kernel_stats_match_live_reference_build():
    """Same, against oracle/_ref/libnvrx_ref.so (the reference's computeStats compiled behind a stand-in cupti.h: a second
    witness -- the pin that needs no reference build is the NumPy restatement above)."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 3, 10, 11, 1000, 8192):
        x = rng.lognormal(2.0, 1.0, n).astype(np.float32)
        a, b = oracle.kernel_stats(x), oracle.ref_kernel_stats(x)
        assert np.array_equal(a, b, equal_nan=True), (n, a, b)


def test_ring_matches_reference_circular_buffer():
    g = load_golden("native.json")
    for rec in g["ring"]:
        vals = np.arange(rec["n"], dtype=np.float32) * 0.5 + 1.0
        lin = oracle.ring_run(vals, rec["capacity"])
        assert lin.size == rec["size"]
        assert hashlib.sha256(lin.tobytes()).hexdigest() == rec["sha256"]
        if oracle.ring_ref_lib() is not None:   # (the reference's CircularBuffer.h from its own file alone: no stand-in header)
            assert np.array_equal(lin, oracle.ref_ring_run(vals, rec["capacity"]))


def _summ(d):
    return {n: {k: v for k, v in s.items()} for n, s in d.items()}


def _scenario(idx):
    """Golden scenario by index: the hand-written ones (scoring.json) first, then the random ones (scoring_fuzz.json)."""
    fixed = load_golden("scoring.json")["scenarios"]
    return fixed[idx] if idx < len(fixed) else load_golden("scoring_fuzz.json")["scenarios"][idx - len(fixed)]


N_FIXED, N_FUZZ = 11, 48


@pytest.mark.parametrize("idx", range(N_FIXED + N_FUZZ))
def test_ref_port_scoring_matches_reference_report_generator(idx):
    """RefPortReportGenerator (dict-level port) == reference ReportGenerator on gloo ranks: the eleven hand-written
    scenarios and 48 random ones (names missing on some ranks / appearing mid-run, ranks without kernels, 1-5 ranks,
    1-4 reports, every combination of score families and gather_on_rank0)."""
    g = _scenario(idx)
    sc = g["scenario"]
    W = sc["world_size"]
    port = oracle.RefPortReportGenerator(W, sc["scores_to_compute"], sc["gather_on_rank0"])
    for t, step in enumerate(sc["steps"]):
        res = port.generate_reports([_summ(step[r][0]) for r in range(W)], [_summ(step[r][1]) for r in range(W)])
        if sc["gather_on_rank0"]:
            exp = g["per_rank"][0]["reports"][t]
            for r in range(W):
                for mine, key in ((res["gpu_rel"], "gpu_relative_perf_scores"), (res["gpu_indiv"], "gpu_individual_perf_scores")):
                    if exp[key]:
                        assert close(mine[r], exp[key][str(r)], rel=1e-7), (sc["name"], t, r, key)
                    else:
                        assert not mine
                for mine, key in ((res["sec_rel"], "section_relative_perf_scores"), (res["sec_indiv"], "section_individual_perf_scores")):
                    assert set(mine.keys()) == set(exp[key].keys()), (sc["name"], t, key)
                    for n in mine:
                        assert close(mine[n][r], exp[key][n][str(r)], rel=1e-7), (sc["name"], t, r, key, n)
            for r in range(1, W):
                assert g["per_rank"][r]["reports"][t] is None
        else:
            for r in range(W):
                exp = g["per_rank"][r]["reports"][t]
                if exp["gpu_relative_perf_scores"]:
                    assert close(res["gpu_rel"][r], exp["gpu_relative_perf_scores"][str(r)], rel=1e-12)
                if exp["gpu_individual_perf_scores"]:
                    assert close(res["gpu_indiv"][r], exp["gpu_individual_perf_scores"][str(r)], rel=1e-12)
                for n, v in exp["section_relative_perf_scores"].items():
                    assert close(res["sec_rel"][r][n], v[str(r)], rel=1e-12)
                for n, v in exp["section_individual_perf_scores"].items():
                    assert close(res["sec_indiv"][r][n], v[str(r)], rel=1e-12)
    # name -> id agreement (name_mapper.py:54-81)
    if sc["gather_on_rank0"] or "relative_perf_scores" in sc["scores_to_compute"]:
        assert port.section_ids == g["per_rank"][0]["ids"]["sections"]
        assert port.kernel_ids == g["per_rank"][0]["ids"]["kernels"]


def _table_from_step(step, W, kid, sid, hist):
    """Build the [R, L] exchange table (layout: oracle/straggler_oracle.c) from per-rank summaries."""
    K, S = len(kid), len(sid)
    L = oracle.table_len(K, S)
    T = np.zeros((W, L), dtype=np.float32)
    T[:, : K + S] = -1.0
    T[:, K + S : 2 * (K + S)] = np.nan
    for r in range(W):
        sec, ker = step[r]
        for k, s in ker.items():
            if "ncclDev" in k:
                continue
            j = kid[k]
            T[r, j] = s["MED"]
            hist[r][("k", k)] = min(hist[r].get(("k", k), np.inf), np.float32(s["MED"]))
            T[r, K + S + j] = hist[r][("k", k)]
            T[r, 2 * (K + S) + j] = np.float32(s["NUM"]) * np.float32(s["AVG"])
        for n, s in sec.items():
            j = K + sid[n]
            T[r, j] = s["MED"]
            hist[r][("s", n)] = min(hist[r].get(("s", n), np.inf), np.float32(s["MED"]))
            T[r, K + S + j] = hist[r][("s", n)]
        T[r, L - 1] = 1.0
    return T


@pytest.mark.parametrize("idx", range(N_FIXED + N_FUZZ))
def test_table_scoring_matches_reference(idx):
    """oracle_score_table (array form, what the HIP score kernel is checked against) reproduces the
    reference's gathered scores within f32 rounding (contract: 1e-4; observed <= 1e-6)."""
    g = _scenario(idx)
    sc = g["scenario"]
    if not sc["gather_on_rank0"]:
        pytest.skip("gathered form only")
    W = sc["world_size"]
    do_rel = "relative_perf_scores" in sc["scores_to_compute"]
    do_ind = "individual_perf_scores" in sc["scores_to_compute"]
    port = oracle.RefPortReportGenerator(W, sc["scores_to_compute"], True)
    hist = [dict() for _ in range(W)]
    for t, step in enumerate(sc["steps"]):
        port._gather_and_assign_ids(
            [[k for k in step[r][1] if "ncclDev" not in k] for r in range(W)], [list(step[r][0]) for r in range(W)]
        )
        kid, sid = port.kernel_ids, port.section_ids
        K, S = len(kid), len(sid)
        T = _table_from_step(step, W, kid, sid, hist)
        sc_out = oracle.score_table(T, K, S, do_ind, do_rel)
        exp = g["per_rank"][0]["reports"][t]
        for r in range(W):
            if do_ind:
                assert close(sc_out[r, 0], exp["gpu_individual_perf_scores"][str(r)], rel=1e-6), (t, r)
            if do_rel:
                assert close(sc_out[r, 1], exp["gpu_relative_perf_scores"][str(r)], rel=1e-6), (t, r)
            for n, j in sid.items():
                if do_ind:
                    assert close(sc_out[r, 2 + j], exp["section_individual_perf_scores"][n][str(r)], rel=1e-6), (t, r, n)
                if do_rel:
                    assert close(sc_out[r, 2 + S + j], exp["section_relative_perf_scores"][n][str(r)], rel=1e-6), (t, r, n)


def test_stress_golden_vs_oracle():
    """8 ranks x 64 sections x 10k samples through the reference Detector (configs #3/#5):
    the oracle reproduces summaries, scores and flagged sets from the regenerated inputs."""
    g = load_golden("stress.json")
    W = 8
    hist = [dict() for _ in range(W)]
    sid = {synth.section_name(s): s for s in range(64)}
    for var in g["variants"]:
        exp = g["rank0"][var["name"]]
        h = hashlib.sha256()
        step = []
        for r in range(W):
            x = synth.stress_samples(r, var["S"], var["n"], var["slow_rank"], var["slow_factor"])
            h.update(x.tobytes())
            kept = x[:, -synth.RING_CAPACITY:]
            st = oracle.rows_stats(kept, np.full(var["S"], kept.shape[1], dtype=np.uint32))
            step.append(({synth.section_name(s): dict(zip(STATS, st[s])) for s in range(var["S"])}, {}))
            if r == 0:
                for s in range(var["S"]):
                    e = exp["local_section_summaries"][synth.section_name(s)]
                    for i, k in enumerate(STATS):
                        assert close(st[s, i], e[k], rel=1e-12), (var["name"], s, k)
        assert h.hexdigest() == g["input_sha256"][var["name"]]
        T = _table_from_step(step, W, {}, sid, hist)
        out = oracle.score_table(T, 0, 64, True, True)
        for r in range(W):
            assert np.isnan(out[r, 0]) and np.isnan(out[r, 1])
            for n, j in sid.items():
                assert close(out[r, 2 + j], exp["section_individual_perf_scores"][n][str(r)], rel=1e-6)
                assert close(out[r, 2 + 64 + j], exp["section_relative_perf_scores"][n][str(r)], rel=1e-6)
        for thr in ("0.75", "0.9"):
            flagged_rel = {n: sorted(int(r) for r in range(W) if out[r, 2 + 64 + j] < float(thr)) for n, j in sid.items()}
            flagged_rel = {n: v for n, v in flagged_rel.items() if v}
            assert flagged_rel == exp["stragglers"][thr]["straggler_sections_relative"], (var["name"], thr)
            flagged_ind = {n: sorted(int(r) for r in range(W) if out[r, 2 + j] < float(thr)) for n, j in sid.items()}
            flagged_ind = {n: v for n, v in flagged_ind.items() if v}
            assert flagged_ind == exp["stragglers"][thr]["straggler_sections_individual"], (var["name"], thr)


def test_port_job_on_gloo_ranks_runs_and_reports_times():
    """The bench's CPU baseline with real collectives (oracle/port_mp.py): two gloo ranks, tiny shape."""
    from oracle import port_mp

    r = port_mp.run(world=2, sections=4, samples=200, reps=3, timeout=120)
    assert r["ranks"] == 2 and r["reps"] == 3
    for k in ("summaries_us", "exchange_scoring_us", "all_gather_object_us", "report_us"):
        assert r[k] > 0.0
    assert r["report_us"] >= 0.5 * (r["summaries_us"] + r["exchange_scoring_us"])


# --------------------------------------------------------------------------------------------------
# property tests: the oracle against the arithmetic it restates, on inputs nobody chose
# --------------------------------------------------------------------------------------------------
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402
from hypothesis.extra import numpy as hnp  # noqa: E402

_F32 = st.floats(min_value=-float(np.float32(1e30)), max_value=float(np.float32(1e30)), allow_nan=False, allow_infinity=False, width=32)
_PROP = settings(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@_PROP
@given(hnp.arrays(np.float32, st.integers(0, 300), elements=_F32))
def test_kernel_stats_equal_the_live_reference_on_random_rows(x):
    """oracle_kernel_stats vs the reference's own computeStats (CuptiProfiler.cpp:44-74, compiled into oracle/_ref from
    where it lies): every output bit for bit -- sorted-order f32 sums, mean-of-middles median, population deviation,
    the all-NaN / 0 result of an empty row -- on arbitrary finite f32 rows (duplicates, negatives, denormals, 1e30)."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    a, b = oracle.kernel_stats(x), oracle.ref_kernel_stats(x)
    assert np.array_equal(a, b, equal_nan=True), (x.tolist(), a, b)


@_PROP
@given(st.integers(0, 200), st.integers(1, 64))
def test_ring_equals_the_live_reference_for_any_length_and_capacity(n, capacity):
    """Overwrite-oldest ring (CircularBuffer.h:53-69): what survives, and in which order.  The checker is the reference's
    header compiled from its own file alone (oracle/_ref/libnvrx_ring_ref.so: nothing stands in for anything)."""
    if oracle.ring_ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    vals = (np.arange(n, dtype=np.float32) * 0.25 - 3.0)
    assert np.array_equal(oracle.ring_run(vals, capacity), oracle.ref_ring_run(vals, capacity))
    assert np.array_equal(oracle.ring_run(vals, capacity), vals[-capacity:] if n > capacity else vals)


@pytest.mark.filterwarnings("ignore:std\\(\\)")
@_PROP
@given(hnp.arrays(np.float64, st.integers(1, 200), elements=st.floats(min_value=-1e12, max_value=1e12, allow_nan=False, width=64)))
def test_section_stats_equal_torch_on_random_rows(x):
    """oracle_section_stats vs the five torch reductions of straggler.py:185-195 on an f64 tensor (the third-party
    arithmetic the reference delegates to): selections exact (torch's LOWER median), mean / unbiased deviation to
    1e-12, NaN deviation of a single sample."""
    import torch

    t = torch.tensor(x.tolist(), dtype=torch.float64)
    got = oracle.section_stats(x)
    assert got[0] == torch.min(t).item() and got[1] == torch.max(t).item() and got[2] == torch.median(t).item()
    assert got[5] == len(x)
    assert close(got[3], torch.mean(t).item(), rel=1e-12, abs_=1e-300)
    std = torch.std(t).item()
    assert close(got[4], std, rel=1e-9, abs_=1e-9 * max(1.0, float(np.abs(x).max()))) or (np.isnan(std) and np.isnan(got[4]))


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(hnp.arrays(np.float32, st.integers(0, 120), elements=_F32))
def test_kernel_stats_equal_a_numpy_restatement_on_arbitrary_rows(x):
    """The same pin on arbitrary finite f32 rows (duplicates, negatives, denormals, 1e30 -- sums that overflow to inf included)."""
    with np.errstate(over="ignore", invalid="ignore"):
        a, b = oracle.kernel_stats(x), _numpy_compute_stats(x)
    assert np.array_equal(a, np.asarray(b, dtype=np.float64), equal_nan=True), (x.tolist(), a, b)
