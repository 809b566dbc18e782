"""GPU parity of the score kernel and of ReportGenerator's dict-input path."""
import numpy as np
import pytest
import torch

from oracle import oracle
from util import close, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from nvrx_straggler.backend import get_backend

    return get_backend()


def _random_table(rng, R, K, S, p_missing=0.15):
    L = oracle.table_len(K, S)
    KS = K + S
    T = np.zeros((R, L), dtype=np.float32)
    med = rng.lognormal(1.0, 0.5, (R, KS)).astype(np.float32)
    hmin = (med * rng.uniform(0.5, 1.0, (R, KS))).astype(np.float32)
    missing = rng.random((R, KS)) < p_missing
    med[missing] = -1.0
    hmin[missing] = np.nan
    T[:, :KS] = med
    T[:, KS : 2 * KS] = hmin
    w = rng.uniform(1, 1000, (R, K)).astype(np.float32)
    w[missing[:, :K]] = 0.0
    T[:, 2 * KS : 2 * KS + K] = w
    T[:, L - 1] = 1.0
    return T


def _score(be, T, K, S, do_indiv=True, do_rel=True, thr=(0.75, 0.75, 0.75, 0.75)):
    R = T.shape[0]
    ws = be.workspace(R, K, S, R, 0)
    ws.send.copy_(torch.from_numpy(T))
    be.score(ws, ws.send, do_indiv, do_rel, thr)
    return ws.scores.copy(), ws.flags.copy(), ws.meta.copy()


@pytest.mark.parametrize("R,K,S", [(1, 0, 1), (1, 3, 0), (2, 2, 2), (8, 0, 64), (8, 5, 6), (8, 4096, 8), (64, 17, 33),
                                   (100, 7, 9), (3, 0, 0), (16, 13000, 40), (65, 0, 64), (1024, 0, 64), (4096, 32, 16)])
def test_score_kernel_matches_oracle(be, R, K, S):
    rng = np.random.default_rng(R * 1000 + K + S)
    T = _random_table(rng, R, K, S)
    for do_indiv, do_rel in ((True, True), (True, False), (False, True)):
        got, flags, meta = _score(be, T, K, S, do_indiv, do_rel, thr=(0.8, 0.7, 0.9, 0.6))
        exp = oracle.score_table(T, K, S, do_indiv, do_rel)
        assert got.shape == exp.shape
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        # GPU scores: f64 sums in a different order than the oracle's serial loop
        assert np.allclose(got[ok], exp[ok], rtol=2e-6, atol=0), np.abs(got[ok] - exp[ok]).max()
        # section scores are a single f64 division rounded to f32: bit-exact
        assert np.array_equal(got[:, 2:][~np.isnan(exp[:, 2:])], exp[:, 2:][~np.isnan(exp[:, 2:])])
        # flags: the ORACLE's scores against the thresholds (strict <, NaN never flagged: reporting.py:84-151).  Section
        # columns are bit-exact, so their flags must be too; a GPU score (columns 0-1, 2e-6 apart from the oracle's serial
        # sum) that sits within that distance of its threshold is the one case left out of the comparison
        thr_cols = np.concatenate([[0.9, 0.8], np.full(S, 0.6), np.full(S, 0.7)])
        with np.errstate(invalid="ignore"):
            exp_flags = (exp.astype(np.float64) < thr_cols[None, :]).astype(np.uint8)
            decided = ~(np.abs(exp.astype(np.float64) - thr_cols[None, :]) <= 4e-6 * thr_cols[None, :])
        decided[:, 2:] = True
        assert np.array_equal(flags[decided], exp_flags[decided])
        assert decided.mean() > 0.999
        assert list(meta[:4]) == [1, R, K, S]


def test_names_flag_is_reduced_over_ranks(be):
    rng = np.random.default_rng(3)
    T = _random_table(rng, 8, 2, 3)
    T[5, -1] = 0.0
    _, _, meta = _score(be, T, 2, 3)
    assert meta[0] == 0


def test_all_ranks_nan_when_a_rank_lacks_everything(be):
    """Any rank with no kernels => relative GPU score NaN everywhere (test_relative_gpu_scores.py:313-317)."""
    rng = np.random.default_rng(4)
    T = _random_table(rng, 4, 3, 0, p_missing=0.0)
    T[2, :3] = -1.0
    got, flags, _ = _score(be, T, 3, 0, False, True)
    assert np.isnan(got[:, 1]).all() and not flags[:, 1].any()


def _summ(d):
    from nvrx_straggler import Statistic as S

    key = {"MIN": S.MIN, "MAX": S.MAX, "MED": S.MED, "AVG": S.AVG, "STD": S.STD, "NUM": S.NUM}
    return {n: {key[k]: v for k, v in s.items()} for n, s in d.items()}


def test_report_generator_dict_path_single_rank_history(be):
    """ReportGenerator.generate_report(dicts) on one rank vs the reference's outputs
    (scoring.json / indiv_history_1rank): individual-score history, missing and new kernels."""
    from nvrx_straggler.reporting import ReportGenerator

    g = [s for s in load_golden("scoring.json")["scenarios"] if s["scenario"]["name"] == "indiv_history_1rank"][0]
    sc = g["scenario"]
    gen = ReportGenerator(sc["scores_to_compute"], gather_on_rank0=sc["gather_on_rank0"], node_name="node0")
    for t, step in enumerate(sc["steps"]):
        sec, ker = step[0]
        rep = gen.generate_report(_summ(sec), _summ(ker))
        exp = g["per_rank"][0]["reports"][t]
        assert close(rep.gpu_individual_perf_scores[0], exp["gpu_individual_perf_scores"]["0"], rel=1e-6), t
        for n, v in exp["section_individual_perf_scores"].items():
            assert close(rep.section_individual_perf_scores[n][0], v["0"], rel=1e-6)
        assert not rep.gpu_relative_perf_scores and not rep.section_relative_perf_scores
        assert rep.rank_to_node == {0: "node0"}
    # individual-only without gather must not touch the shared name mapper (test_name_mapper.py:101-123)
    assert gen.name_mapper.kernel_counter == 0 and gen.name_mapper.section_counter == 0


def test_report_generator_exact_known_answers(be):
    """Known answers of the reference's own unit tests, single rank: relative score of the only rank
    is 1 (test_relative_gpu_scores.py:46-62)."""
    from nvrx_straggler import Statistic as S
    from nvrx_straggler.reporting import ReportGenerator

    def summary(t):
        t = np.asarray(t, dtype=np.float64)
        return {S.MIN: t.min(), S.MAX: t.max(), S.MED: float(np.median(t)), S.AVG: t.mean(), S.STD: t.std(), S.NUM: t.size}

    gen = ReportGenerator(["relative_perf_scores"], gather_on_rank0=False, node_name="testnode")
    rep = gen.generate_report({}, kernel_summaries={"kernel0": summary([1.0, 1.0, 2.0]), "ncclDevKernel_x": summary([5.0])})
    assert rep.gpu_relative_perf_scores[0] == pytest.approx(1.0)
    assert "ncclDevKernel_x" not in rep.local_kernel_summaries
    rep = gen.generate_report({}, kernel_summaries={"kernel0": summary(1.25 * np.array([1.0, 1.0, 2.0]))})
    assert rep.gpu_relative_perf_scores[0] == pytest.approx(1.0)
    assert rep.identify_stragglers()["straggler_gpus_relative"] == set()


@pytest.mark.parametrize("R,K,S", [(8, 0, 64), (8, 5, 64), (64, 0, 64)])
def test_completion_word_never_precedes_the_results(be, R, K, S):
    """k_score1 publishes the sequence word after draining its write-through stores (no system-scope fence): the host
    must never see the new sequence number next to an older report's scores / flags.  Two tables with different
    answers alternate through ONE result block, checked the instant the word arrives, 20 000 times."""
    rng = np.random.default_rng(5)
    tabs = [_random_table(rng, R, K, S, p_missing=0.0) for _ in range(2)]
    tabs[1][:, : K + S] *= 3.0  # every median differs -> every individual score differs
    exp = [oracle.score_table(t, K, S, True, True) for t in tabs]
    dev = [torch.from_numpy(t).cuda() for t in tabs]
    ws = be.workspace(R, K, S, R, 0)
    torch.cuda.synchronize()
    W = 2 + 2 * S
    for i in range(20_000):
        be.score(ws, dev[i & 1], True, True, (0.75, 0.75, 0.75, 0.75))
        got = ws.scores
        e = exp[i & 1]
        # individual section scores: hmin / med, bit-exact and different between the two tables
        assert np.array_equal(got[:, 2 : 2 + S], e[:, 2 : 2 + S]), i
        assert np.array_equal(got[:, 2 + S : W], e[:, 2 + S : W]), i
        assert ws.meta[4] == ws.seq and ws.meta[1] == R
