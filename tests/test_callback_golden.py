"""A21: ``StragglerDetectionCallback`` against the REFERENCE callback's own output (tests/golden/callback.json, made by
``make_golden.py callback`` from ptl_resiliency/straggler_det_callback.py:37-265 under the scripted trainer of
``callback_script.py``): logger records (level + text), ``log_dict`` payloads and keyword arguments, the stop flag, the
checkpoint calls and ``sys.exit`` of the stop path, ``Detector.initialize`` / ``wrap_callables`` / ``shutdown`` calls.
The reports are scripted (plain ``Report`` records), so no device is needed here; the 2-process GPU test
(test_gpu_multiproc.py) checks the lines a real run logs against the same golden's line shapes."""
import json
import os

import pytest

import callback_script

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "callback.json")


def _golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _ours():
    import nvrx_straggler
    from nvidia_resiliency_ext.ptl_resiliency import StragglerDetectionCallback
    from nvrx_straggler.reporting import Report

    return StragglerDetectionCallback, nvrx_straggler, Report


def test_constructor_error_text_is_the_reference_one():
    cls, _, _ = _ours()
    assert callback_script.constructor_error(cls) == _golden()["constructor_error"]


@pytest.mark.parametrize("index", range(len(callback_script.SCENARIOS)), ids=[s[0] for s in callback_script.SCENARIOS])
def test_callback_transcript_equals_the_reference_callbacks(index):
    cls, straggler, report_cls = _ours()
    want = _golden()["scenarios"][index]
    got = callback_script.drive(cls, straggler, report_cls, callback_script.SCENARIOS[index])
    assert got["scenario"] == want["scenario"]
    for key in ("initialize_calls", "wrap_calls", "shutdown_calls", "checkpoint_calls", "scores_to_compute"):
        assert got[key] == want[key], key
    assert len(got["iterations"]) == len(want["iterations"])
    for i, (g, w) in enumerate(zip(got["iterations"], want["iterations"])):
        assert g["records"] == w["records"], (i, g["records"], w["records"])
        assert g["should_stop"] == w["should_stop"] and g["exit"] == w["exit"], i
        assert len(g["log_dict"]) == len(w["log_dict"]), i
        for (gp, gk), (wp, wk) in zip(g["log_dict"], w["log_dict"]):
            assert gk == wk and gp.keys() == wp.keys(), i
            for k in wp:
                assert gp[k] == wp[k] or (gp[k] is not None and wp[k] is not None and abs(gp[k] - wp[k]) <= 1e-7), (i, k, gp[k], wp[k])


def test_the_golden_exercises_every_branch_of_the_callback():
    """The scripted scenarios reach: both straggler warnings, best/worst and print-all formatting, NaN statistics of an
    empty mapping, the stop path with and without a checkpoint callback / asynchronous checkpoint io, a rank that holds
    no report, a failing ``log_dict``."""
    text = json.dumps(_golden())
    for needle in ("worse relative performance", "performance dropped", "Worst performing 2/8", "Best performing 3/8",
                   "Terminating training", "Async checkpointing detected", "Failed to log GPU performance scores",
                   "processing time: T sec", "maybe_finalize_save_checkpoint"):
        assert needle in text, needle
    g = _golden()["scenarios"]
    assert any(it["exit"] == 1 for sc in g for it in sc["iterations"])
    assert any(p[k] is None for sc in g for it in sc["iterations"] for p, _ in it["log_dict"] for k in p)


def test_a_flagged_reporter_adds_one_telemetry_line_and_nothing_else_changes(caplog):
    """The one branch the reference callback does not have: with the ``Detector`` live and the REPORTING rank among the
    flagged ones, one extra warning carries what ROCm SMI sees on its GPU (here, without a GPU, the 'unavailable' text);
    a flagged set without the reporter, or a detector that is not initialised, logs exactly the reference's lines."""
    import logging

    from nvrx_straggler import Detector, backend
    from oracle_backend import OracleBackend

    cls, _, report_cls = _ours()
    cb = cls(**dict(callback_script.CONFIGS["print2_log_stop"], logger_name="test.straggler.telemetry"))
    specs = callback_script.reports(8)
    with_reporter, without_reporter = report_cls(**specs[5]), report_cls(**specs[3])   # ranks {0, 5} / {3} flagged

    class _Module:
        def log_dict(self, payload, **kw):
            pass

    def lines(report):
        caplog.clear()
        assert cb._digest(_Module(), report) is True
        return [r.getMessage() for r in caplog.records if r.levelno == logging.WARNING]

    caplog.set_level(logging.INFO, logger="test.straggler.telemetry")
    assert not Detector.initialized
    cold = lines(with_reporter)
    assert len(cold) == 2 and all(m.startswith("STRAGGLER DETECTION WARNING") for m in cold)
    backend.set_backend(OracleBackend())
    try:
        Detector.initialize(scores_to_compute=["relative_perf_scores", "individual_perf_scores"], gather_on_rank0=True)
        live = lines(with_reporter)
        assert live[:2] == cold and len(live) == 3
        assert live[2].startswith("rank 0: gpu telemetry")
        assert lines(without_reporter) == [m for m in lines(without_reporter) if m.startswith("STRAGGLER DETECTION WARNING")]
        assert len(lines(without_reporter)) == 1
    finally:
        Detector.shutdown()
        backend.set_backend(None)
