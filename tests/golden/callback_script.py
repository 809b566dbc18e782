"""Scripted drive of a ``StragglerDetectionCallback`` (reference or ours): the SAME fake trainer, the same sequence of
synthetic reports, one transcript of everything observable -- logger records, ``log_dict`` payloads, the stop flag,
checkpoint calls, ``sys.exit``.  ``make_golden.py`` runs it against the reference's callback
(ptl_resiliency/straggler_det_callback.py:37-265) and freezes the transcripts in ``callback.json``;
``tests/test_callback_golden.py`` runs it against this package's callback and compares.

No reference import happens here: the callback class, its ``straggler`` module and the ``Report`` class are passed in.
"""
import logging
import re

#: callback configurations (constructor kwargs) x report scripts
CONFIGS = {
    "print2_log_stop": dict(report_time_interval=1.0, calc_relative_gpu_perf=True, calc_individual_gpu_perf=True,
                            num_gpu_perf_scores_to_print=2, gpu_relative_perf_threshold=0.7,
                            gpu_individual_perf_threshold=0.7, stop_if_detected=True, enable_ptl_logging=True),
    "quiet_rel_only": dict(report_time_interval=5.0, calc_relative_gpu_perf=True, calc_individual_gpu_perf=False,
                           num_gpu_perf_scores_to_print=0, gpu_relative_perf_threshold=0.9,
                           gpu_individual_perf_threshold=0.5, stop_if_detected=False, enable_ptl_logging=False,
                           profiling_interval=3),
    "print3_indiv_only_log": dict(report_time_interval=2.5, calc_relative_gpu_perf=False, calc_individual_gpu_perf=True,
                                  num_gpu_perf_scores_to_print=3, gpu_relative_perf_threshold=0.7,
                                  gpu_individual_perf_threshold=0.8, stop_if_detected=False, enable_ptl_logging=True),
}


def _scores(n, low=None, value=0.61):
    """rank -> score for n ranks: a spread of healthy values, ``low`` ranks at ``value``."""
    out = {r: round(0.9 + 0.013 * ((r * 7) % 8), 6) for r in range(n)}
    for r in low or ():
        out[r] = value
    return out


def reports(n_ranks):
    """The scripted iterations: ``None`` = no report this iteration (interval not elapsed), else the report's fields."""
    nodes = {r: f"node{r // 4}" for r in range(n_ranks)}
    base = dict(rank_to_node=nodes, local_section_summaries={}, local_kernel_summaries={}, section_relative_perf_scores={},
                section_individual_perf_scores={}, generate_report_elapsed_time=1.5, gather_on_rank0=True, rank=0)
    return [
        None,
        dict(base, gpu_relative_perf_scores=_scores(n_ranks), gpu_individual_perf_scores=_scores(n_ranks)),
        None,
        dict(base, gpu_relative_perf_scores=_scores(n_ranks, low=[3 % n_ranks]), gpu_individual_perf_scores=_scores(n_ranks)),
        dict(base, gpu_relative_perf_scores=_scores(n_ranks), gpu_individual_perf_scores=_scores(n_ranks, low=[1 % n_ranks, 6 % n_ranks], value=0.75)),
        dict(base, gpu_relative_perf_scores=_scores(n_ranks, low=[0, 5 % n_ranks], value=0.3),
             gpu_individual_perf_scores=_scores(n_ranks, low=[5 % n_ranks], value=0.2)),
        dict(base, gpu_relative_perf_scores={}, gpu_individual_perf_scores={}),   # empty mappings: NaN min / median / max
    ]


SCENARIOS = [
    # (name, config, ranks, global_rank, checkpoint callback?, async checkpoint io?, log_dict raises?)
    ("rank0_8ranks", "print2_log_stop", 8, 0, False, False, False),
    ("rank0_3ranks_prints_all", "print2_log_stop", 3, 0, False, False, False),
    ("rank0_checkpoint_exit", "print2_log_stop", 8, 0, True, True, False),
    ("rank0_checkpoint_sync_io", "print2_log_stop", 8, 0, True, False, False),
    ("rank1_sees_no_report", "print2_log_stop", 8, 1, False, False, False),
    ("quiet", "quiet_rel_only", 8, 0, False, False, False),
    ("indiv_only", "print3_indiv_only_log", 8, 0, False, False, False),
    ("log_dict_fails", "print3_indiv_only_log", 8, 0, False, False, True),
]

_ELAPSED = re.compile(r"(Straggler report processing time: )[0-9.]+( sec\.)")
_IDS = re.compile(r"\{(StragglerId\([^{}]*\))\}")


def normalise(message: str) -> str:
    """Wall-clock figures out, set members in a stable order (a set of dataclasses prints in hash order)."""
    message = _ELAPSED.sub(r"\g<1>T\g<2>", message)

    def order(m):
        items = re.findall(r"StragglerId\(rank=\d+, node='[^']*'\)", m.group(0))
        return "{" + ", ".join(sorted(items, key=lambda s: int(re.search(r"rank=(\d+)", s).group(1)))) + "}"

    return _IDS.sub(order, message)


def drive(callback_cls, straggler_module, report_cls, scenario, patch_gather=False):
    """Run one scenario; returns the transcript (a JSON-able dict)."""
    name, config, n_ranks, global_rank, with_ckpt, async_io, log_fails = scenario
    script = reports(n_ranks)
    calls = {"initialize": [], "wrap": [], "shutdown": 0}
    state = {"i": -1}
    det = straggler_module.Detector
    saved = {k: det.__dict__.get(k) for k in ("generate_report_if_interval_elapsed", "is_interval_elapsed", "initialize",
                                               "wrap_callables", "shutdown")}

    def gen():
        state["i"] += 1
        spec = script[state["i"]]
        if spec is None or global_rank != 0:
            return None   # gather_on_rank0: the other ranks never hold a report
        return report_cls(**spec)

    det.generate_report_if_interval_elapsed = staticmethod(gen)
    det.is_interval_elapsed = staticmethod(lambda: script[state["i"]] is not None)
    det.initialize = staticmethod(lambda **kw: calls["initialize"].append({k: (list(v) if isinstance(v, (list, tuple)) else v)
                                                                           for k, v in kw.items()}))
    det.wrap_callables = staticmethod(lambda callable_ids, **kw: calls["wrap"].append(
        [[type(c.obj).__name__, c.name] for c in callable_ids]))
    det.shutdown = staticmethod(lambda: calls.__setitem__("shutdown", calls["shutdown"] + 1))

    records = []

    class Grab(logging.Handler):
        def emit(self, record):
            records.append([record.levelname, normalise(record.getMessage())])

    logger_name = f"golden.callback.{name}"
    log = logging.getLogger(logger_name)
    log.setLevel(logging.DEBUG)
    log.propagate = False
    handler = Grab()
    log.addHandler(handler)

    ckpt_calls = []

    class Ckpt:
        def _monitor_candidates(self, trainer):
            ckpt_calls.append("monitor_candidates")
            return {"step": 7}

        def _save_last_checkpoint(self, trainer, candidates):
            ckpt_calls.append(["save_last_checkpoint", dict(candidates)])

    class SyncIO:
        pass

    class AsyncIO:
        def maybe_finalize_save_checkpoint(self, blocking=False):
            ckpt_calls.append(["maybe_finalize_save_checkpoint", blocking])

    class Strategy:
        checkpoint_io = AsyncIO() if async_io else SyncIO()

        def training_step(self, batch):
            return batch

    class Trainer:
        def __init__(self):
            self.strategy = Strategy()
            self.global_rank = global_rank
            self.should_stop = False
            self.checkpoint_callback = Ckpt() if with_ckpt else None

    logged = []

    class Module:
        def log_dict(self, payload, **kw):
            if log_fails:
                raise RuntimeError("logger backend is down")
            logged.append([{k: (None if v != v else v) for k, v in payload.items()}, dict(sorted(kw.items()))])

    out = {"scenario": name, "config": config, "iterations": []}
    try:
        cb = callback_cls(logger_name=logger_name, **CONFIGS[config])
        if patch_gather:
            setattr(cb, "_gather_flag_from_rank0", lambda flag: bool(flag))   # the reference's needs a CUDA device (:231-238)
        trainer, module = Trainer(), Module()
        cb.setup(trainer, module, "fit")
        cb.setup(trainer, module, "fit")   # second call must be a no-op
        exit_code = None
        for it in range(len(script)):
            n_rec, n_log = len(records), len(logged)
            try:
                cb.on_train_batch_end(trainer, module, None, None, it)
            except SystemExit as e:
                exit_code = e.code
            out["iterations"].append({"records": records[n_rec:], "log_dict": logged[n_log:],
                                      "should_stop": trainer.should_stop, "exit": exit_code})
            if exit_code is not None:
                break
        cb.teardown(trainer, module, "fit")
        cb.teardown(trainer, module, "fit")
        out["initialize_calls"] = calls["initialize"]
        out["wrap_calls"] = calls["wrap"]
        out["shutdown_calls"] = calls["shutdown"]
        out["checkpoint_calls"] = ckpt_calls
        out["scores_to_compute"] = list(cb.scores_to_compute)
    finally:
        log.removeHandler(handler)
        for k, v in saved.items():
            if v is None:
                try:
                    delattr(det, k)
                except AttributeError:
                    pass
            else:
                setattr(det, k, v)
    return out


def constructor_error(callback_cls):
    """Neither score family requested: the message of the ValueError (straggler_det_callback.py:92-96)."""
    try:
        callback_cls(report_time_interval=1.0, calc_relative_gpu_perf=False, calc_individual_gpu_perf=False,
                     num_gpu_perf_scores_to_print=1, gpu_relative_perf_threshold=0.7, gpu_individual_perf_threshold=0.7,
                     stop_if_detected=False, enable_ptl_logging=False)
    except ValueError as e:
        return str(e)
    return None
