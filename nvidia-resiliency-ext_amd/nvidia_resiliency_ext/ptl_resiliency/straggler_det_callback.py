"""``StragglerDetectionCallback``: PyTorch-Lightning hook-up of the MI355X straggler detector.

Constructor arguments, hooks (``setup / teardown / on_train_batch_end``), log texts and the
stop-on-straggler behaviour follow the reference callback (ptl_resiliency/straggler_det_callback.py:
37-265).  Differences: Lightning is optional -- if neither ``lightning`` nor ``pytorch_lightning`` is
installed the callback derives from ``object`` and still works with any trainer exposing the same
hooks -- and the rank-0 "stragglers found" broadcast travels on the process group's own device.
"""
from __future__ import annotations

import importlib.util
import logging
import sys
import time
from typing import Optional

import torch

import nvrx_straggler as straggler
from nvrx_straggler import dist_utils


def _callback_base():
    for mod in ("lightning.pytorch.callbacks", "pytorch_lightning.callbacks"):
        try:
            if importlib.util.find_spec(mod.split(".")[0]) is not None:
                return importlib.import_module(mod).Callback
        except (ImportError, ValueError):
            continue
    return object


Callback = _callback_base()


class StragglerDetectionCallback(Callback):
    def __init__(
        self,
        report_time_interval: float,
        calc_relative_gpu_perf: bool,
        calc_individual_gpu_perf: bool,
        num_gpu_perf_scores_to_print: int,
        gpu_relative_perf_threshold: float,
        gpu_individual_perf_threshold: float,
        stop_if_detected: bool,
        enable_ptl_logging: bool,
        profiling_interval: int = 1,
        logger_name: Optional[str] = "nemo_logger.StragglerDetectionCallback",
    ):
        """
        Args:
            report_time_interval: seconds between straggler checks.
            calc_relative_gpu_perf / calc_individual_gpu_perf: which score families to compute.
            num_gpu_perf_scores_to_print: how many best and worst ranks to print each report
                (0: print only when stragglers are found).
            gpu_relative_perf_threshold / gpu_individual_perf_threshold: flagging thresholds.
            stop_if_detected: stop training (after a final checkpoint) when stragglers are found.
            enable_ptl_logging: log min/median/max GPU scores through ``pl_module.log_dict``.
            profiling_interval: forwarded to ``Detector.initialize``.
            logger_name: name of the ``logging`` logger to use.

        Raises:
            ValueError: neither score family requested.
        """
        self.initialized = False
        self.logger = logging.getLogger(logger_name)
        self.report_time_interval = report_time_interval
        self.calc_relative_gpu_perf = calc_relative_gpu_perf
        self.calc_individual_gpu_perf = calc_individual_gpu_perf
        self.num_gpu_perf_scores_to_print = num_gpu_perf_scores_to_print
        self.gpu_relative_perf_threshold = gpu_relative_perf_threshold
        self.gpu_individual_perf_threshold = gpu_individual_perf_threshold
        self.stop_if_detected = stop_if_detected
        self.enable_ptl_logging = enable_ptl_logging
        self.profiling_interval = profiling_interval
        self.scores_to_compute = []
        if calc_relative_gpu_perf:
            self.scores_to_compute.append("relative_perf_scores")
        if calc_individual_gpu_perf:
            self.scores_to_compute.append("individual_perf_scores")
        if not self.scores_to_compute:
            raise ValueError(
                "No straggler performance scores specified. "
                "Check if calc_relative_gpu_perf=True or calc_individual_gpu_perf=True"
            )
        self.interval_est_was_reset = False

    # ---- Lightning hooks -----------------------------------------------------------------------
    def setup(self, trainer, pl_module, stage):
        if self.initialized:
            return
        straggler.Detector.initialize(
            scores_to_compute=self.scores_to_compute,
            gather_on_rank0=True,
            profiling_interval=self.profiling_interval,
            report_time_interval=self.report_time_interval,
        )
        step_owner = trainer.strategy
        assert getattr(step_owner, "training_step", None), f"{type(step_owner)} does not have 'training_step' method."
        # every training step becomes a profiled section named "<Strategy class>.training_step"
        straggler.Detector.wrap_callables(callable_ids=[straggler.CallableId(step_owner, "training_step")])
        self.initialized = True

    def teardown(self, trainer, pl_module, stage):
        if self.initialized:
            straggler.Detector.shutdown()
            self.initialized = False

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx):
        started = time.monotonic()
        detector = straggler.Detector
        report = detector.generate_report_if_interval_elapsed()
        # gather_on_rank0: rank 0 alone holds the report and decides; the decision reaches the others below
        found = bool(report) and trainer.global_rank == 0 and self._digest(pl_module, report)
        if not detector.is_interval_elapsed():
            return  # no report was due this iteration
        if self.stop_if_detected and self._decision_of_rank0(found):
            self._halt(trainer)
        self.logger.info(f"Straggler report processing time: {time.monotonic() - started:.3f} sec.")

    # ---- what happens with a report (rank 0) -----------------------------------------------------------
    #: (constructor switch, Report field, identify_stragglers key, heading, logging prefix, warning text)
    _FAMILIES = (
        ("calc_relative_gpu_perf", "gpu_relative_perf_scores", "straggler_gpus_relative", "GPU relative performance",
         "gpu_relative_perf", "Some GPUs have worse relative performance."),
        ("calc_individual_gpu_perf", "gpu_individual_perf_scores", "straggler_gpus_individual", "GPU individual performance",
         "gpu_individual_perf", "Some GPUs performance dropped."),
    )

    def _digest(self, pl_module, report) -> bool:
        """Warn about flagged GPUs, print the best / worst ranks, feed the PTL loggers; True if anything was flagged."""
        flagged = report.identify_stragglers(
            gpu_rel_threshold=self.gpu_relative_perf_threshold,
            gpu_indiv_threshold=self.gpu_individual_perf_threshold,
        )
        hits = [(text, flagged[key]) for _, _, key, _, _, text in self._FAMILIES if flagged[key]]
        for text, ranks in hits:
            self.logger.warning(f"STRAGGLER DETECTION WARNING: {text} Affected ranks: {ranks}")
        if hits:
            self._telemetry_of_a_flagged_reporter(ranks for _, ranks in hits)
        enabled = [f for f in self._FAMILIES if getattr(self, f[0])]
        n = self.num_gpu_perf_scores_to_print
        if n > 0:
            for _, field, _, heading, _, _ in enabled:
                self.logger.info(f"\n{heading}:\n{self._ranking(getattr(report, field), report.rank_to_node, n, n)}")
        if self.enable_ptl_logging:
            for _, field, _, _, prefix, _ in enabled:
                self._log_extremes(pl_module, getattr(report, field), prefix)
        return bool(hits)

    def _telemetry_of_a_flagged_reporter(self, groups) -> None:
        # MI355X extra: when the reporting rank itself is flagged, say what ROCm SMI sees on its GPU (clock below
        # peak, hot junction, power) -- the first things to rule out for a slow GPU
        det = straggler.Detector
        me = getattr(det.reporter, "rank", None) if det.initialized else None
        if me is not None and any(getattr(s, "rank", None) == me for group in groups for s in group):
            self.logger.warning(f"rank {me}: {det.gpu_telemetry_line()}")

    @staticmethod
    def _ranking(rank_to_score, rank_to_node, num_best=3, num_worst=3) -> str:
        """Worst ``num_worst`` and best ``num_best`` ranks (all of them if there are no more than that), one
        ``Rank= Node= Score=`` line each; ties go by rank, as sorting (score, rank) pairs gives them."""
        ascending = sorted((score, rank) for rank, score in rank_to_score.items())
        total = len(ascending)

        def lines(pairs):
            return "".join(f"  Rank={rank} Node={rank_to_node[rank]} Score={score:.2f}\n" for score, rank in pairs)

        if total <= num_best + num_worst:
            return lines(ascending)
        return (f" Worst performing {num_worst}/{total} ranks:\n" + lines(ascending[:num_worst])
                + f" Best performing {num_best}/{total} ranks:\n" + lines(reversed(ascending[-num_best:])))

    # the reference's name for the same text (a static helper some users call directly)
    _format_gpu_scores = _ranking

    def _log_extremes(self, pl_module, rank_to_score, prefix) -> None:
        """min / median / max of one score family to every PTL logger; NaN when the family is empty."""
        stats = dict.fromkeys(("min", "median", "max"), float("nan"))
        if rank_to_score:
            t = torch.tensor(list(rank_to_score.values()), dtype=torch.float32)
            stats = {"min": torch.min(t).item(), "median": torch.median(t).item(), "max": torch.max(t).item()}
        try:
            pl_module.log_dict({f"{prefix}/{k}": v for k, v in stats.items()}, logger=True, batch_size=1, rank_zero_only=True)
        except Exception as e:  # logging must never take training down
            self.logger.error(f"Failed to log GPU performance scores: {e}")

    # ---- stopping the job -----------------------------------------------------------------------------
    def _decision_of_rank0(self, flag) -> bool:
        """Rank 0's verdict on every rank (a broadcast on the process group's own device; no group: the flag)."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=dist_utils.get_device_for_backend(None))
        torch.distributed.broadcast(t, 0)
        return bool(t.item() > 0)

    def _halt(self, trainer) -> None:
        self.logger.error("Detected stragglers. Terminating training...")
        trainer.should_stop = True
        checkpointing = trainer.checkpoint_callback
        if not checkpointing:
            return
        # a last checkpoint, an asynchronous one awaited, then out
        checkpointing._save_last_checkpoint(trainer, checkpointing._monitor_candidates(trainer))
        finalize = getattr(trainer.strategy.checkpoint_io, "maybe_finalize_save_checkpoint", None)
        if finalize is not None:
            self.logger.info("Async checkpointing detected, waiting for it to complete...")
            finalize(blocking=True)
        sys.exit(1)
