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
    def _wrap_ptl_callables(self, trainer):
        assert getattr(trainer.strategy, "training_step", None), (
            f"{type(trainer.strategy)} does not have 'training_step' method."
        )
        straggler.Detector.wrap_callables(callable_ids=[straggler.CallableId(trainer.strategy, "training_step")])

    def setup(self, trainer, pl_module, stage):
        if self.initialized:
            return
        straggler.Detector.initialize(
            scores_to_compute=self.scores_to_compute,
            gather_on_rank0=True,
            profiling_interval=self.profiling_interval,
            report_time_interval=self.report_time_interval,
        )
        self._wrap_ptl_callables(trainer)
        self.initialized = True

    def teardown(self, trainer, pl_module, stage):
        if self.initialized:
            straggler.Detector.shutdown()
            self.initialized = False

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx):
        t0 = time.monotonic()
        report = straggler.Detector.generate_report_if_interval_elapsed()
        found = False
        if trainer.global_rank == 0 and report:
            found = self._handle_straggler_report(pl_module, report)
        if straggler.Detector.is_interval_elapsed():  # a report was produced this iteration
            if self.stop_if_detected and self._gather_flag_from_rank0(found):
                self._stop_training(trainer)
            self.logger.info(f"Straggler report processing time: {time.monotonic() - t0:.3f} sec.")

    # ---- report handling -------------------------------------------------------------------------
    def _print_stragglers(self, stragglers):
        rel = stragglers["straggler_gpus_relative"]
        if rel:
            self.logger.warning(
                f"STRAGGLER DETECTION WARNING: Some GPUs have worse relative performance. Affected ranks: {rel}"
            )
        indiv = stragglers["straggler_gpus_individual"]
        if indiv:
            self.logger.warning(
                f"STRAGGLER DETECTION WARNING: Some GPUs performance dropped. Affected ranks: {indiv}"
            )
        # MI355X extra: when the reporting rank itself is flagged, say what ROCm SMI sees on its GPU
        # (clock below peak, hot junction, power) -- the first things to rule out for a slow GPU
        me = getattr(straggler.Detector.reporter, "rank", None) if straggler.Detector.initialized else None
        if me is not None and any(getattr(s, "rank", None) == me for s in list(rel) + list(indiv)):
            self.logger.warning(f"rank {me}: {straggler.Detector.gpu_telemetry_line()}")

    @staticmethod
    def _format_gpu_scores(rank_to_score, rank_to_node, num_best=3, num_worst=3) -> str:
        ordered = sorted(((s, r) for r, s in rank_to_score.items()), reverse=True)  # best first
        n = len(ordered)

        def line(s, r):
            return f"  Rank={r} Node={rank_to_node[r]} Score={s:.2f}\n"

        if n <= num_best + num_worst:
            return "".join(line(s, r) for s, r in reversed(ordered))
        out = f" Worst performing {num_worst}/{n} ranks:\n"
        out += "".join(line(s, r) for s, r in reversed(ordered[-num_worst:]))
        out += f" Best performing {num_best}/{n} ranks:\n"
        out += "".join(line(s, r) for s, r in ordered[:num_best])
        return out

    def _print_gpu_scores(self, report):
        assert self.num_gpu_perf_scores_to_print > 0
        n = self.num_gpu_perf_scores_to_print
        if self.calc_relative_gpu_perf:
            text = self._format_gpu_scores(report.gpu_relative_perf_scores, report.rank_to_node, n, n)
            self.logger.info(f"\nGPU relative performance:\n{text}")
        if self.calc_individual_gpu_perf:
            text = self._format_gpu_scores(report.gpu_individual_perf_scores, report.rank_to_node, n, n)
            self.logger.info(f"\nGPU individual performance:\n{text}")

    def _log_gpu_perf_scores(self, pl_module, rank_to_score, rank_to_node, score_prefix):
        lo = med = hi = float("nan")
        values = list(rank_to_score.values())
        if values:
            t = torch.tensor(values, dtype=torch.float32)
            lo, med, hi = torch.min(t).item(), torch.median(t).item(), torch.max(t).item()
        payload = {f"{score_prefix}/min": lo, f"{score_prefix}/median": med, f"{score_prefix}/max": hi}
        try:
            pl_module.log_dict(payload, logger=True, batch_size=1, rank_zero_only=True)
        except Exception as e:  # logging must never take training down
            self.logger.error(f"Failed to log GPU performance scores: {e}")

    def _log_gpu_scores(self, pl_module, report):
        assert self.enable_ptl_logging is True
        if self.calc_relative_gpu_perf:
            self._log_gpu_perf_scores(pl_module, report.gpu_relative_perf_scores, report.rank_to_node, "gpu_relative_perf")
        if self.calc_individual_gpu_perf:
            self._log_gpu_perf_scores(pl_module, report.gpu_individual_perf_scores, report.rank_to_node, "gpu_individual_perf")

    def _handle_straggler_report(self, pl_module, report) -> bool:
        stragglers = report.identify_stragglers(
            gpu_rel_threshold=self.gpu_relative_perf_threshold,
            gpu_indiv_threshold=self.gpu_individual_perf_threshold,
        )
        found = bool(stragglers["straggler_gpus_relative"] or stragglers["straggler_gpus_individual"])
        if found:
            self._print_stragglers(stragglers)
        if self.num_gpu_perf_scores_to_print > 0:
            self._print_gpu_scores(report)
        if self.enable_ptl_logging:
            self._log_gpu_scores(pl_module, report)
        return found

    def _gather_flag_from_rank0(self, flag) -> bool:
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=dist_utils.get_device_for_backend(None))
        torch.distributed.broadcast(t, 0)
        return bool(t.item() > 0)

    def _stop_training(self, trainer) -> None:
        self.logger.error("Detected stragglers. Terminating training...")
        trainer.should_stop = True
        ckpt = trainer.checkpoint_callback
        if ckpt:
            ckpt._save_last_checkpoint(trainer, ckpt._monitor_candidates(trainer))
            io = trainer.strategy.checkpoint_io
            if hasattr(io, "maybe_finalize_save_checkpoint"):
                self.logger.info("Async checkpointing detected, waiting for it to complete...")
                io.maybe_finalize_save_checkpoint(blocking=True)
            sys.exit(1)
