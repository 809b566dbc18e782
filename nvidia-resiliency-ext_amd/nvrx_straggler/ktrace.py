"""Per-kernel GPU timing with real kernel names: the CUPTI-activity equivalent on MI355X.

``NVRX_GPU_TIMING=kernels`` selects this profiler instead of the per-region device timestamps of
``hip_profiler``.  It keeps the interface of the reference's native module (``CuptiProfiler`` with
``initialize / shutdown / start / stop / get_stats / reset``, cupti_src/cupti_module_py.cpp:33-55) and its
data model: while a profiled section is open every kernel the process launches is recorded under the key
``<mangled name>_blk_x_y_z_grid_x_y_z`` with its duration in microseconds (CuptiProfiler.cpp:186-191), so
the rank's GPU score is the kernel-weighted mean of reporting.py:219-253 over real kernels, and RCCL's
``ncclDev*`` kernels are left out exactly as in the reference (reporting.py:330-336).

Underneath, ``libnvrx_ktrace.so`` (include/nvrx_ktrace.h) is a rocprofiler-sdk tool: dispatch records are
collected on the SDK's thread, ``harvest()`` drains them and appends each kernel's durations to its DEVICE
ring row, and the statistics (mean-of-middles median, population stddev; CuptiProfiler.cpp:44-74) are
computed by the same HIP kernel as every other row.

The SDK only accepts tools before the HIP runtime initialises.  Importing ``nvrx_straggler`` registers the tool
right away (``setup``: ``rocprofiler_force_configure`` with the SDK's tool search kept off the large libraries) when
the mode is ``kernels`` -- named by ``NVRX_GPU_TIMING=kernels``, or chosen for the processes of a multi-rank job
(``timing_mode``); if HIP was initialised earlier the profiler raises with the advice to import the package first.
"""
from __future__ import annotations

import ctypes
import os
import threading
import warnings
import weakref
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_uint32, c_uint64
from typing import Dict, Optional

import numpy as np

from . import _native
from . import backend as _backend_mod
from .hip_profiler import KernelStats

_LIB_NAME = "libnvrx_ktrace.so"
# NVRX_LIB_DIR: load the native libraries from another directory (the sanitizer build of `make -C csrc asan` lives in
# lib_asan/; tools/run_sanitized.sh points here)
_LIB_PATH = os.path.join(os.environ.get("NVRX_LIB_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib"), _LIB_NAME)


class Record(Structure):
    _fields_ = [("key", c_uint32), ("us", c_float)]


RECORD_DTYPE = np.dtype([("key", np.uint32), ("us", np.float32)])
_ROW_UNKNOWN = -2  # key id without an entry in the key -> row table yet (-1: the rings had no row left for it)

# every symbol include/nvrx_ktrace.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("nvrx_ktrace_setup", c_int, [c_int]),
    ("nvrx_ktrace_hidden_libraries", c_int, []),
    ("nvrx_ktrace_ready", c_int, []),
    ("nvrx_ktrace_start", c_int, []),
    ("nvrx_ktrace_stop", c_int, []),
    ("nvrx_ktrace_flush", c_int, []),
    ("nvrx_ktrace_drain", c_int, [POINTER(Record), c_int]),
    ("nvrx_ktrace_pending", c_int, []),
    ("nvrx_ktrace_dropped", c_uint64, []),
    ("nvrx_ktrace_num_keys", c_int, []),
    ("nvrx_ktrace_key_name", c_char_p, [c_uint32]),
    ("nvrx_ktrace_reset", c_int, []),
    ("nvrx_ktrace_last_error", c_char_p, []),
]

_lib = None
_lock = threading.Lock()
_setup_error: Optional[str] = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load libnvrx_ktrace.so (once); raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise RuntimeError(f"{_LIB_NAME} not found at {_LIB_PATH}; build it with `make -C nvidia-resiliency-ext_amd/csrc`")
            lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
            for name, restype, argtypes in SYMBOLS:
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def _check(rc: int) -> int:
    if rc < 0:
        msg = load().nvrx_ktrace_last_error()
        raise RuntimeError(f"nvrx_ktrace error {rc}: {msg.decode() if msg else '?'}")
    return rc


def setup(max_pending: int = 0) -> None:
    """Register the tool with rocprofiler-sdk.  Call before the first HIP call of the process.

    Default: ``rocprofiler_force_configure`` right now, with the SDK's tool search kept away from the large libraries of
    the process (``nvrx_ktrace.cpp``, "tool discovery guard").  What rounds 1-3 knew as the SDK's start-up stall is that
    search: the SDK ELF-parses EVERY loaded shared library for a ``rocprofiler_configure`` symbol and its parser reads
    each file front to back -- 10.7 GB of ``read()`` calls in a PyTorch process, 3 s from a warm page cache, 70 s in the
    build container, 130-165 s on a GPU box with cold storage (``tools/archive/debug/readtrace.c`` has the backtrace:
    ``rocprofiler_set_api_table -> ... -> std::istream::read``).  Handing our tool over explicitly while the big libraries
    are hidden from that one search costs 0.1 GB of reads and 0.05 s (``profiles/r04c_ktrace_start_up.txt``).

    ``NVRX_KTRACE_FORCE=0`` takes the SDK's own route instead -- the library named in ``ROCP_TOOL_LIBRARIES``, loaded
    when the HIP runtime registers with the SDK (what ``rocprofv3`` does for its tool) -- and pays the full search."""
    global _setup_error
    if os.environ.get("NVRX_KTRACE_FORCE", "1") != "0":
        try:
            _check(load().nvrx_ktrace_setup(int(max_pending)))
            _setup_error = None
        except RuntimeError as e:
            _setup_error = str(e)
            raise
        return
    libs = [p for p in os.environ.get("ROCP_TOOL_LIBRARIES", "").split(":") if p]
    if _LIB_PATH not in libs:
        if not os.path.exists(_LIB_PATH):
            _setup_error = f"{_LIB_NAME} not found at {_LIB_PATH}; build it with `make -C nvidia-resiliency-ext_amd/csrc`"
            raise RuntimeError(_setup_error)
        os.environ["ROCP_TOOL_LIBRARIES"] = ":".join(libs + [_LIB_PATH])
    _setup_error = None


_mode: Optional[str] = None
_mode_note: str = ""


def _hip_is_up() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_initialized())
    except Exception:  # noqa: BLE001
        return False


# what launchers export for "processes in this job": torchrun / torch.distributed.run and whoever follows it, then srun,
# Open MPI's mpirun, and PMI-based launchers (MPICH, Intel MPI, Cray) -- a DDP script started by those reads its rank
# from the same variables before it calls init_process_group
_JOB_SIZE_VARS = ("WORLD_SIZE", "SLURM_NTASKS", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE")


def _job_size() -> tuple:
    """(processes in this job as far as the environment tells, the variable that said so)."""
    for var in _JOB_SIZE_VARS:
        raw = os.environ.get(var, "").strip()
        if not raw:
            continue
        try:
            n = int(raw)
        except ValueError:
            continue
        if var == "WORLD_SIZE" or n > 1:
            return n, var
    return 1, ""


def timing_mode() -> str:
    """How ``profile_cuda=True`` sections measure GPU time in this process: ``stamp`` | ``event`` | ``kernels``.

    ``NVRX_GPU_TIMING`` names it outright.  Unset (or ``auto``) it is decided ONCE, the first time anybody asks -- at
    ``import nvrx_straggler``:

    * a process of a multi-rank job (``WORLD_SIZE`` > 1 in the environment, what ``torchrun`` and every launcher that
      follows its convention exports; without it srun's, mpirun's or a PMI launcher's job size) gets ``kernels`` -- the reference's data model (CuptiProfiler.cpp:168-207): the GPU
      score is the kernel-weighted mean over real kernels and RCCL's ``ncclDev*`` kernels, whose duration is peer-wait
      time, are left out (reporting.py:330-336).  One row per REGION cannot do that: a step that ends in a collective
      lasts as long as the slowest rank's on every rank and a slow GPU scores 1.0 (tests/test_host_logic.py,
      ``test_region_timing_flattens_gpu_scores...``; tests/test_gpu_multiproc.py runs it on real kernels);
    * a single-process job, or a process whose HIP runtime is already up (rocprofiler-sdk accepts tools only before
      that), gets ``stamp``.

    Registration is cheap now (``setup``); if it fails the mode falls back to ``stamp`` and ``mode_note()`` says why."""
    global _mode, _mode_note
    if _mode is not None:
        return _mode
    with _lock:
        if _mode is not None:
            return _mode
        want = os.environ.get("NVRX_GPU_TIMING", "").strip().lower() or "auto"
        if want in ("stamp", "event"):
            _mode = want
        elif want == "kernels":
            _mode = "kernels"  # asked for by name: errors of the registration surface when the profiler is built
        else:
            world, said_by = _job_size()
            if world <= 1:
                _mode, _mode_note = "stamp", "single-process job"
            elif not os.path.exists("/dev/kfd"):
                _mode, _mode_note = "stamp", "no AMD GPU driver node (/dev/kfd) in this process' view"
            elif _hip_is_up():
                _mode, _mode_note = "stamp", ("multi-rank job, but the HIP runtime was initialised before nvrx_straggler was "
                                              "imported: rocprofiler-sdk accepts tools only before that")
            else:
                _mode, _mode_note = "kernels", f"multi-rank job ({said_by}={world})"
    if _mode == "kernels":
        try:
            setup()
        except Exception as e:  # noqa: BLE001
            if want == "auto":
                with _lock:
                    _mode, _mode_note = "stamp", f"per-kernel tracing could not be registered ({e})"
    return _mode


def mode_note() -> str:
    timing_mode()
    return _mode_note


def _reset_mode_for_tests() -> None:
    global _mode, _mode_note
    _mode, _mode_note = None, ""


def setup_from_env() -> None:
    """Import-time hook: settle the timing mode, which registers the tracer when the mode is ``kernels`` (it has to
    happen before anything touches HIP; errors of an explicitly requested mode surface at first use)."""
    timing_mode()


def drain_all() -> np.ndarray:
    """Flush, then pop every pending record: structured array with fields ``key`` (u32) and ``us`` (f32)."""
    lib = load()
    _check(lib.nvrx_ktrace_flush())
    chunks = []
    cap = 1 << 16
    buf = (Record * cap)()
    while True:
        n = _check(lib.nvrx_ktrace_drain(buf, cap))
        if n == 0:
            break
        chunks.append(np.frombuffer(buf, dtype=RECORD_DTYPE, count=n).copy())
        if n < cap:
            break
    return np.concatenate(chunks) if chunks else np.empty(0, dtype=RECORD_DTYPE)


def key_name(key: int) -> str:
    name = load().nvrx_ktrace_key_name(int(key))
    return name.decode() if name else f"unknown_key_{key}"


class KernelTraceProfiler:
    """rocprofiler-sdk stand-in for the CUPTI profiler object; one live instance per process."""

    _live: Optional["weakref.ReferenceType[KernelTraceProfiler]"] = None

    def __init__(self, bufferSize: int = 1024 * 1024 * 8, numBuffers: int = 8, statsMaxLenPerKernel: int = 1024,
                 rings=None, max_keys: int = 4096):
        live = KernelTraceProfiler._live() if KernelTraceProfiler._live is not None else None
        if live is not None and not live._closed:
            raise RuntimeError("Only one CuptiProfiler instance is allowed.")
        self._lib = load()
        if _setup_error is not None and not self._lib.nvrx_ktrace_ready():
            raise RuntimeError(_setup_error)
        if not self._lib.nvrx_ktrace_ready():
            setup()  # no-op when the import hook already ran; too late if HIP is already up (checked in initialize)
        self._owns_rings = rings is None
        if rings is None:
            rings = _backend_mod.get_backend().make_rings(1, int(max_keys), int(statsMaxLenPerKernel))
        self._rings = rings
        self._initialized = False
        self._started = False
        self._closed = False
        self._key_rows: Dict[int, int] = {}  # tracer key id -> ring row (-1: no row left)
        self._row_table = np.full(1024, _ROW_UNKNOWN, dtype=np.int32)  # the same as an array indexed by key id
        self.keys_without_row = 0
        KernelTraceProfiler._live = weakref.ref(self)

    # ---- lifecycle -----------------------------------------------------------------------------
    def _ensure_ready(self) -> None:
        if not self._lib.nvrx_ktrace_ready():
            import torch

            torch.cuda.init()  # the SDK calls the tool's initialiser when the runtime comes up
            if not self._lib.nvrx_ktrace_ready():
                raise RuntimeError(
                    "kernel tracing did not come up: rocprofiler-sdk was configured before nvrx_ktrace registered. "
                    "Import nvrx_straggler with NVRX_GPU_TIMING=kernels before the first HIP call, or set "
                    f"ROCP_TOOL_LIBRARIES={_LIB_PATH}"
                )

    def initialize(self) -> None:
        self._ensure_ready()
        self._initialized = True

    def shutdown(self) -> None:
        if self._started:
            self.stop()
        self._initialized = False

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            if self._owns_rings:
                self._rings.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- tracing window ----------------------------------------------------------------------------
    def start(self, key: str = "") -> None:
        """Trace every kernel launched from now on (reference: cuptiActivityEnable); ``key`` is unused --
        kernels name themselves."""
        if self._started:
            return
        _check(self._lib.nvrx_ktrace_start())
        self._started = True

    def stop(self, cpu_row: int = -1, cpu_value: float = 0.0) -> bool:
        if not self._started:
            return False
        _check(self._lib.nvrx_ktrace_stop())
        self._started = False
        return False  # the section's wall-time sample is the caller's to push

    # ---- results -----------------------------------------------------------------------------------
    def harvest(self, wait: bool = True) -> int:
        """Move the recorded kernel durations into their device ring rows.  ``wait``: let the device finish
        first, as the reference does before it reads its statistics (straggler.py:234)."""
        if wait:
            import torch

            torch.cuda.synchronize()
        return self.ingest(drain_all())

    def ingest(self, recs: np.ndarray) -> int:
        """Append drained ``(key, us)`` records (``RECORD_DTYPE``, arrival order) to the device rings: every key id is
        looked up in a key -> ring-row table (rows are handed out the first time a key shows up: cold) and ALL records
        go to the device with one ``nvrx_ring_push_pairs`` call = one scatter launch, whether they belong to two
        kernel keys or to four thousand (data_shared test sizes of the reference: tests/straggler/unit/test_data_shared.py:62-66)."""
        if recs.size == 0:
            return 0
        keys = recs["key"]
        table = self._row_table
        top = int(keys.max())
        if top >= table.size:
            grown = np.full(max(top + 1, 2 * table.size), _ROW_UNKNOWN, dtype=np.int32)
            grown[: table.size] = table
            table = self._row_table = grown
        rows = table[keys]
        if (rows == _ROW_UNKNOWN).any():
            rings = self._rings
            for k in np.unique(keys[rows == _ROW_UNKNOWN]).tolist():
                try:
                    row = rings.row_for(_native.KIND_KERNEL, key_name(k))
                except RuntimeError:
                    row = -1
                    self.keys_without_row += 1
                    if self.keys_without_row == 1:
                        warnings.warn("straggler rings are full: further kernel names are not recorded "
                                      "(raise max_rows in Detector.initialize)")
                table[k] = row
                self._key_rows[k] = row
            rows = table[keys]
        self._rings.push_pairs(rows, recs["us"])
        return int(recs.size)

    def active_rows(self) -> Dict[str, int]:
        r = self._rings
        return {k: row for k, row in r.kernel_row_names.items() if r.count(row) > 0}

    def get_stats(self) -> Dict[str, KernelStats]:
        self.harvest(wait=True)
        rows = self.active_rows()
        out: Dict[str, KernelStats] = {}
        if not rows:
            return out
        stats = self._rings.peek_stats()
        for key, row in rows.items():
            v = stats[row]
            ks = KernelStats()
            ks.min, ks.max, ks.median, ks.avg, ks.stddev = (float(v[i]) for i in range(5))
            ks.num_calls = int(v[5])
            out[key] = ks
        return out

    def reset(self) -> None:
        _check(self._lib.nvrx_ktrace_reset())
        for row in self._rings.kernel_row_names.values():
            self._rings.set_count(row, 0)

    @property
    def dropped(self) -> int:
        return int(self._lib.nvrx_ktrace_dropped())
