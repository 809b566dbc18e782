"""Per-kernel GPU timing with real kernel names: the CUPTI-activity equivalent on MI355X.

``NVRX_GPU_TIMING=kernels`` (the default of a multi-rank job) selects this profiler instead of the per-region device
timestamps of ``hip_profiler``.  It keeps the interface of the reference's native module (``CuptiProfiler`` with
``initialize / shutdown / start / stop / get_stats / reset``, cupti_src/cupti_module_py.cpp:33-55) and its data
model: while a profiled section is open every kernel the process launches is recorded under the key
``<mangled name>_blk_x_y_z_grid_x_y_z`` with its duration in microseconds (CuptiProfiler.cpp:186-191), so the
rank's GPU score is the kernel-weighted mean of reporting.py:219-253 over real kernels, and RCCL's ``ncclDev*``
kernels are left out exactly as in the reference (reporting.py:330-336).

The data path is native end to end (``libnvrx_ktrace.so``, include/nvrx_ktrace.h): the rocprofiler-sdk callback
thread turns each batch of dispatch records into (ring row, microseconds) pairs and appends them to the DEVICE rings
through a sink of two function pointers into ``libnvrx_straggler_hip.so`` -- per key an overwrite-oldest ring, the
newest ``ring_cap`` durations survive, as in the reference (CuptiProfiler.cpp:168-207, CircularBuffer.h:53-61).
Python sees no record.  At report time the training thread calls ``harvest()``: ONE C call (``nvrx_ktrace_sync``:
every kernel enqueued in a section so far has finished and is in the rings -- the role of
``torch.cuda.synchronize()`` in straggler.py:234 without waiting for the rest of the device), plus, when kernel keys
nobody has seen before turned up, their names (cold).  The statistics (mean-of-middles median, population stddev;
CuptiProfiler.cpp:44-74) are computed by the same HIP kernel as every other row.

The SDK only accepts tools before the HIP runtime initialises.  In a job launched the ``torchrun`` way (``WORLD_SIZE`` in
the environment, more than one rank) -- or with ``NVRX_GPU_TIMING=kernels`` -- importing ``nvrx_straggler`` registers the
tool right away (``setup``); everywhere else the import touches nothing and the mode is settled at
``Detector.initialize`` (``setup_from_env``).  If HIP was initialised earlier, ``auto`` falls back to region stamps and an
explicit ``kernels`` raises with the advice to import the package first.
"""
from __future__ import annotations

import ctypes
import logging
import os
import threading
import warnings
import weakref
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_uint32, c_uint64, c_void_p
from typing import Dict, Optional

import numpy as np

from . import _native
from . import backend as _backend_mod
from .hip_profiler import KernelStats

_log = logging.getLogger(__name__)
_LIB_NAME = "libnvrx_ktrace.so"
# NVRX_DEBUG_LIB_DIR: load the native libraries from another directory (the sanitizer build of `make -C csrc asan` lives in
# lib_asan/; tools/run_sanitized.sh points here)
_LIB_PATH = os.path.join(os.environ.get("NVRX_DEBUG_LIB_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib"), _LIB_NAME)
ERR_UNSAFE = -16


class Record(Structure):
    _fields_ = [("key", c_uint32), ("us", c_float)]


RECORD_DTYPE = np.dtype([("key", np.uint32), ("us", np.float32)])


class Sink(Structure):
    """``nvrx_ktrace_sink``: where the tracer's thread appends the durations."""

    _fields_ = [("ctx", c_void_p), ("push", c_void_p), ("row_alloc", c_void_p), ("kind", c_int32)]


class Dispatch(Structure):
    """``nvrx_ktrace_dispatch``: one kernel execution as the feed entry takes it."""

    _fields_ = [("kernel_id", c_uint64), ("workgroup", c_uint32 * 3), ("grid", c_uint32 * 3), ("start_ns", c_uint64), ("end_ns", c_uint64)]


DISPATCH_DTYPE = np.dtype([("kernel_id", np.uint64), ("workgroup", np.uint32, (3,)), ("grid", np.uint32, (3,)),
                           ("start_ns", np.uint64), ("end_ns", np.uint64)])
assert DISPATCH_DTYPE.itemsize == ctypes.sizeof(Dispatch) == 48

# every symbol include/nvrx_ktrace.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("nvrx_ktrace_setup", c_int, [c_int]),
    ("nvrx_ktrace_set_max_pending", c_int, [c_int]),
    ("nvrx_ktrace_release_env", c_int, []),
    ("nvrx_ktrace_hidden_libraries", c_int, []),
    ("nvrx_ktrace_ready", c_int, []),
    ("nvrx_ktrace_set_sink", c_int, [POINTER(Sink)]),
    ("nvrx_ktrace_hold", c_int, [c_int]),
    ("nvrx_ktrace_include_blits", c_int, [c_int]),
    ("nvrx_ktrace_tap", c_int, [c_int]),
    ("nvrx_ktrace_start", c_int, []),
    ("nvrx_ktrace_stop", c_int, []),
    ("nvrx_ktrace_sync", c_int, [c_double]),
    ("nvrx_ktrace_forgive", c_int, []),
    ("nvrx_ktrace_flush", c_int, []),
    ("nvrx_ktrace_drain", c_int, [POINTER(Record), c_int]),
    ("nvrx_ktrace_pending", c_int, []),
    ("nvrx_ktrace_dropped", c_uint64, []),
    ("nvrx_ktrace_counter", c_uint64, [c_int]),
    ("nvrx_ktrace_num_keys", c_int, []),
    ("nvrx_ktrace_key_name", c_char_p, [c_uint32]),
    ("nvrx_ktrace_key_row", c_int, [c_uint32]),
    ("nvrx_ktrace_reset", c_int, []),
    ("nvrx_ktrace_last_error", c_char_p, []),
    ("nvrx_ktrace_feed_kernel_name", c_int, [c_uint64, c_char_p, c_int]),
    ("nvrx_ktrace_feed", c_int, [c_void_p, c_int, c_int]),
]
COUNTERS = ("enqueued", "arrived", "delivered", "lost_no_row", "sink_errors", "own_skipped", "keys_without_row", "forgiven",
            "pump_flushes", "counting", "rows_assigned", "blit_skipped", "by_callback")

_lib = None
_lock = threading.Lock()
_setup_error: Optional[str] = None
_setup_route: str = ""  # how the tool was handed to the SDK ("force_configure" | "ROCP_TOOL_LIBRARIES")


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
            lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)  # (loads librocprofiler-sdk, whose start-up code sets GLOG_* variables)
            for name, restype, argtypes in SYMBOLS:
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


_GLOG_NAMES = ("GLOG_minloglevel", "GLOG_logtostderr", "GLOG_alsologtostderr", "GLOG_stderrthreshold", "GLOG_v")


def release_env() -> None:
    """Once this process' HIP runtime is up: take back what rocprofiler-sdk wrote into the process environment --
    ``ROCPROFILER_REGISTER_FORCE_LOAD=1`` (``nvrx_ktrace_release_env``; a child that inherits it loads and configures the SDK
    on ``import torch``: the full tool search, and no tool can register there any more) and the ``GLOG_*`` switches its
    library sets when it is loaded (those the job itself had not set).  ``unsetenv`` is not safe against a ``getenv``
    running in another thread at that very moment (glibc); it is done ONCE, right after the runtime has come up -- the SDK
    wrote the variables under the same conditions -- and the alternative is every child process misbehaving."""
    load().nvrx_ktrace_release_env()
    libc = ctypes.CDLL(None)
    for name in _GLOG_NAMES:
        if name not in os.environ:  # (os.environ is the environment this interpreter STARTED with, plus the job's own changes)
            libc.unsetenv(name.encode())
    if _added_to_tool_libraries:
        # the SDK's own route (taken when the link-map guard refused): the library has been loaded by now; a child that
        # inherited the variable would run the SDK's full tool search at ITS first HIP call whether it traces or not
        libs = [p for p in os.environ.get("ROCP_TOOL_LIBRARIES", "").split(":") if p and p != _LIB_PATH]
        if libs:
            os.environ["ROCP_TOOL_LIBRARIES"] = ":".join(libs)
        else:
            os.environ.pop("ROCP_TOOL_LIBRARIES", None)


def _release_env_when_hip_is_up() -> None:
    """What registering leaves in the environment is taken back the moment this process' HIP runtime has come up (PyTorch runs
    the queued call at the end of its lazy CUDA initialisation): before the job has built a communicator or started worker
    threads, and also when no profiler is ever initialised (a rank that imports the package and never uses the detector used to
    hand ``ROCPROFILER_REGISTER_FORCE_LOAD=1`` to every child it started).  The profiler's ``initialize`` does the same again,
    which is then a no-op."""
    def _quietly():
        try:
            release_env()
        except Exception as e:  # noqa: BLE001  (never fail somebody's first CUDA call over housekeeping)
            _log.debug("nvrx straggler: release_env failed: %s", e)

    try:
        import torch

        torch.cuda._lazy_call(_quietly)
    except Exception:  # noqa: BLE001  (a PyTorch without the hook: the profiler's initialize() releases)
        pass


def _check(rc: int) -> int:
    if rc < 0:
        msg = load().nvrx_ktrace_last_error()
        raise RuntimeError(f"nvrx_ktrace error {rc}: {msg.decode() if msg else '?'}")
    return rc


def counters() -> Dict[str, int]:
    """The tracer's process-wide counters (include/nvrx_ktrace.h, ``nvrx_ktrace_counter``)."""
    lib = load()
    return {name: int(lib.nvrx_ktrace_counter(i)) for i, name in enumerate(COUNTERS)}


_added_to_tool_libraries = False


def _name_in_tool_libraries() -> None:
    global _added_to_tool_libraries
    libs = [p for p in os.environ.get("ROCP_TOOL_LIBRARIES", "").split(":") if p]
    if _LIB_PATH not in libs:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_NAME} not found at {_LIB_PATH}; build it with `make -C nvidia-resiliency-ext_amd/csrc`")
        os.environ["ROCP_TOOL_LIBRARIES"] = ":".join(libs + [_LIB_PATH])
        _added_to_tool_libraries = True


def setup(max_pending: int = 0) -> None:
    """Register the tool with rocprofiler-sdk.  Call before the first HIP call of the process.

    Default: ``rocprofiler_force_configure`` right now, with the SDK's tool search kept away from the large libraries of
    the process (``nvrx_ktrace.cpp``, "tool discovery guard").  What rounds 1-3 knew as the SDK's start-up stall is that
    search: the SDK ELF-parses EVERY loaded shared library for a ``rocprofiler_configure`` symbol and its parser reads
    each file front to back -- 10.7 GB of ``read()`` calls in a PyTorch process, 3 s from a warm page cache, 130-165 s on a
    GPU box with cold storage.  Handing our tool over explicitly while the big libraries are hidden from that one search
    costs 0.1 GB of reads and 0.05 s (``profiles/r04c_ktrace_start_up.txt``).  The guard touches loader state, so the
    library only applies it while every other thread of the process is asleep (the situation at import time: BLAS pool
    workers parked on a futex); when it refuses (``NVRX_KTRACE_ERR_UNSAFE``) the SDK's own route is taken instead.

    ``NVRX_DEBUG_KTRACE_FORCE=0`` takes the SDK's own route outright -- the library named in ``ROCP_TOOL_LIBRARIES``, loaded
    when the HIP runtime registers with the SDK (what ``rocprofv3`` does for its tool) -- and pays the full search."""
    global _setup_error, _setup_route
    try:
        if os.environ.get("NVRX_DEBUG_KTRACE_FORCE", "1") != "0":
            rc = load().nvrx_ktrace_setup(int(max_pending))
            if rc != ERR_UNSAFE:
                _check(rc)
                _setup_error, _setup_route = None, "force_configure"
                _release_env_when_hip_is_up()
                return
            msg = load().nvrx_ktrace_last_error()
            if _hip_is_up():  # (naming the library for the HIP runtime's start-up is pointless once it has started)
                raise RuntimeError("per-kernel tracing has to be registered before the HIP runtime starts: import nvrx_straggler "
                                   "(in a multi-rank job, or with NVRX_GPU_TIMING=kernels) before the first HIP call")
            _log.info("nvrx straggler: %s -- registering through ROCP_TOOL_LIBRARIES instead (the SDK searches every loaded "
                      "library for tools when HIP starts: seconds to minutes)", msg.decode() if msg else "tool-search guard refused")
        _name_in_tool_libraries()
        _setup_error, _setup_route = None, "ROCP_TOOL_LIBRARIES"
        _release_env_when_hip_is_up()
    except RuntimeError as e:
        _setup_error = str(e)
        raise


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


def _foreign_tool() -> str:
    """Another rocprofiler-sdk tool that is in charge of this process (``rocprofv3 -- python ...`` names its tool library in
    ROCP_TOOL_LIBRARIES and preloads the SDK), or ''.  Two tools can coexist in the SDK, but a profiling run is the user's
    measurement: the tracer stays out of it unless asked for by name."""
    libs = [p for p in os.environ.get("ROCP_TOOL_LIBRARIES", "").split(":") if p and os.path.basename(p) != _LIB_NAME]
    return libs[0] if libs else ""


def timing_mode() -> str:
    """How ``profile_cuda=True`` sections measure GPU time in this process: ``stamp`` | ``event`` | ``kernels``.

    ``NVRX_GPU_TIMING`` names it outright.  Unset (or ``auto``) it is decided ONCE, the first time anybody asks -- at
    ``import nvrx_straggler`` when ``WORLD_SIZE`` is in the environment, else at ``Detector.initialize`` (``setup_from_env``):

    * a process of a multi-rank job (``WORLD_SIZE`` > 1 in the environment, what ``torchrun`` and every launcher that
      follows its convention exports; without it srun's, mpirun's or a PMI launcher's job size) gets ``kernels`` -- the reference's data model (CuptiProfiler.cpp:168-207): the GPU
      score is the kernel-weighted mean over real kernels and RCCL's ``ncclDev*`` kernels, whose duration is peer-wait
      time, are left out (reporting.py:330-336).  One row per REGION cannot do that: a step that ends in a collective
      lasts as long as the slowest rank's on every rank and a slow GPU scores 1.0 (tests/test_host_logic.py,
      ``test_region_timing_flattens_gpu_scores...``; tests/test_gpu_multiproc.py runs it on real kernels);
    * a single-process job, a process whose HIP runtime is already up (rocprofiler-sdk accepts tools only before
      that), or one that runs under another rocprofiler-sdk tool (``rocprofv3``), gets ``stamp``.

    If the registration fails the mode falls back to ``stamp`` and ``mode_note()`` says why.  The ranks of a job agree on
    ONE mode at their first collective report (``Detector``: a rank that could not trace kernels pulls everybody to
    ``stamp``, with a warning -- mixed modes share no kernel names and every relative GPU score would be NaN)."""
    global _mode, _mode_note
    if _mode is not None:
        return _mode
    with _lock:
        if _mode is not None:
            return _mode
        want = os.environ.get("NVRX_GPU_TIMING", "").strip().lower() or "auto"
        if want in ("stamp", "event"):
            _mode, _mode_note = want, f"NVRX_GPU_TIMING={want}"
        elif want == "kernels":
            _mode, _mode_note = "kernels", "NVRX_GPU_TIMING=kernels"  # asked for by name: errors of the registration surface when the profiler is built
        else:
            world, said_by = _job_size()
            foreign = _foreign_tool()
            if world <= 1:
                _mode, _mode_note = "stamp", "single-process job"
            elif not os.path.exists("/dev/kfd"):
                _mode, _mode_note = "stamp", "no AMD GPU driver node (/dev/kfd) in this process' view"
            elif _hip_is_up():
                _mode, _mode_note = "stamp", ("multi-rank job, but the HIP runtime was initialised before nvrx_straggler was "
                                              "imported: rocprofiler-sdk accepts tools only before that")
            elif foreign:
                _mode, _mode_note = "stamp", f"multi-rank job under another rocprofiler-sdk tool (ROCP_TOOL_LIBRARIES={foreign})"
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


def mode_code() -> int:
    """What the ranks MIN-reduce to agree on a mode: 1 = per-kernel keys, 0 = one row per region (stamp / event)."""
    return 1 if timing_mode() == "kernels" else 0


def fall_back_to_regions(why: str) -> None:
    """The job's common mode is per-region timing (some rank cannot trace kernels): this process follows."""
    global _mode, _mode_note
    with _lock:
        _mode, _mode_note = "stamp", why


def _reset_mode_for_tests() -> None:
    global _mode, _mode_note
    _mode, _mode_note = None, ""


def setup_from_env() -> None:
    """Import-time hook.  rocprofiler-sdk accepts tools only BEFORE the process' HIP runtime starts, so a multi-rank job that
    is to trace kernels by name has to register early; but an import should not reconfigure a process that never uses the
    detector.  The rule:

    * ``WORLD_SIZE`` is in the environment (``torchrun`` and every launcher that follows its convention), or
      ``NVRX_GPU_TIMING=kernels`` asks for the mode by name: the mode is settled NOW -- for a job of more than one rank that
      registers the tracer (``timing_mode``); scripts typically select their GPU right after the imports, and by
      ``Detector.initialize`` it would be too late;
    * otherwise (single process; ``srun`` / ``mpirun`` / PMI launchers without ``WORLD_SIZE``) NOTHING is touched by the
      import: the mode is settled at ``Detector.initialize``.  Such a job gets per-kernel tracing if the detector is
      initialised before the first HIP call (as the reference's own example does, examples/straggler/example.py:60-66), region
      stamps otherwise -- or set ``NVRX_KTRACE_AT_IMPORT=1``.

    ``NVRX_KTRACE_AT_IMPORT=0`` defers in every case, ``=1`` settles at import in every case."""
    when = os.environ.get("NVRX_KTRACE_AT_IMPORT", "")
    if when == "0":
        return
    if when == "1" or "WORLD_SIZE" in os.environ or os.environ.get("NVRX_GPU_TIMING", "").strip().lower() == "kernels":
        timing_mode()


def drain_all() -> np.ndarray:
    """Without a sink (no live profiler): flush, then pop every pending record -- structured array with fields ``key``
    (u32) and ``us`` (f32).  Diagnostics / C-ABI hosts that keep their own rings; the package itself never drains."""
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


def feed(dispatches: np.ndarray, counted: bool = True, through_inbox: bool = False) -> None:
    """Hand ``DISPATCH_DTYPE`` records to the tracer's data path (``nvrx_ktrace_feed``): consumed on the calling thread, as
    the tracer's own threads do with a batch -- or, ``through_inbox``, left in the inbox as the SDK's completion callback
    leaves them, for the pump thread / the next ``harvest`` to bring in.  Tests and the benchmark's feeder thread."""
    a = np.ascontiguousarray(dispatches, dtype=DISPATCH_DTYPE)
    _check(load().nvrx_ktrace_feed(a.ctypes.data, int(a.size), int(bool(counted)) | (2 if through_inbox else 0)))


def feed_kernel_name(kernel_id: int, name: str, own: bool = False) -> None:
    _check(load().nvrx_ktrace_feed_kernel_name(int(kernel_id), name.encode(), int(own)))


_sink_ctx: Optional[int] = None  # address of the ring context the tracer's thread currently appends to (None: no sink)


def detach_sink_of(ctx_address: Optional[int]) -> None:
    """Rings that are about to be destroyed call this: if the tracer's thread is appending to THEM, it lets go first
    (``nvrx_ktrace_set_sink(NULL)`` returns once a batch in progress is through).  No-op for any other rings."""
    global _sink_ctx
    if _lib is not None and ctx_address is not None and _sink_ctx == ctx_address:
        _lib.nvrx_ktrace_set_sink(None)
        _sink_ctx = None


_exit_hook_registered = False


def _register_exit_hook() -> None:
    """At interpreter exit (before the HIP runtime is torn down): stop tracing and take the sink off the rings, so that the
    SDK's thread does not append a late record to a context that is going away."""
    global _exit_hook_registered
    if _exit_hook_registered:
        return
    _exit_hook_registered = True
    import atexit

    def _quiesce():
        global _sink_ctx
        lib = _lib
        if lib is None:
            return
        try:
            if lib.nvrx_ktrace_ready():
                lib.nvrx_ktrace_stop()
            lib.nvrx_ktrace_set_sink(None)
            _sink_ctx = None
        except Exception:  # noqa: BLE001
            pass

    atexit.register(_quiesce)


def _sync_patience_s() -> float:
    """How long a report waits for the records of its window's kernels before it synchronises the device the reference's
    way (``NVRX_DEBUG_KTRACE_SYNC_PATIENCE_S``, read when the profiler is built: a report does not look at the environment)."""
    try:
        return float(os.environ.get("NVRX_DEBUG_KTRACE_SYNC_PATIENCE_S", "2.0"))
    except ValueError:
        return 2.0


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
        if not self._lib.nvrx_ktrace_ready() and timing_mode() == "kernels":
            setup()  # no-op when the import hook already ran; too late if HIP is already up (checked in initialize)
        self._owns_rings = rings is None
        if rings is None:
            rings = _backend_mod.get_backend().make_rings(1, int(max_keys), int(statsMaxLenPerKernel))
        self._rings = rings
        self._initialized = False
        self._started = False
        self._closed = False
        self._rows_known = 0  # keys of the tracer whose ring row (and name) this object has learnt
        self._keys_seen = 0  # key ids [0, _keys_seen) have been looked at ...
        self._key_pending: list = []  # ... and these of them have no row under this sink (yet)
        self.keys_without_row = 0
        self._warned_leak = False
        self._starts = 0            # traced section entries so far
        self._said_blind = False    # "no kernel launch seen" has been said (harvest)
        self._counting = True  # dispatches are counted at enqueue (read back in initialize)
        self.sync_patience_s = _sync_patience_s()
        # memsets / memcpys are not kernels to CUPTI; NVRX_KTRACE_BLITS=1 records ROCm's blit kernels all the same
        self._lib.nvrx_ktrace_include_blits(1 if os.environ.get("NVRX_KTRACE_BLITS", "0") == "1" else 0)
        # from now on the tracer's thread appends every kernel duration to these rings
        ctx, push, row_alloc = rings.ktrace_sink()
        global _sink_ctx
        self._sink = Sink(ctx, push, row_alloc, _native.KIND_KERNEL)
        _check(self._lib.nvrx_ktrace_set_sink(ctypes.byref(self._sink)))
        _sink_ctx = ctx
        _register_exit_hook()
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
        # (settled by the tool's initialiser, which has run by now; without the SDK -- records fed through nvrx_ktrace_feed --
        #  the feeder does the counting)
        self._counting = bool(self._lib.nvrx_ktrace_counter(9)) or not self._lib.nvrx_ktrace_ready()
        # the runtime is up: what rocprofiler_force_configure left in the environment (ROCPROFILER_REGISTER_FORCE_LOAD=1 ...) must
        # not reach the children this process starts from now on (DataLoader workers, spawned ranks)
        release_env()
        self._initialized = True

    def shutdown(self) -> None:
        if self._started:
            self.stop()
        self._initialized = False

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            # the tracer's thread lets go of the rings before they are destroyed (returns once a batch in progress is through)
            global _sink_ctx
            self._lib.nvrx_ktrace_set_sink(None)
            _sink_ctx = None
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
        self._starts += 1

    def stop(self, cpu_row: int = -1, cpu_value: float = 0.0) -> bool:
        if not self._started:
            return False
        _check(self._lib.nvrx_ktrace_stop())
        self._started = False
        return False  # the section's wall-time sample is the caller's to push

    # ---- results -----------------------------------------------------------------------------------
    def harvest(self, wait: bool = True) -> int:
        """Report time.  ``wait``: every kernel enqueued inside a section so far has finished and its duration is in the
        rings when this returns (straggler.py:234 synchronises the whole device for that); ``wait=False`` (asynchronous
        reports) only looks: what has not arrived yet counts in the next window.  One C call; names of kernel keys seen
        for the first time are fetched on top (cold).  Returns the number of dispatches still missing."""
        lib = self._lib
        if wait and not self._counting:
            # dispatches are not counted (NVRX_DEBUG_KTRACE_COUNT=0, or the SDK refused the ENQUEUE callback): nvrx_ktrace_sync can
            # only flush what HAS completed, so the device is waited for first, as the reference does (straggler.py:234)
            import torch

            torch.cuda.synchronize()
        missing = lib.nvrx_ktrace_sync(self.sync_patience_s if wait else 0.0)
        if missing < 0:
            _check(missing)
        if not self._said_blind and self._starts and self._counting:
            self._said_blind = True  # (looked at once, at the first report after a traced entry)
            if lib.nvrx_ktrace_counter(0) == 0 and _hip_is_up():
                # Traced sections have run and the tracer has not seen ONE kernel launch in this process.  Sections without GPU
                # work look like that -- and so does a tracer that registered after the HIP runtime had started (a C-level first
                # HIP call torch.cuda.is_initialized() did not know about at registration): rocprofiler-sdk then intercepts
                # nothing, every GPU score is NaN, and nothing else would say so.
                _log.warning("nvrx straggler: per-kernel GPU timing has seen no kernel launch in %d traced section entr%s. If these "
                             "sections launch GPU work, the tracer was registered after the HIP runtime started (import nvrx_straggler "
                             "before the first HIP call, or use NVRX_GPU_TIMING=stamp): GPU scores stay NaN until then.",
                             self._starts, "y" if self._starts == 1 else "ies")
        if missing > 0 and wait:
            missing = self._wait_the_long_way(missing)
        if lib.nvrx_ktrace_counter(10) != self._rows_known or lib.nvrx_ktrace_counter(6) != self.keys_without_row:
            self._learn_keys()
        return missing

    def _wait_the_long_way(self, missing: int) -> int:
        """The kernels of the window are still running after ``sync_patience_s`` (a long step, a collective
        waiting for a straggler) -- or a dispatch was counted and its record never came.  Wait for the device as the
        reference does, look again, and if records are STILL missing stop expecting them."""
        import torch

        torch.cuda.synchronize()
        lib = self._lib
        missing = lib.nvrx_ktrace_sync(0.2)
        if missing > 0:
            _check(lib.nvrx_ktrace_flush())
            missing = lib.nvrx_ktrace_sync(0.0)
        if missing > 0:
            lib.nvrx_ktrace_forgive()
            if not self._warned_leak:
                self._warned_leak = True
                _log.warning("nvrx straggler: %d traced kernel dispatch(es) never produced a record; not waiting for them", missing)
        return 0 if missing < 0 else missing

    def _learn_keys(self) -> None:
        """Kernel keys the tracer's thread has met since the last look: their names and ring rows (cold path)."""
        lib, rings = self._lib, self._rings
        n = lib.nvrx_ktrace_num_keys()
        pending = self._key_pending
        pending.extend(range(self._keys_seen, n))  # key ids nobody has looked at yet
        self._keys_seen = n
        still = []
        for k in pending:  # (only the keys without a known row: a job with thousands of keys does not re-ask about all of them)
            row = lib.nvrx_ktrace_key_row(k)  # (-2: no record of this key has come under this sink yet; -1: no row was left)
            if row >= 0:
                rings.kernel_row_names[key_name(k)] = row
                self._rows_known += 1
            else:
                still.append(k)
        self._key_pending = still
        note = getattr(rings, "note_rows_used", None)
        if note is not None:
            note()  # the rows the tracer's thread took for these keys now count as used (reports cover them)
        lost = int(lib.nvrx_ktrace_counter(6))
        if lost > self.keys_without_row:
            if self.keys_without_row == 0:
                warnings.warn("straggler rings are full: further kernel names are not recorded "
                              "(raise max_rows in Detector.initialize)")
            self.keys_without_row = lost

    def hold(self, on: bool) -> None:
        """Bracket of an asynchronous report's "statistics launch + ring reset": durations that arrive meanwhile wait on the
        tracer's thread and land in the next window (``nvrx_ktrace_hold``)."""
        self._lib.nvrx_ktrace_hold(int(on))

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
        """Forget every recorded duration (reference: flush + map.clear(), CuptiProfiler.cpp:148-152): what the kernels
        enqueued so far have produced is waited for first -- a record that arrived after the reset would bring its key back."""
        self.harvest(wait=True)  # (also learns the rows of keys nobody has asked about yet)
        _check(self._lib.nvrx_ktrace_reset())
        for row in self._rings.kernel_row_names.values():
            self._rings.set_count(row, 0)

    @property
    def dropped(self) -> int:
        """Durations that did not make it into a ring: their key found no row left (rings full)."""
        return int(self._lib.nvrx_ktrace_counter(3)) + int(self._lib.nvrx_ktrace_dropped())
