"""GPU telemetry through ROCm SMI: the context a straggler verdict needs on MI355X.

A low ``gpu_individual_perf_score`` says a GPU got slower than it used to be; the usual reasons are visible to
the system-management interface -- a clock that sits below its peak level, a hot junction, a power cap.  The
reference has no such hook (its native side is CUPTI only); BASELINE.json's north star names ``rocm_smi`` as the
system-side half of the profiling backend on ROCm, so ``sample()`` reads, for one device, the current and peak
shader / memory clocks, edge / junction / HBM temperatures, socket power and busy percentage straight from
``librocm_smi64.so`` (ctypes, no subprocess).  ``StragglerDetectionCallback`` appends it to its log line for
ranks it flags; ``Detector.gpu_telemetry()`` returns it to user code.  Nothing here is on the report's hot path
and every field is optional: a value the driver does not expose is left out.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Dict, Optional

_RSMI_MAX_NUM_FREQUENCIES = 33  # rocm_smi.h:75
_CLK_SYS, _CLK_MEM = 0x0, 0x4    # rsmi_clk_type_t (rocm_smi.h:357-363)
_TEMP_CURRENT = 0x0              # rsmi_temperature_metric_t
_TEMP_EDGE, _TEMP_JUNCTION, _TEMP_MEMORY = 0, 1, 2  # rsmi_temperature_type_t


class _Frequencies(ctypes.Structure):  # rsmi_frequencies_t (rocm_smi.h:776-801)
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * _RSMI_MAX_NUM_FREQUENCIES)]


_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()


def _load() -> ctypes.CDLL:
    global _lib
    with _lock:
        if _lib is None:
            last = None
            for path in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "librocm_smi64.so"), "librocm_smi64.so.1",
                         "librocm_smi64.so"):
                try:
                    lib = ctypes.CDLL(path)
                    break
                except OSError as e:
                    last = e
            else:
                raise RuntimeError(f"librocm_smi64.so could not be loaded: {last}")
            lib.rsmi_init.argtypes = [ctypes.c_uint64]
            lib.rsmi_num_monitor_devices.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
            lib.rsmi_dev_gpu_clk_freq_get.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(_Frequencies)]
            lib.rsmi_dev_temp_metric_get.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
            lib.rsmi_dev_current_socket_power_get.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
            lib.rsmi_dev_power_ave_get.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
            lib.rsmi_dev_busy_percent_get.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
            rc = lib.rsmi_init(0)
            if rc != 0:
                raise RuntimeError(f"rsmi_init failed with status {rc}")
            _lib = lib
    return _lib


def num_devices() -> int:
    n = ctypes.c_uint32(0)
    rc = _load().rsmi_num_monitor_devices(ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"rsmi_num_monitor_devices failed with status {rc}")
    return int(n.value)


def sample(device_index: int = 0) -> Dict[str, float]:
    """Telemetry of one device (ROCm SMI index; equals the HIP index unless ``HIP_VISIBLE_DEVICES`` reorders).

    Keys (present when the driver reports them): ``sclk_mhz``, ``sclk_peak_mhz``, ``mclk_mhz``, ``mclk_peak_mhz``,
    ``temp_edge_c``, ``temp_junction_c``, ``temp_hbm_c``, ``power_w``, ``busy_pct``, and ``sclk_frac`` =
    current / peak shader clock (the number to look at next to a low individual GPU score)."""
    lib = _load()
    dv = int(device_index)
    out: Dict[str, float] = {}
    for tag, clk in (("sclk", _CLK_SYS), ("mclk", _CLK_MEM)):
        f = _Frequencies()
        if lib.rsmi_dev_gpu_clk_freq_get(dv, clk, ctypes.byref(f)) == 0 and 0 < f.num_supported <= _RSMI_MAX_NUM_FREQUENCIES:
            levels = [f.frequency[i] for i in range(f.num_supported)]
            if f.current < f.num_supported:
                out[f"{tag}_mhz"] = levels[f.current] / 1e6
            out[f"{tag}_peak_mhz"] = max(levels) / 1e6
    if out.get("sclk_peak_mhz"):
        out["sclk_frac"] = out.get("sclk_mhz", 0.0) / out["sclk_peak_mhz"]
    for tag, sensor in (("edge", _TEMP_EDGE), ("junction", _TEMP_JUNCTION), ("hbm", _TEMP_MEMORY)):
        t = ctypes.c_int64(0)
        if lib.rsmi_dev_temp_metric_get(dv, sensor, _TEMP_CURRENT, ctypes.byref(t)) == 0:
            out[f"temp_{tag}_c"] = t.value / 1000.0  # millidegrees
    p = ctypes.c_uint64(0)
    if lib.rsmi_dev_current_socket_power_get(dv, ctypes.byref(p)) == 0 or lib.rsmi_dev_power_ave_get(dv, 0, ctypes.byref(p)) == 0:
        out["power_w"] = p.value / 1e6  # microwatts
    b = ctypes.c_uint32(0)
    if lib.rsmi_dev_busy_percent_get(dv, ctypes.byref(b)) == 0:
        out["busy_pct"] = float(b.value)
    return out


# rsmi_dev_perf_level_t (rocm_smi.h:164-184)
PERF_LEVEL_AUTO, PERF_LEVEL_LOW, PERF_LEVEL_HIGH, PERF_LEVEL_MANUAL = 0, 1, 2, 3
PERF_LEVEL_STABLE_STD, PERF_LEVEL_STABLE_PEAK, PERF_LEVEL_STABLE_MIN_MCLK, PERF_LEVEL_STABLE_MIN_SCLK = 4, 5, 6, 7
_RSMI_STATUS = {1: "invalid arguments", 2: "not supported by this device / driver", 3: "file error", 4: "permission denied (needs root and a "
                "writable sysfs)", 5: "out of resources", 6: "internal exception", 7: "input out of bounds", 8: "initialisation error",
                10: "busy", 0xFFFFFFFF: "refused by the driver (unknown error: read-only sysfs inside a container is the usual reason)"}


class SmiRefused(RuntimeError):
    """ROCm SMI would not change the device's performance state (no root, read-only sysfs, unsupported level)."""


def perf_level(device_index: int = 0) -> int:
    lib = _load()
    lvl = ctypes.c_int(0)
    lib.rsmi_dev_perf_level_get.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_int)]
    rc = lib.rsmi_dev_perf_level_get(int(device_index), ctypes.byref(lvl))
    if rc != 0:
        raise SmiRefused(f"rsmi_dev_perf_level_get: {_RSMI_STATUS.get(rc & 0xFFFFFFFF, rc)}")
    return int(lvl.value)


def set_perf_level(device_index: int, level: int) -> None:
    """``rsmi_dev_perf_level_set_v1``: the ROCm counterpart of ``nvidia-smi -lgc`` the reference's example suggests for
    slowing one GPU down (examples/straggler/example.py:20).  ``PERF_LEVEL_LOW`` / ``PERF_LEVEL_STABLE_MIN_SCLK`` pin
    the shader clock at its lowest level; ``PERF_LEVEL_AUTO`` gives the device back.  Raises ``SmiRefused`` when the
    driver says no (it needs root and a writable sysfs: bare metal or a privileged container)."""
    lib = _load()
    lib.rsmi_dev_perf_level_set_v1.argtypes = [ctypes.c_uint32, ctypes.c_int]
    rc = lib.rsmi_dev_perf_level_set_v1(int(device_index), int(level))
    if rc != 0:
        raise SmiRefused(f"rsmi_dev_perf_level_set_v1(device {device_index}, level {level}): {_RSMI_STATUS.get(rc & 0xFFFFFFFF, rc)}")


class slowed_down:
    """``with slowed_down(device):`` -- the device's shader clock held at its lowest level for the duration (tries the
    levels that do that in turn), automatic level restored on exit.  ``SmiRefused`` if none is accepted."""

    def __init__(self, device_index: int = 0):
        self.device = int(device_index)
        self.level = None

    def __enter__(self):
        last = None
        for lvl in (PERF_LEVEL_LOW, PERF_LEVEL_STABLE_MIN_SCLK):
            try:
                set_perf_level(self.device, lvl)
                self.level = lvl
                return self
            except SmiRefused as e:
                last = e
        raise last

    def __exit__(self, *exc):
        if self.level is not None:
            try:
                set_perf_level(self.device, PERF_LEVEL_AUTO)
            except SmiRefused:
                pass
        return False


def describe(device_index: int = 0) -> str:
    """One log-friendly line, or an explanation of why there is none."""
    try:
        s = sample(device_index)
    except Exception as e:  # noqa: BLE001  (telemetry must never take a training job down)
        return f"gpu telemetry unavailable: {e}"
    order = ("sclk_mhz", "sclk_peak_mhz", "mclk_mhz", "temp_junction_c", "temp_hbm_c", "power_w", "busy_pct")
    return "gpu telemetry: " + ", ".join(f"{k}={s[k]:.0f}" for k in order if k in s)
