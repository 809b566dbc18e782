"""The tracer soak of tests/test_gpu_01_ktrace_datapath.py (a second stream, a second launching thread, 64-deep rings, a report every 37
entries) for 75 s per mode instead of 4: python tools/long_tracer_soak.py (needs an MI355X)."""
import json, sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nvidia-resiliency-ext_amd")]
import test_gpu_01_ktrace_datapath as t
for a in ("0", "1"):
    out = t._run(t.SOAK_SCRIPT, {"SOAK_SECONDS": "75", "SOAK_ASYNC": a}, timeout=400)
    c = out["counters"]
    ok = (c["enqueued"] == c["arrived"] + c["forgiven"] and c["forgiven"] == 0 and c["sink_errors"] == 0 and c["lost_no_row"] == 0
          and c["delivered"] + c["own_skipped"] + c["blit_skipped"] == c["arrived"])
    print("soak", "async" if a == "1" else "sync", {k: out[k] for k in ("entries", "reports", "kernel_samples_reported", "keys")},
          {k: c[k] for k in ("enqueued", "arrived", "delivered", "own_skipped", "blit_skipped", "forgiven", "pump_flushes")}, "OK" if ok else "MISMATCH")
