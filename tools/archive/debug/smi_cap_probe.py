"""Can this box slow its GPU through ROCm SMI (perf level / clock range / power cap)?  Times 30 x matmul(4096^2, bf16)
before and after each attempt and always restores the automatic level."""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd"), REPO]
import torch
from nvrx_straggler import gpu_telemetry as gt
lib = gt._load()
x = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
def bench():
    for _ in range(5): x @ x
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): x @ x
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 30 * 1e3
def clocks():
    s = gt.sample(0); return {k: round(v) for k, v in s.items() if "clk" in k or k == "power_w"}
lvl = ctypes.c_int(0)
print("perf_level_get rc", lib.rsmi_dev_perf_level_get(0, ctypes.byref(lvl)), "level", lvl.value)
print("baseline ms/matmul %.3f" % bench(), clocks(), flush=True)
lib.rsmi_dev_perf_level_set_v1.argtypes = [ctypes.c_uint32, ctypes.c_int]
lib.rsmi_dev_perf_level_set.argtypes = [ctypes.c_uint32, ctypes.c_int]
lib.rsmi_dev_clk_range_set.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
lib.rsmi_dev_power_cap_set.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]
lib.rsmi_dev_power_cap_range_get.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
lib.rsmi_dev_power_cap_get.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
try:
    for name, code in (("LOW", 1), ("STABLE_MIN_SCLK", 7), ("DETERMINISM", 8), ("MANUAL", 3)):
        rc = lib.rsmi_dev_perf_level_set_v1(0, code)
        print(f"perf_level_set_v1({name}) rc {rc}", end=" ")
        if rc == 0:
            time.sleep(0.5); print("-> ms/matmul %.3f" % bench(), clocks(), end=" ")
            if name == "MANUAL":
                rc2 = lib.rsmi_dev_clk_range_set(0, 500, 900, 0)
                print("| clk_range_set(500,900) rc", rc2, end=" ")
                if rc2 == 0:
                    time.sleep(0.5); print("-> ms/matmul %.3f" % bench(), clocks(), end=" ")
        print("| restore rc", lib.rsmi_dev_perf_level_set_v1(0, 0), flush=True)
    rc = lib.rsmi_dev_clk_range_set(0, 500, 900, 0)
    print("clk_range_set(500,900) without MANUAL rc", rc, end=" ")
    if rc == 0:
        time.sleep(0.5); print("-> ms/matmul %.3f" % bench(), clocks(), end=" ")
    print("| restore rc", lib.rsmi_dev_perf_level_set_v1(0, 0), flush=True)
    lo, hi, cur = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    print("power_cap_range_get rc", lib.rsmi_dev_power_cap_range_get(0, 0, ctypes.byref(hi), ctypes.byref(lo)), 'min', lo.value, 'max', hi.value,
          "cap_get rc", lib.rsmi_dev_power_cap_get(0, 0, ctypes.byref(cur)), cur.value)
    if cur.value:
        rc = lib.rsmi_dev_power_cap_set(0, 0, max(lo.value, cur.value // 4))
        print("power_cap_set(", max(lo.value, cur.value // 4), ") rc", rc, end=" ")
        if rc == 0:
            time.sleep(0.5); print("-> ms/matmul %.3f" % bench(), clocks(), end=" ")
            print("| restore rc", lib.rsmi_dev_power_cap_set(0, 0, cur.value), end="")
        print(flush=True)
finally:
    lib.rsmi_dev_perf_level_set_v1(0, 0)
print("after restore ms/matmul %.3f" % bench(), clocks())
