"""Spawn N gloo ranks on CPU (file-store rendezvous), run a worker in each, collect its return value."""
import os
import sys
import tempfile
import traceback

import torch.multiprocessing as mp


def _entry(rank, world, store, fn, kwargs, q, use_oracle_backend, device=None, backend_kwargs=None, env=None):
    try:
        os.environ.update(env or {})
        here = os.path.dirname(os.path.abspath(__file__))
        repo = os.path.dirname(here)
        for p in (repo, os.path.join(repo, "nvidia-resiliency-ext_amd"), here, os.path.join(here, "golden")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch

        torch.set_num_threads(1)
        if device is None and not os.path.exists("/dev/kfd"):
            # (a GPU-less host: probe for devices BEFORE anything loads librocprofiler-sdk -- see conftest.probe_devices_first...)
            torch.cuda.is_available()
            torch._C._get_accelerator()
        if device is not None:  # GPU tests: several ranks share one device, product backend
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(device)
        os.environ["RANK"] = str(rank)
        os.environ["WORLD_SIZE"] = str(world)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        if world > 1:
            torch.distributed.init_process_group("gloo", init_method=f"file://{store}", world_size=world, rank=rank)
        if use_oracle_backend:
            from nvrx_straggler import backend
            from oracle_backend import OracleBackend

            backend.set_backend(OracleBackend(**(backend_kwargs or {})))
        res = fn(rank, world, **kwargs)
        q.put((rank, "ok", res))
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
    except BaseException:
        q.put((rank, "error", traceback.format_exc()))


def run_ranks(fn, world, timeout=180, use_oracle_backend=True, device=None, backend_kwargs=None, env=None, **kwargs):
    """Returns [result of rank 0, rank 1, ...]; raises if any rank failed.  ``use_oracle_backend=False`` +
    ``device=0``: every rank runs the PRODUCT backend on that GPU (multi-process GPU tests, gloo group)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.NamedTemporaryFile(delete=True) as f:
        store = f.name
    procs = [ctx.Process(target=_entry, args=(r, world, store, fn, kwargs, q, use_oracle_backend, device, backend_kwargs, env)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    try:
        for _ in range(world):
            rank, status, payload = q.get(timeout=timeout)
            if status != "ok":
                raise AssertionError(f"rank {rank} failed:\n{payload}")
            out[rank] = payload
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    return [out[r] for r in range(world)]
