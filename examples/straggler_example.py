#!/usr/bin/env python3
"""Straggler detection on MI355X in a small data-parallel training loop.

What the reference's ``examples/straggler/example.py`` shows on CUDA, on ROCm: every rank wraps its forward pass in
``Detector.detection_section("fwd", profile_cuda=True)``, all ranks call ``Detector.generate_report()`` every
``--report-interval`` steps, rank 0 prints the relative and individual GPU scores and whoever ``identify_stragglers``
flags, with the ROCm SMI telemetry line of this rank's GPU next to it.  The data is synthetic (MNIST-shaped): there is
nothing to download.

    # one process per GPU over RCCL (the package is imported before torch touches the GPU: in a multi-rank job that
    # selects per-kernel GPU timing, the reference's data model -- collectives inside the section do not hide a slow GPU)
    python examples/straggler_example.py --num-processes 8

    # a box with ONE GPU: the ranks share it over gloo
    python examples/straggler_example.py --num-processes 2 --share-gpu

To see a straggler, slow one GPU down while it runs -- the ROCm counterpart of the reference's ``nvidia-smi -lgc 800``:

    rocm-smi -d 3 --setperflevel low          # ... and `--setperflevel auto` to give it back

or let the example do it: ``--slow-rank 3`` holds that rank's shader clock at its lowest level through ROCm SMI from
step ``--slow-from`` on (root and a writable sysfs needed).  Where the driver refuses, or on a box whose ranks share one
GPU, ``--slow-by simulated`` puts a stand-in for "a kernel that takes longer on a slower GPU" into every rank's section:
a spin kernel whose duration is its argument, 1.5x longer on the slow rank -- the report then reads as it would with a
GPU at two thirds of its speed.
"""
import argparse
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "nvidia-resiliency-ext_amd")]

# the drop-in import path of the reference package; importing it BEFORE the HIP runtime starts lets a multi-rank job use
# per-kernel GPU timing (rocprofiler-sdk accepts tools only before that)
from nvidia_resiliency_ext.attribution import straggler  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402
from torch.nn.parallel import DistributedDataParallel as DDP  # noqa: E402


class Model(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(784, width), nn.ReLU(), nn.Linear(width, width), nn.ReLU(), nn.Linear(width, width),
                                    nn.ReLU(), nn.Linear(width, 10))

    def forward(self, x):
        return self.layers(torch.flatten(x, 1))


def train(args) -> None:
    # the order of the reference's example (examples/straggler/example.py:60-66): the detector first, the GPU afterwards --
    # the detector's device side is created at the first section, on the device that is current then
    straggler.Detector.initialize(gather_on_rank0=True)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device_index = 0 if args.share_gpu else local_rank
    if world > 1:
        dist.init_process_group("gloo" if args.share_gpu else "nccl")
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    torch.manual_seed(42 + rank)
    model = Model(args.width).to(device)
    net = DDP(model, device_ids=None if args.share_gpu else [device_index]) if world > 1 else model
    optim = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.5)
    loss_fn = nn.CrossEntropyLoss()
    data = torch.randn(args.batch_size, 1, 28, 28, device=device)
    target = torch.randint(0, 10, (args.batch_size,), device=device)
    slow_ctx = None
    t_start = time.monotonic()
    for step in range(args.steps):
        if rank == args.slow_rank and step == args.slow_from and args.slow_by == "clock":
            from nvrx_straggler import gpu_telemetry

            try:
                slow_ctx = gpu_telemetry.slowed_down(device_index).__enter__()
                print(f"[rank {rank}] shader clock of GPU {device_index} held at its lowest level from step {step} on", flush=True)
            except gpu_telemetry.SmiRefused as e:
                print(f"[rank {rank}] ROCm SMI refused to slow the GPU down ({e}); use --slow-by simulated", flush=True)
        with straggler.Detector.detection_section("fwd", profile_cuda=True):
            output = net(data)
            if args.slow_by == "simulated":   # one kernel whose duration says how fast "this GPU" is
                slow = rank == args.slow_rank and step >= args.slow_from
                torch.cuda._sleep(int(args.simulated_cycles * (1.5 if slow else 1.0)))
        loss = loss_fn(output, target)
        optim.zero_grad()
        loss.backward()
        optim.step()
        if step % args.report_interval == 0 and step:
            report = straggler.Detector.generate_report()
            if rank == 0:
                print(f"step {step}: GPUs relative perf: { {r: round(s, 3) for r, s in report.gpu_relative_perf_scores.items()} }")
                print(f"step {step}: GPUs individual perf: { {r: round(s, 3) for r, s in report.gpu_individual_perf_scores.items()} }")
                found = report.identify_stragglers(gpu_rel_threshold=args.threshold, gpu_indiv_threshold=args.threshold)
                for kind in ("straggler_gpus_relative", "straggler_gpus_individual"):
                    if found[kind]:
                        print(f"step {step}: {kind}: {sorted((s.rank, s.node) for s in found[kind])}")
                print(f"step {step}: {straggler.Detector.gpu_telemetry_line()}", flush=True)
    if slow_ctx is not None:
        slow_ctx.__exit__(None, None, None)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"time per step [ms]: {(time.monotonic() - t_start) / args.steps * 1e3:.3f}")
    straggler.Detector.shutdown()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-processes", type=int, default=1)
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on GPU 0 over gloo (a box with one GPU)")
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--batch-size", type=int, default=4096)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--report-interval", type=int, default=300)
    ap.add_argument("--threshold", type=float, default=0.75)
    ap.add_argument("--slow-rank", type=int, default=-1)
    ap.add_argument("--slow-from", type=int, default=300)
    ap.add_argument("--slow-by", choices=["clock", "simulated"], default="clock")
    ap.add_argument("--simulated-cycles", type=float, default=3e6, help="--slow-by simulated: spin cycles of the stand-in kernel")
    args = ap.parse_args()
    if "RANK" in os.environ or args.num_processes == 1:
        train(args)
        return
    with socket.socket() as s:   # one process per rank, the environment torchrun would give them, rendezvous on 127.0.0.1
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable] + sys.argv,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.num_processes),
                                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0"))
             for r in range(args.num_processes)]
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    sys.exit(rc)


if __name__ == "__main__":
    main()
