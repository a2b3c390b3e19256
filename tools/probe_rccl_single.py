#!/usr/bin/env python3
"""GPU box (1 GPU): a one-rank RCCL group -- checks that backend "nccl" initialises here and all-reduces the
fp64 scalar the path uses (the 8-byte sum of FPKM), and times it.  The N > 1 path itself is covered by the
2-rank gloo tests (tests/test_dist_cpu.py) and by 2 ranks sharing the GPU (SB_DIST_BACKEND=gloo torchrun ... bench.py --gpus 2)."""
import os, torch, torch.distributed as dist, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t); torch.cuda.synchronize(); print("allreduce f64 ok", t.item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): dist.all_reduce(t)
e1.record(); torch.cuda.synchronize(); print("us per 8-byte all_reduce (1 rank): %.1f" % (e0.elapsed_time(e1) * 10))
dist.barrier(); dist.destroy_process_group(); print("done")
