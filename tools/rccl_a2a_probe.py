#!/usr/bin/env python3
"""all_to_all_single over a ONE-rank RCCL group at growing message sizes, against the identity it should be:

    python tools/rccl_a2a_probe.py          (on the GPU box; profiles/r03_rccl_a2a_probe.log)

RCCL 2.26.6 (torch 2.10 + ROCm 7.0) copies a message up to 1 GiB correctly and only the first HALF of a larger one.  The sharded
proof's row exchange (parallel.py TorchComm.all_to_all_tensor) therefore never hands RCCL a one-rank all-to-all (local copy) and
cuts messages above 512 MiB per pair into pieces; a small known-answer self-check alone would not have caught this."""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for words in (1 << 12, 1 << 24, 1 << 27, (1 << 28) - 8, 1 << 28, 135 << 22):
    send = torch.arange(words, dtype=torch.int64, device="cuda") * 3 + 1
    recv = torch.zeros(words, dtype=torch.int64, device="cuda")
    dist.all_to_all_single(recv, send)
    torch.cuda.synchronize()
    bad = int((recv != send).sum().item())
    first = int(torch.nonzero(recv != send)[0].item()) if bad else -1
    print("a2a world 1, %d words (%.2f GB): %d words differ, first %d" % (words, words * 8 / 1e9, bad, first), flush=True)
    del send, recv
dist.destroy_process_group()
