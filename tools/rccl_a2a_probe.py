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
