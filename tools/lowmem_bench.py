"""Development aid: the low-memory route (scratch = len / 64, len / 16, len / 4) against the ordinary one on 10^9 u32 keys."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdst_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(5)
src = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
keys = torch.empty_like(src)
for div in (64, 16, 4):
    scratch = torch.empty(n // div, dtype=torch.uint32, device="cuda")
    best = 1e9
    for _ in range(3):
        keys.copy_(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rdst_amd.sort_device_tensor_lowmem(keys.view(torch.uint32), scratch)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    k = keys ^ (-(2**31))
    ok = bool((k[1:] >= k[:-1]).all())
    print(f"scratch len/{div:<3d} ({scratch.numel() * 4 / 1e6:8.1f} MB): {best * 1e3:8.2f} ms  {n / best / 1e9:6.2f} Gkeys/s  ok={ok}", flush=True)
    del scratch
part = torch.empty(n // 64, dtype=torch.uint32, device="cuda")
keys.copy_(src)
torch.cuda.synchronize()
t0 = time.perf_counter()
split = rdst_amd.partition_device(keys.view(torch.uint32), 3, 0x80, part)
torch.cuda.synchronize()
print(f"partition (top byte == 0x80), scratch len/64: {(time.perf_counter() - t0) * 1e3:8.2f} ms, split {split}")
