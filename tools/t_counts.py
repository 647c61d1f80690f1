import sys, time
sys.path.insert(0, "/root/repo")
import torch, rdst_amd
n = 10**9
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=g).view(torch.uint32)
for name, fn in (("level_counts(3)", lambda: rdst_amd.level_counts(x, 3)), ("all_level_counts", lambda: rdst_amd.all_level_counts(x)),
                 ("scatter_level(3)", lambda: rdst_amd.scatter_level(x, 3))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 5 * 1e3, "ms", flush=True)
