"""Per-stage times of one 10^9-key sort (HIP events between the launches), for each mode of the hybrid route:
    python tools/stage_times.py [dtype] [modes...]     modes: 1 default, 3 whole keys to K4, 2 ranked K4, 0 LSD"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdst_amd

name = sys.argv[1] if len(sys.argv) > 1 else "uint32"
modes = [int(x) for x in sys.argv[2:]] or [1, 3, 2, 0]
n = int(float(os.environ.get("RDST_N", "1e9")))
cfg = int(os.environ.get("RDST_CFG", "-1"))
if cfg >= 0:
    rdst_amd.set_tuning(pass_config=cfg)
it = torch.int32 if name in ("uint32", "float32", "int32") else torch.int64
g = torch.Generator(device="cuda").manual_seed(5)
info = torch.iinfo(it)
src = torch.randint(info.min, info.max, (n,), dtype=it, device="cuda", generator=g)
tmp = torch.empty_like(src).view(getattr(torch, name))
for mode in modes:
    rdst_amd.set_hybrid(mode if mode else False)
    acc = {}
    wall = []
    for rep in range(6):
        buf = src.clone().view(getattr(torch, name))
        rdst_amd.set_profiling(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rdst_amd.sort_device_tensor(buf, tmp)
        e1.record()
        torch.cuda.synchronize()
        p = rdst_amd.profile_run(-1, buf.element_size())
        rdst_amd.set_profiling(False)
        if rep == 0:
            continue
        wall.append(e0.elapsed_time(e1))
        for i, (nm, lv, ms) in enumerate(p["stages"]):
            acc.setdefault((i, nm, lv), []).append(ms)
    total = sum(sum(v) / len(v) for v in acc.values())
    print(f"mode {mode} route {rdst_amd.last_route()} wall {sum(wall) / len(wall):.3f} ms (stages sum {total:.3f})  " +
          "  ".join(f"{nm}{'' if lv is None else lv}={sum(v) / len(v):.3f}" for (i, nm, lv), v in sorted(acc.items())))
