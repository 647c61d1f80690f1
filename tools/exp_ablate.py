"""Development aid: price the stages of the scatter-pass kernel by switching them off in a
-DRDST_EXPERIMENTS build of the library (results are wrong by design; only times matter)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "_build", os.environ.get("RDST_EXP_LIB", "librdst_hip_exp.so"))


def main():
    from rdst_amd import _lib
    _lib.LIB_PATH = SO
    import rdst_amd
    lib = _lib.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    src = torch.randint(-(2**31), 2**31, (n,), dtype=torch.int32, device="cuda", generator=g)
    keys, tmp = torch.empty_like(src), torch.empty_like(src)
    names = {0: "full (plain first loads)", 32: "full, coherent loads only"} if len(sys.argv) > 2 else {0: "full", 1: "no look-back", 2: "no stores", 4: "sequential stores", 8: "no ranking", 16: "no ticket",
             1 | 16: "no look-back, no ticket", 1 | 2: "no look-back, no stores", 1 | 2 | 8: "no lb, no stores, no ranking",
             1 | 4: "no look-back, sequential stores", 1 | 2 | 8 | 16: "loads + counts only"}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record(); keys.copy_(src); e1.record(); torch.cuda.synchronize()
    print('device copy of the same array (4 GB read + 4 GB write): %.3f ms' % e0.elapsed_time(e1), flush=True)
    lds_list = [int(x) for x in os.environ.get("RDST_EXP_LDS", "0").split(",")]
    split = os.environ.get("RDST_EXP_SPLIT", "1") != "0"
    for cfg, lds_total in [(c, l) for c in [int(x) for x in os.environ.get("RDST_EXP_CFGS", "0,1,2,3").split(",")] for l in lds_list]:
        rdst_amd.set_tuning(cfg, 0, chain_split=split)
        lib.rdst_hip_exp_set_lds(ctypes.c_uint32(lds_total))
        print(f"-- cfg {cfg} dynamic LDS forced to >= {lds_total} B", flush=True)
        for mask, name in names.items():
            lib.rdst_hip_exp_set_ablation(ctypes.c_uint32(mask))
            tile = {0: 8192, 1: 12288, 2: 18432, 3: 21504}[cfg]
            stats_tiles = n // tile + 10 if not (mask & 1) else 0
            if stats_tiles:
                lib.rdst_hip_exp_stats(None, ctypes.c_uint64(stats_tiles))
            rdst_amd.set_profiling(True)
            for _ in range(3):
                keys.copy_(src)
                rdst_amd.sort_device_tensor(keys.view(torch.uint32), tmp.view(torch.uint32), check=False)
            torch.cuda.synchronize()
            try:
                rdst_amd.device_status()
            except Exception as e:  # noqa: BLE001
                print("   (device status:", str(e)[-60:], ")")
            pr = [rdst_amd.profile_run(r, 4) for r in range(1, 3)]
            p0 = sum(p["passes"][0] for p in pr) / len(pr)
            lb = ""
            if stats_tiles:
                import numpy as np
                rec = np.zeros((stats_tiles, 4), dtype=np.uint32)
                lib.rdst_hip_exp_stats(rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_uint64(stats_tiles))
                r = rec[rec[:, 0] > 0]  # a chain's first tile has no walk
                lb = (f" | LB(digit 0): windows mean {r[:,0].mean():.1f} p50 {np.median(r[:,0]):.0f} p99 {np.percentile(r[:,0],99):.0f}; "
                      f"blocked polls mean {r[:,1].mean():.1f}; tiles walked mean {r[:,2].mean():.1f} p99 {np.percentile(r[:,2],99):.0f}; "
                      f"clocks mean {r[:,3].mean():.0f} p50 {np.median(r[:,3]):.0f} p99 {np.percentile(r[:,3],99):.0f}")
            print(f"cfg {cfg} mask {mask:2d} {name:34s}: pass0 {p0:7.3f} ms  all {[round(x,2) for x in pr[-1]['passes']]}{lb}", flush=True)
            rdst_amd.set_profiling(False)


if __name__ == "__main__":
    main()
