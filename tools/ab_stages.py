"""Development aid: per-stage times of the 10^9-key sort for several builds of the library, interleaved on one box.
    python tools/ab_stages.py <dtype> <lib.so> [<lib.so> ...]       (paths relative to the repo root)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dtype, libs = sys.argv[1], sys.argv[2:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, RDST_HIP_LIB=os.path.join(root, lib))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "stage_times.py"), dtype, "1"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("mode")]
        print(f"{lib:36s} {line[-1] if line else 'FAILED ' + out.stderr[-400:]}", flush=True)
