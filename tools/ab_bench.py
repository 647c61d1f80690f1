"""Development aid: A/B two builds of the library in one process-per-arm loop on the same box."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
arms = sys.argv[1:]  # library paths
code = '''
import sys, json
sys.path.insert(0, %r)
from rdst_amd import _lib
_lib.LIB_PATH = %r
import runpy
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "6", "--warmup", "2"]
runpy.run_path(%r, run_name="__main__")
'''
for rep in range(3):
    for lib in arms:
        out = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, lib), os.path.join(root, "bench.py"))],
                             capture_output=True, text=True)
        if not any(l.startswith("{") for l in out.stdout.splitlines()):
            print(lib, "FAILED:", out.stderr[-600:]); continue
        out = out.stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        print(f"{lib:40s} {d['value']:8.2f} Gkeys/s  pass {d['roofline']['avg_launch_ms']:.4f} ms  hist {d['roofline']['histogram_ms']:.4f} ms", flush=True)
