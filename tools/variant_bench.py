#!/usr/bin/env python3
"""Diagnostic: build libcrowdstep variants with extra -D flags (timing experiments, results meaningless) and print
the kernel time bench.py measures for each.  usage: tools/variant_bench.py NAME=FLAGS ...   (FLAGS comma separated)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from social_navigation_pyenvs_amd.csrc import build as hb  # noqa: E402

extra = [a for a in sys.argv[1:] if a.startswith("--")]
for spec in [a for a in sys.argv[1:] if not a.startswith("--")]:
    name, _, flags = spec.partition("=")
    so = os.path.join(ROOT, "gpurun_out", f"libcrowdstep_{name}.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    cmd = [hb.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-o", so] + [f for f in flags.split(",") if f] + \
          [os.path.join(hb.CSRC, x) for x in hb.SOURCES]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    code = ("import sys; sys.argv=['bench.py','--no-cpu-baseline']+%r; sys.path.insert(0,%r);"
            "from social_navigation_pyenvs_amd import _lib; _lib.LIB_PATH=%r; import bench; bench.main()" % (extra, ROOT, so))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{name:24s} kernel {d['roofline']['kernel_avg_ms'] * 1e3:8.2f} us   finite {d['finite_fraction']:.3f}", flush=True)
    except Exception:
        print(name, "FAILED", out.stdout[-300:], out.stderr[-600:], flush=True)
