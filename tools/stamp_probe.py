#!/usr/bin/env python3
"""Diagnostic: build libcrowdstep_stamps.so with -DCS_STAMPS and print the share of wave cycles
each section of k_sfm_step takes (never shipped, never timed for throughput)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from social_navigation_pyenvs_amd import _lib, scenarios as sc  # noqa: E402
from social_navigation_pyenvs_amd.csrc import build as hb  # noqa: E402

# libcrowdstep_stamps.so beside the product library (content-keyed: built in the build container, it travels to the GPU box and is reused there)
so = hb.build(variant="stamps", extra_flags=["-DCS_STAMPS"])
_lib.LIB_PATH = so
_lib._lib = None
from social_navigation_pyenvs_amd.batched import CrowdWorlds  # noqa: E402

W, n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 25
model = sys.argv[2] if len(sys.argv) > 2 else "hsfm_farina"
if model == "orca":
    pos, yaw, g = sc.circular_crossing(W, n, 7.0, 1000)
    S = sc.make_states(pos, yaw, g)
    d = g[:, :, 0] - S[:, :, 0:2]
    S[:, :, 5:7] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    cw = CrowdWorlds(S, g, None, np.full((W, n), 0.01), None, type="orca", layout="soa")
    gg, b, wpb = cw.launch_geometry() if False else ((W + (64 // n) - 1) // (64 // n), 64, 64 // n)
    buf = _lib.DeviceBuffer((gg, 8), np.uint64)
    _lib.load().cs_debug_set_stamp_buffer(C.c_void_p(buf.ptr))
    # STAMP_WARMUP Gym steps first (not counted), then STAMP_STEPS counted ones: bench.py's two named phases of the crossing are
    # (0, 20) "first20" and (25, 20) "dense"
    warm, steps = int(os.environ.get("STAMP_WARMUP", "0")), int(os.environ.get("STAMP_STEPS", "3"))
    for _ in range(warm):
        cw.step(0.0125, 20)
    cw.sync()
    # the kernel WRITES its wavefronts' stamps (it does not add to the buffer): one download per launch, summed here
    st = np.zeros((gg, 8), np.float64)
    for _ in range(steps):
        cw.step(0.0125, 20)
        cw.sync()
        st += buf.download().astype(np.float64)
    names = ["pre (robot/loads)", "neighbour selection", "ORCA lines", "LP2 (+LP1)", "LP3", "update+goal+respawn", "-", "-"]
    tot = st.sum(1).mean()
    sub = 20 * steps
    print(f"ORCA N={n} W={W}, Gym steps {warm}..{warm + steps}: mean wave cycles per substep {tot / sub:.0f} (s_memtime ticks = shader clocks; the stamps drain the pipes: shares, not absolute costs)")
    for k, nm in enumerate(names):
        print(f"  {nm:28s} {st[:, k].mean() / sub:9.1f} cyc/substep  {100 * st[:, k].mean() / tot:5.1f} %")
    per_wave = st.sum(1)
    print(f"  the launch waits for its slowest wavefront: slowest / mean wavefront = {per_wave.max() / per_wave.mean():.3f}, 99th percentile / mean = {np.percentile(per_wave, 99) / per_wave.mean():.3f}")
    h = len(per_wave) // 2
    print(f"  second half of the grid (the younger wavefront of each SIMD) / first half: {per_wave[h:].mean() / per_wave[:h].mean():.3f}")
    if os.environ.get("STAMP_DUMP"):
        np.save(os.environ["STAMP_DUMP"], st)
    sys.exit(0)
if len(sys.argv) > 4 and sys.argv[4] in ("circle", "walls"):   # cfg2-style: circular crossing only, no respawn rule; "walls": cfg5-style
    pos, yaw, g = sc.circular_crossing(W, n, 7.0, 1000)
    S, goals, P = sc.make_states(pos, yaw, g), g, np.tile(sc.default_params(model), (n, 1))
    cw = CrowdWorlds(S, goals, P, None, sc.polygon_walls() if sys.argv[4] == "walls" else None, type=model, all_params_equal=True, layout="soa")
else:
    S, goals, P, rb = sc.hybrid_worlds(W, n, model)
    cw = CrowdWorlds(S, goals, P, None, sc.polygon_walls() if (len(sys.argv) > 4 and sys.argv[4] == "hybridwalls") else None, type=model, all_params_equal=True, respawn_bounds=rb,
                     respawn_worlds=(np.arange(W) % 2 == 1).astype(np.int32), layout="soa")
g, b, wpb = cw.launch_geometry()
buf = _lib.DeviceBuffer((g * (b // 64), 20), np.uint64)
lib = _lib.load()
lib.cs_debug_set_stamp_buffer(C.c_void_p(buf.ptr))
# STAMP_WARMUP Gym steps first, then the stamps of STAMP_STEPS launches summed (the kernel overwrites its buffer: one download per launch)
warm, steps = int(os.environ.get("STAMP_WARMUP", "0")), int(os.environ.get("STAMP_STEPS", "5"))
for _ in range(warm):
    cw.step(0.0125, 20)
cw.sync()
st = np.zeros((g * (b // 64), 20), np.float64)
for _ in range(steps):
    cw.step(0.0125, 20)
    cw.sync()
    st += buf.download().astype(np.float64)
st /= steps
names = ["goal switch", "rot+desired+walls", "pair-once: contact pass/ballot", "torque+euler+lds write", "barrier", "respawn check", "all-partners pair loop (not pair-once)", "loop top",
         "pair-once: zero accumulators", "pair-once: partner groups", "pair-once: reaction sum", "contact passes per 1000 wavefront-substeps (a count, not cycles)"]
tot = st[:, :11].sum(1).mean()
print(f"N={n} {model}, Gym steps {warm}..{warm + steps}: mean wave cycles in loop = {tot:.0f} (per substep {tot / 20:.0f})")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {st[:, k].mean() / 20:9.1f} cyc/substep  {100 * st[:, k].mean() / tot:5.1f} %")
fixed = ["entry -> every load of the load phase returned", "-> first substep (prologue arithmetic, LDS publication)", "last substep -> stores issued (epilogue)",
         "-> stores acknowledged (not waited for by the wavefront)"]
if st[:, 18].mean() > 0:   # slots 16 / 17 / 18 of the last launch (not summed): wall clock of each wavefront's first / last instruction, its wait for the kernel arguments
    last = buf.download().astype(np.float64)
    t0, t1 = last[:, 17], last[:, 18]
    print(f"  last launch, wall clock (s_memrealtime, 10 ns ticks): first wavefront starts at 0, the last one {10e-3 * (t0.max() - t0.min()):.2f} us later; "
          f"a wavefront lives {10e-3 * (t1 - t0).mean():.2f} us (min {10e-3 * (t1 - t0).min():.2f}, max {10e-3 * (t1 - t0).max():.2f}); first start -> last end {10e-3 * (t1.max() - t0.min()):.2f} us")
    q = [0, 10, 50, 90, 99, 100]
    print("  start of the wavefronts after the first one, percentiles " + str(q) + ": " + " ".join(f"{10e-3 * v:.2f}" for v in np.percentile(t0 - t0.min(), q)) + " us")
    print("  end of the wavefronts before the last one,   percentiles " + str(q) + ": " + " ".join(f"{10e-3 * v:.2f}" for v in np.percentile(t1.max() - t1, q)) + " us")
    wgw = int(os.environ.get("CROWDSTEP_WG_WAVES", "4"))
    xcd = (np.arange(len(t0)) // wgw) % 8          # workgroups go to the eight XCDs in turn
    print("  median start per XCD (workgroup index mod 8): " + " ".join(f"{10e-3 * (np.median(t0[xcd == k]) - t0.min()):.2f}" for k in range(8)) + " us;  spread inside an XCD (90th - 10th percentile): "
          + " ".join(f"{10e-3 * (np.percentile(t0[xcd == k], 90) - np.percentile(t0[xcd == k], 10)):.2f}" for k in range(8)) + " us")
    print("  wavefront life per XCD (median): " + " ".join(f"{10e-3 * np.median((t1 - t0)[xcd == k]):.2f}" for k in range(8)) + " us")
    h = len(t0) // 2
    print(f"  first half of the grid starts {10e-3 * (np.median(t0[:h]) - t0.min()):.2f} us after the first wavefront (median), the second half {10e-3 * (np.median(t0[h:]) - t0.min()):.2f} us")
    print(f"  first instruction -> kernel arguments loaded: {last[:, 16].mean():.0f} shader clocks (max {last[:, 16].max():.0f})")
print("  around the substeps (shader clocks per launch, mean over the wavefronts):")
for k, nm in enumerate(fixed):
    print(f"    {nm:62s} {st[:, 12 + k].mean():9.0f} cycles  ({st[:, 12 + k].mean() / 2.4e3:5.2f} us at 2.4 GHz)")
per_wave = st[:, :11].sum(1)
print(f"  the launch waits for its slowest wavefront: slowest / mean wavefront = {per_wave.max() / per_wave.mean():.3f}, 99th percentile / mean = {np.percentile(per_wave, 99) / per_wave.mean():.3f}")
print(f"  second half of the grid (the younger wavefront of each SIMD) / first half: {per_wave[len(per_wave) // 2:].mean() / per_wave[:len(per_wave) // 2].mean():.3f}")
if os.environ.get("STAMP_DUMP"):
    np.save(os.environ["STAMP_DUMP"], st)
