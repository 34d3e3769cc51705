#!/usr/bin/env python3
"""bench.py -- throughput of the crowd-step hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch: ONE Gym step (= `substeps` fused Euler substeps
of dt = 0.0125 s, social_nav_gym.py:240-245 / env.config:2-4) of every world resident on the GPU,
i.e. one cs_step launch.  Workload at N=1 = BASELINE.json configs[2] (the config the metric names):
4096 worlds x 25-agent Headed-SFM (hsfm_farina) hybrid scenario (worlds with an even global id: circular
crossing R=7; odd: 14x3 m parallel traffic with respawn), synthetic random-goal crowds, state resident in HBM.
Worlds are independent: each rank owns its own 4096 worlds (weak scaling, no collective on the data
path); the only collectives are the timing barrier and the MAX over ranks.

Timing -- STATIONARY: after the W warm-up steps the worlds (state rows, goal lists, robot rows) are snapshotted; the K
steps are captured once into a HIP graph (untimed); the timed region -- [untimed: restore the snapshot] barrier +
synchronize | ONE graph launch = exactly K steps | synchronize + barrier -- is repeated R times (`timing_repeats`,
default 50), so every repeat times the SAME trajectory segment (Gym steps W .. W+K of the episode that starts at the
reset: with W = 20, K = 200 that is the span of a whole reference episode, time_limit 50 s = 200 Gym steps).
`ms_per_step` is the MEDIAN over the repeats of (MAX over ranks of the elapsed wall time) / K; HIP events on the launch
stream around every replay give the kernel duration of the same samples (mean and median reported; kernel <= wall).

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): bench.py starts its own N ranks -- N fresh child processes, one
per GPU, before this process imports torch or touches HIP -- relays rank 0's line and exits with the children's worst exit code
(launch_ranks).  Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks the launcher made are used as they are.

Prints ONE COMPACT JSON line (< 4 KB: compact_line) on rank 0; the full blocks (every other_configs roofline, the PMC figures and
their sources) go to gpurun_out/bench_full.json and to stderr.  The line carries
  roofline      algorithmic bytes per launch / average kernel duration against the 8 TB/s HBM3E peak (the contract's
                figure: a NORMALISED ALGORITHMIC THROUGHPUT -- the 20 fused substeps keep the state in registers / LDS,
                so real HBM traffic is ~3 % of peak), plus `frac_vs_copy` (against a device-copy bandwidth measured in
                this run), `traffic` / `valu` (rocprofv3 PMC figures of this command from profiles/pmc_summary.json, tagged
                with their source files and with `pmc_build_matches`: whether they were measured on the library build that is
                running now), `valu_frac` (PMC VALU instructions per launch x 4 issue cycles / (1024 SIMDs x kernel cycles at
                2.4 GHz): the bound that really holds for these kernels)
  ranks         per rank: device index, PCI address, kernel time -- which card every rank sat on; `dist` = backend, world size, RCCL version
  gym_step      the device-resident Gym step (reward + bookkeeping, 20 substeps + observation, masked copy): us per batched step without
                resets, with same-step auto-reset and in NEXT_STEP mode (N = 1, rank 0)
  other_configs kernel time and roofline fraction of BASELINE.json configs[1], [3] (two named phases of the crossing) and [4]
                (cfg5 = `--total-worlds 65536` x 50 HSFM humans of which 3 immobile + 3 polygon walls, strong-split over the
                ranks), and of the shapes beside the benchmark's own (visible robot, 30 humans, per-agent parameters,
                hsfm_new_guo), measured in the same run with the same protocol
  cpu_baseline  the C oracle timed on the host cores of this box (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per agent-substep, f32 (SURVEY.md §8d / BASELINE.md §4)
ALG_BYTES = {"sfm": 52, "hsfm": 76, "orca": 48}
# flop-equivalents per pair evaluation: Helbing 45, Guo 60, Moussaid 120
PAIR_FLOPS = {"helbing": 45, "guo": 60, "moussaid": 120}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3   # MI355X vector fp32 peak (MI355X_MICROARCH.md)
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9
# What one SIMD needs per wave64 vector instruction when >= 2 wavefronts share it, MEASURED per instruction class on an MI355X
# (tools/valu_issue_ceiling.hip, profiles/archive/r5_valu_issue_ceiling.txt: independent streams in inline assembly at 1 / 2 / 4 / 8
# wavefronts per SIMD, shader cycles by s_memtime; round 4 priced everything at 4): v_fma / v_mul / v_add / v_mov / v_and..xor 2.2;
# v_min / v_max / v_max3 / v_cmp / v_cndmask (VOP3 or fed by a compare) / DPP / f64 min-max 4.1; v_exp / v_rcp / v_rsq / v_sqrt 8.2.
# One wavefront ALONE on a SIMD issues no faster than one instruction per 4.4 - 5.6 cycles whatever the class.
VALU_CLASS_CYCLES = {"fma": 2.2, "other": 4.1, "trans": 8.2}
CPU_ROWS = ("cfg2", "cfg4_first20", "cfg4_dense", "cfg5_shard", "moussaid", "cfg3_new_guo", "robot26", "n30", "peragent", "n15", "n40")   # other_configs rows that get their own CPU figure (cfg5 = 8 x its shard)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_summary.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=50, help="timed replays of the K-step graph (median reported)")
    ap.add_argument("--worlds", type=int, default=4096, help="worlds per GPU (weak scaling)")
    ap.add_argument("--total-worlds", type=int, default=None, help="worlds of the whole job, split evenly over the ranks (strong scaling)")
    ap.add_argument("--agents", type=int, default=25)
    ap.add_argument("--model", default="hsfm_farina")
    ap.add_argument("--scenario", default="hybrid", choices=["hybrid", "circle", "traffic"])
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--dt", type=float, default=0.0125)
    ap.add_argument("--layout", default="soa", choices=["aos", "soa"])
    ap.add_argument("--walls", action="store_true", help="add 3 shared polygon walls")
    ap.add_argument("--static", type=int, default=0, help="first N humans immobile (circular_crossing_with_static_obstacles flavour; --scenario circle)")
    ap.add_argument("--device-generator", action="store_true", help="worlds from cs_generate_worlds (generators.static_obstacle_crossing: circle crossing R=14, seed 1000 + global id): what the cfg5 entry of other_configs runs")
    ap.add_argument("--robot", action="store_true", help="a visible robot as the last state row of every world (rows = agents + 1), driven by a constant action")
    ap.add_argument("--per-agent-params", action="store_true", help="every human its own (jittered) parameter row: all_params_equal = False")
    ap.add_argument("--orca-math", default="default", choices=["default", "exact", "fast", "fma"], help="cs_worlds.orca_math of ORCA worlds (default: CROWDSTEP_ORCA_MATH, else exact -- the build that equals the restatement bit for bit)")
    ap.add_argument("--no-restore", action="store_true", help="let the worlds evolve from replay to replay (the round-2 protocol) instead of restoring the snapshot")
    ap.add_argument("--eager", action="store_true", help="launch every step from Python (default: one HIP graph of K steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-gym-step", action="store_true", help="skip the device-resident Gym step figures (gym_step)")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"), help="where the full (uncompacted) result goes")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the N > 1 code path on a one-GPU box")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --same-device: rehearse the N > 1 path with several ranks on ONE GPU")
    ap.add_argument("--same-device", action="store_true", help="every rank uses GPU 0 (rehearsal on a one-GPU box; never for measurements)")
    return ap.parse_args(argv)


def spec_key(spec) -> str:
    """key of a configuration in profiles/pmc_summary.json"""
    W = spec["worlds"]
    return (f"{spec['model']}_{spec['agents']}_{spec['scenario']}" + ("_walls" if spec["walls"] else "") + ("_static" if spec["static"] else "")
            + ("_robot" if spec.get("robot") else "") + ("_peragent" if spec.get("per_agent") else "")
            + (f"_{spec['phase_key']}" if spec.get("phase_key") else "") + (f"_{spec['orca_math']}" if spec.get("orca_math") not in (None, "default", "exact") else "")
            + ("" if W in (4096, 8192) else f"_{W}"))


def workload_spec(args) -> dict:
    # ORCA's cost follows the crossing: the two named phases of other_configs keep their PMC entries when they are run on their own
    phase = {(0, 20): "first20", (25, 20): "dense"}.get((args.warmup, args.steps)) if (args.model == "orca" and args.scenario == "circle") else None
    return dict(phase_key=phase, orca_math=args.orca_math, name="main", device_generator=bool(args.device_generator), model=args.model, agents=args.agents, scenario=args.scenario, walls=bool(args.walls),
                static=int(args.static), substeps=args.substeps, dt=args.dt, layout=args.layout, robot=bool(args.robot),
                per_agent=bool(args.per_agent_params), worlds=args.worlds, total_worlds=args.total_worlds)


def other_config_specs(args) -> list[dict]:
    """BASELINE.json configs[1], [3], [4] beside the headline configs[2], and the shapes around the benchmark's own."""
    base = dict(substeps=args.substeps, dt=args.dt, layout=args.layout, walls=False, static=0, total_worlds=None, robot=False, per_agent=False,
                scenario="hybrid", model="hsfm_farina", agents=25, worlds=4096, warmup=20, steps=50)
    return [
        dict(base, name="cfg2", model="sfm_helbing", agents=10, scenario="circle",
             title="4096 worlds/GPU x 10-agent SFM (sfm_helbing) circle crossing"),
        # ORCA's cost follows the crossing (how many agents have an infeasible linear programme): two named, restored phases
        dict(base, name="cfg4_first20", model="orca", scenario="circle", warmup=0, steps=20, phase_key="first20",
             title="4096 worlds/GPU x 25-agent ORCA circle crossing, Gym steps 0-20 from the reset (agents still near the rim)"),
        dict(base, name="cfg4_dense", model="orca", scenario="circle", warmup=25, steps=20, phase_key="dense",
             title="4096 worlds/GPU x 25-agent ORCA circle crossing, Gym steps 25-45 (the crowd meets at the centre: the dense phase)"),
        # (the two rows above: the library's default arithmetic, EXACT -- every substep equals oracle/orca_oracle.c bit for bit; below: the opt-in
        # "fma" arithmetic, within 1e-5 per substep except where float32 does not determine RVO2's answer, tests/test_gpu_orca_fast.py)
        dict(base, name="cfg4_first20_fma", model="orca", scenario="circle", warmup=0, steps=20, phase_key="first20", orca_math="fma",
             title="cfg4_first20 with cs_worlds.orca_math = CS_ORCA_MATH_FMA (opt-in: v_rcp / v_sqrt / v_rsq, mul + fma)"),
        dict(base, name="cfg4_dense_fma", model="orca", scenario="circle", warmup=25, steps=20, phase_key="dense", orca_math="fma",
             title="cfg4_dense with cs_worlds.orca_math = CS_ORCA_MATH_FMA (opt-in)"),
        dict(base, name="cfg5", agents=50, scenario="circle", walls=True, static=3, worlds=8192, total_worlds=65536, device_generator=True,
             title="65536 worlds (whole job, strong split) x 50-agent HSFM circle crossing R=14 (generators.static_obstacle_crossing: "
                   "the worlds tests/test_gpu_fullsize.py checks), 3 immobile humans + 3 polygon walls"),
        dict(base, name="cfg5_shard", agents=50, scenario="circle", walls=True, static=3, worlds=8192, device_generator=True,
             title="8192 worlds/GPU x 50-agent HSFM + 3 immobile humans + 3 polygon walls: ONE GPU's shard of cfg5 (what each of 8 ranks runs), Gym steps 20-70"),
        dict(base, name="moussaid", model="hsfm_new_moussaid", title="4096 worlds/GPU x 25-agent hsfm_new_moussaid hybrid scenario (the Moussaid pair force: 3 of the 9 model types)"),
        dict(base, name="cfg3_new_guo", model="hsfm_new_guo", title="4096 worlds/GPU x 25-agent hsfm_new_guo hybrid scenario (SURVEY.md §8d cfg3's second model)"),
        dict(base, name="robot26", robot=True, title="4096 worlds/GPU x 25-agent hsfm_farina hybrid scenario + a VISIBLE robot (26 rows per world), constant action"),
        dict(base, name="n30", agents=30, title="4096 worlds/GPU x 30-agent hsfm_farina hybrid scenario"),
        # crowd sizes OFF the table of compile-time row counts (6, 10, 11, 20, 25, 26, 30, 50, 51): the run-time-row build
        dict(base, name="n15", agents=15, title="4096 worlds/GPU x 15-agent hsfm_farina hybrid scenario (no shape-specialised build: run-time row count, 4 worlds per wavefront)"),
        dict(base, name="n40", agents=40, title="4096 worlds/GPU x 40-agent hsfm_farina hybrid scenario (no shape-specialised build: run-time row count, 1 world per wavefront)"),
        dict(base, name="peragent", per_agent=True, title="4096 worlds/GPU x 25-agent hsfm_farina hybrid scenario, per-agent parameters (all_params_equal = False)"),
    ]


def shard_of(spec, rank, world_size):
    from social_navigation_pyenvs_amd.sharding import world_shard

    return world_shard(rank, world_size, spec["worlds"], spec["total_worlds"])


def host_worlds(spec, rank, world_size):
    """Host arrays of this rank's shard (CPU only: also what tests/test_sharding_gloo.py runs under gloo).  Worlds are functions
    of (seed, GLOBAL world id): the batch is the same whatever the number of ranks."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.sharding import shard_seed

    first, W = shard_of(spec, rank, world_size)
    n, model = spec["agents"], spec["model"]
    seed0 = shard_seed(first)
    hmodel = "sfm_helbing" if model == "orca" else model
    respawn_bounds = respawn_worlds = None
    walls = sc.polygon_walls() if spec["walls"] else None
    if spec["scenario"] == "hybrid":
        S, goals, P, respawn_bounds = sc.hybrid_worlds(W, n, hmodel, seed0=seed0, first_world=first)
        respawn_worlds = ((first + np.arange(W)) % 2 == 1).astype(np.int32)
    elif spec["scenario"] == "circle":
        radius = 7.0 if n <= 30 else 7.0 * n / 25.0
        if spec["static"] > 0:
            S, goals, P, _ = sc.static_obstacle_worlds(W, n, hmodel, radius=radius, seed0=seed0, first_world=first, n_static=spec["static"])
        else:
            pos, yaw, goals = sc.circular_crossing(W, n, radius, seed0, first_world=first)
            S, P = sc.make_states(pos, yaw, goals), np.tile(sc.default_params(hmodel), (n, 1))
    else:
        pos, yaw, goals = sc.parallel_traffic(W, n, seed0=seed0, first_world=first)
        S, P = sc.make_states(pos, yaw, goals), np.tile(sc.default_params(hmodel), (n, 1))
        respawn_bounds = (7.0, 1.5)
    margin = None
    if model == "orca":  # cfg4: RVO2 agents, radius + 0.01, preferred velocity in columns 5:7 (unit vector to the goal)
        P = None
        d = goals[:, :, 0] - S[:, :, 0:2]
        S[:, :, 5:7] = d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
        margin = np.full(S.shape[:2], 0.01)
    if spec.get("per_agent"):   # every human its own parameters (+-5 %, a function of the global world id): the all-partners loop
        from social_navigation_pyenvs_amd.scenarios import _u01

        wid = np.arange(first, first + W, dtype=np.uint64)
        with np.errstate(over="ignore"):
            jit = np.stack([np.stack([_u01(seed0 + 11, wid, i, k, 0) for k in range(20)], -1) for i in range(n)], 1)   # [W, n, 20]
        P = np.asarray(P)[None] * (0.95 + 0.1 * jit)
    robot = action = None
    if spec.get("robot"):       # a visible robot crossing the scene on the y axis (last state row of every world, moved by its action)
        robot = np.zeros((W, 13))
        robot[:, 0:2] = (0.0, -4.0); robot[:, 2] = np.pi / 2; robot[:, 8] = 0.3; robot[:, 9] = 80.0; robot[:, 10:12] = (0.0, 4.0); robot[:, 12] = 1.0
        S = np.concatenate([S, robot[:, None, :]], axis=1)
        action = np.tile(np.array([0.0, 0.4]), (W, 1))
    return dict(first=first, W=W, S=S, goals=goals, P=P, walls=walls, margin=margin, respawn_bounds=respawn_bounds,
                respawn_worlds=respawn_worlds, robot=robot, action=action, all_params_equal=not spec.get("per_agent"))


def build_worlds(spec, rank, world_size):
    """(CrowdWorlds, host arrays or None, worlds of this rank)"""
    from social_navigation_pyenvs_amd._lib import DeviceBuffer
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    n, model = spec["agents"], spec["model"]
    if spec.get("device_generator"):
        from social_navigation_pyenvs_amd import generators as gen

        first, W = shard_of(spec, rank, world_size)
        cw = gen.static_obstacle_crossing(W, n, model, first_world=first, radius=14.0, n_static=spec["static"], walls=spec["walls"], layout=spec["layout"])
        cw.bench_action = None
        return cw, None, W
    h = host_worlds(spec, rank, world_size)
    cw = CrowdWorlds(h["S"], h["goals"], h["P"], h["margin"], h["walls"], type=model, all_params_equal=h["all_params_equal"],
                     respawn_bounds=h["respawn_bounds"], respawn_worlds=h["respawn_worlds"], layout=spec["layout"],
                     robot_row=h["robot"] is not None, robot=h["robot"], orca_math=spec.get("orca_math") or "default")
    cw.bench_action = None if h["action"] is None else DeviceBuffer.from_numpy(h["action"])
    return cw, h, h["W"]


_CPU_CORES = None


def host_cpu_info() -> dict:
    """CPU model string and logical CPU count of this box (a CPU figure states both)"""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count()}


def cpu_baseline(spec, seconds=12.0, single_seconds=4.0):
    """The C oracle (port of the reference's array kernel; ORCA: the RVO2 restatement) on this box's host cores, on a bounded sample of
    the SAME workload: the first `sample` worlds of the configuration's host generator, blocks of `substeps` substeps.  All cores
    (OpenMP over worlds) and one.  SFM / HSFM: f64 like the reference, stepped in place from the reset (the all-pairs cost does not follow
    the crowd's state).  ORCA: float32 like RVO2, advanced (untimed) to the Gym step the GPU window starts at -- its cost follows the
    crossing -- then exactly the window's steps per repetition, every repetition from that state."""
    from oracle import crowd_oracle as orc
    from social_navigation_pyenvs_amd.batched import HUMAN_MODELS as SFMS

    orc.build()
    global _CPU_CORES
    if _CPU_CORES is None:   # (a single-core leg sets OpenMP's thread count for the process: ask once, before any)
        _CPU_CORES = min(orc.num_threads(), orc.effective_cores())
    cores = _CPU_CORES
    sample = min(spec["worlds"], 32 * max(1, cores))
    host = host_worlds(dict(spec, worlds=sample, total_worlds=None, device_generator=False), 0, 1)
    n, nsub, dt = spec["agents"], spec["substeps"], spec["dt"]
    info = host_cpu_info()
    if spec["model"] == "orca":
        S0, g0, margin = host["S"].astype(np.float32), host["goals"].astype(np.float32), host["margin"].astype(np.float32)
        for _ in range(int(spec.get("warmup", 0))):
            S0, g0, _ = orc.orca_step_block(S0, g0, margin, dt, nsub, threads=cores)
        steps = int(spec.get("steps", 20))

        def timed(sel, threads, secs):
            done, reps, t0 = 0, 0, time.perf_counter()
            while True:
                s_, g_ = S0[:sel], g0[:sel]
                for _ in range(steps):
                    s_, g_, _ = orc.orca_step_block(s_, g_, margin[:sel], dt, nsub, threads=threads)
                done += sel * nsub * steps
                reps += 1
                if time.perf_counter() - t0 >= secs:
                    break
            el = time.perf_counter() - t0
            return done * n / el, reps * steps, el
        what = f"float32 RVO2 restatement (oracle/orca_oracle.c), Gym steps {spec.get('warmup', 0)}..{spec.get('warmup', 0) + steps}"
    else:
        type_id = SFMS.index(spec["model"])
        respawn = host["respawn_bounds"] is not None
        rp = (host["respawn_bounds"][0], host["respawn_bounds"][1], 0.0) if respawn else (0.0, 0.0, 0.0)

        def timed(sel, threads, secs):
            idx = np.arange(sel)
            groups = [(idx, respawn)] if host["respawn_worlds"] is None else \
                [(idx[host["respawn_worlds"][idx] == 1], True), (idx[host["respawn_worlds"][idx] == 0], False)]
            peq = bool(host.get("all_params_equal", True))
            P_of = (lambda g_: host["P"]) if np.asarray(host["P"]).ndim == 2 else (lambda g_: np.asarray(host["P"])[g_])
            runners = [orc.StepBlockRunner(type_id, host["S"][g_], host["goals"][g_], host["walls"], P_of(g_),
                                           np.zeros((g_.size, host["S"].shape[1])), peq, respawn=rs, respawn_par=rp,
                                           dtype=np.float64, threads=threads,
                                           robot=None if host.get("robot") is None else host["robot"][g_],
                                           action=None if host.get("action") is None else host["action"][g_]) for g_, rs in groups if g_.size]
            done, reps, t0 = 0, 0, time.perf_counter()
            while True:
                for r in runners:
                    r.run(dt, nsub)
                    done += r.W * nsub
                reps += 1
                if time.perf_counter() - t0 >= secs:
                    break
            el = time.perf_counter() - t0
            return done * n / el, reps, el
        what = "f64 C oracle (oracle/sfm_step.inc)"
    v_all, reps, el = timed(sample, cores, seconds)
    out = {"value": v_all, "unit": "agent-substeps/s", "cores": cores, "kind": "port", "cpu_model": info["cpu_model"], "nproc": info["nproc"]}
    tail = ""
    if single_seconds > 0:
        v_one, reps1, el1 = timed(min(sample, 64), 1, single_seconds)
        out["single_core_value"] = v_one
        tail = f"; single-core leg: {min(sample, 64)} worlds x {reps1 * nsub} substeps, {el1:.1f} s"
    out["sample"] = f"{sample} worlds x {n} agents x {reps * nsub} substeps, {what}, OpenMP over worlds on {cores} threads, {el:.1f} s{tail}"
    return out


class Snapshot:
    """The mutable buffers of a batch (state rows, goal lists, robot rows) as they stand after the warm-up; restore() copies them
    back on the launch stream (device to device, outside the timed region) so that every timed replay steps the same trajectory."""

    def __init__(self, cw, stream):
        from social_navigation_pyenvs_amd import _lib

        self.lib, self.stream, self.pairs = _lib.load(), stream, []
        for buf in (cw.d_state, cw.d_goals, cw.d_robot):
            if buf is None:
                continue
            keep = _lib.DeviceBuffer((max(buf.nbytes // 4, 1),))
            self.pairs.append((buf, keep))
            _lib.check(self.lib.cs_memcpy_d2d(C.c_void_p(keep.ptr), C.c_void_p(buf.ptr), C.c_size_t(buf.nbytes), C.c_void_p(stream)))
        _lib.stream_sync(stream)

    def restore(self):
        from social_navigation_pyenvs_amd import _lib

        for buf, keep in self.pairs:
            _lib.check(self.lib.cs_memcpy_d2d(C.c_void_p(buf.ptr), C.c_void_p(keep.ptr), C.c_size_t(buf.nbytes), C.c_void_p(self.stream)))

    def free(self):
        for _, keep in self.pairs:
            keep.free()


class Runner:
    """Barrier / timing plumbing shared by the main workload and the other configurations."""

    def __init__(self, torch, dist, stream, reduce_device="cuda"):
        self.torch, self.dist, self.stream, self.reduce_device = torch, dist, stream, reduce_device

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, cw, spec, steps, warmup, repeats, eager=False, restore=True):
        """Times `repeats` x (exactly `steps` steps between barrier + synchronize on both sides), every repeat from the state the
        `warmup` steps left (restored outside the timed region).  Returns the per-repeat wall times (MAX over ranks) and the
        HIP-event kernel time per step of every repeat (this rank) -- the same samples."""
        from social_navigation_pyenvs_amd import _lib
        from social_navigation_pyenvs_amd.sharding import max_over_ranks

        cw.stream = self.stream
        dt, n_sub = spec["dt"], spec["substeps"]
        action = getattr(cw, "bench_action", None)
        for _ in range(warmup):
            cw.step(dt, n_sub, action)
        self.barrier()
        snap = Snapshot(cw, self.stream) if restore else None
        wall, kern = [], []
        if eager:
            # one launch per step from Python, a HIP event pair around every launch
            for _ in range(repeats):
                starts = [_lib.Event() for _ in range(steps)]
                stops = [_lib.Event() for _ in range(steps)]
                if snap:
                    snap.restore()
                self.barrier()
                t0 = time.perf_counter()
                for k in range(steps):
                    starts[k].record(self.stream)
                    cw.step(dt, n_sub, action)
                    stops[k].record(self.stream)
                _lib.stream_sync(self.stream)
                self.barrier()
                wall.append(time.perf_counter() - t0)
                kern.append(float(np.mean([starts[k].elapsed_ms(stops[k]) for k in range(steps)])))
        else:
            # the K steps are captured once into a HIP graph (untimed) and replayed with ONE launch per timed region:
            # no per-launch host gap; HIP events on the launch stream bracket the K kernels of every replay
            with _lib.Graph.capture(self.stream) as graph:
                for k in range(steps):
                    cw.step(dt, n_sub, action)
            graph.launch()                      # one untimed replay (first launch of an instantiated graph uploads it)
            _lib.stream_sync(self.stream)
            e0, e1 = _lib.Event(), _lib.Event()
            for _ in range(repeats):
                if snap:
                    snap.restore()              # untimed: the replay below starts from the post-warm-up worlds every time
                self.barrier()
                t0 = time.perf_counter()
                e0.record(self.stream)
                graph.launch()
                e1.record(self.stream)
                _lib.stream_sync(self.stream)
                self.barrier()
                wall.append(time.perf_counter() - t0)
                kern.append(e0.elapsed_ms(e1) / steps)
        if snap:
            snap.free()
        dev = self.reduce_device if self.dist is not None else None
        wall = [max_over_ranks(w, self.dist, device=dev) for w in wall]
        return np.array(wall), np.array(kern)

    def gather(self, value: float) -> list:
        if self.dist is None or self.dist.get_world_size() == 1:
            return [float(value)]
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.reduce_device or "cpu")
        out = [self.torch.zeros_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, t)
        return [float(x.item()) for x in out]

    def gather_objects(self, obj) -> list:
        """every rank's (small, JSON-able) record, in rank order: one all_gather of fixed-size byte tensors on the device the
        backend reduces on -- the collective the timing path already uses, no object (pickle) collective on RCCL"""
        if self.dist is None or self.dist.get_world_size() == 1:
            return [obj]
        raw = json.dumps(obj).encode()
        size = 512
        if len(raw) > size:
            raise ValueError("rank record too large")
        dev = self.reduce_device or "cpu"
        t = self.torch.zeros(size, dtype=self.torch.uint8, device=dev)
        t[:len(raw)] = self.torch.tensor(list(raw), dtype=self.torch.uint8, device=dev)
        out = [self.torch.zeros_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, t)
        return [json.loads(bytes(x.cpu().tolist()).rstrip(b"\0").decode()) for x in out]

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        """Device-to-device copy bandwidth measured in this run (read + written bytes / time): the practical HBM ceiling
        (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy) beside the 8 TB/s vendor peak."""
        from social_navigation_pyenvs_amd import _lib

        a, b = _lib.DeviceBuffer((nbytes // 4,)), _lib.DeviceBuffer((nbytes // 4,))
        lib = _lib.load()
        _lib.check(lib.cs_memset(C.c_void_p(a.ptr), C.c_int(1), C.c_size_t(nbytes), C.c_void_p(self.stream)))
        for _ in range(2):
            _lib.check(lib.cs_memcpy_d2d(C.c_void_p(b.ptr), C.c_void_p(a.ptr), C.c_size_t(nbytes), C.c_void_p(self.stream)))
        e0, e1 = _lib.Event(), _lib.Event()
        best = None
        for _ in range(reps):
            e0.record(self.stream)
            _lib.check(lib.cs_memcpy_d2d(C.c_void_p(b.ptr), C.c_void_p(a.ptr), C.c_size_t(nbytes), C.c_void_p(self.stream)))
            e1.record(self.stream)
            ms = e0.elapsed_ms(e1)
            best = ms if best is None else min(best, ms)
        a.free(); b.free()
        return 2.0 * nbytes / (best * 1e-3) / 1e9


def family_of(model):
    return "orca" if model == "orca" else ("hsfm" if model.startswith("hsfm") else "sfm")


def build_id() -> str:
    from social_navigation_pyenvs_amd import _lib

    lib = _lib.load()
    lib.cs_build_id.restype = C.c_char_p
    return lib.cs_build_id().decode()


def pmc_summary() -> dict:
    try:
        return json.load(open(PMC_SUMMARY))
    except Exception:
        return {}


def roofline_block(spec, W, kern_ms, copy_gbs, cw, window=None):
    n, n_sub = spec["agents"], spec["substeps"]
    fam = family_of(spec["model"])
    alg = ALG_BYTES[fam] * W * n * n_sub          # walls are shared by all worlds: 0 B per world-substep
    k_avg = float(np.mean(kern_ms))
    achieved = alg / (k_avg * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "meaning": "normalised ALGORITHMIC throughput (bytes one substep would move if the state round-tripped HBM every "
                      "substep) -- the fused launch keeps the state in registers / LDS; see traffic / valu_frac for what the "
                      "hardware really does",
           "frac_vs_copy": (achieved / copy_gbs) if copy_gbs else None, "copy_GBs_measured": copy_gbs,
           "kernel": "k_orca_step" if fam == "orca" else "k_sfm_step", "variant": cw.step_variant(),
           "kernel_avg_ms": k_avg, "kernel_median_ms": float(np.median(kern_ms)), "kernel_min_ms": float(np.min(kern_ms)),
           "kernel_max_ms": float(np.max(kern_ms)), "algorithmic_bytes_per_launch": alg, "bytes_per_agent_substep": ALG_BYTES[fam]}
    summary = pmc_summary()
    pm = summary.get(spec_key(spec)) or {}
    per_agent = pm.get("hbm_bytes_per_agent_launch")
    out["traffic"] = per_agent * W * n if per_agent is not None else None
    out["traffic_source"] = pm.get("traffic_source")
    out["traffic_GBs"] = (out["traffic"] / (k_avg * 1e-3) / 1e9) if out["traffic"] is not None else None
    # the PMC figures are read back from the round's committed counter passes: say whether those ran on THIS library build
    out["pmc_build_id"] = pm.get("build_id", summary.get("_build_id"))
    out["pmc_build_matches"] = (out["pmc_build_id"] == build_id()) if pm else None
    if fam != "orca":
        soc = "guo" if spec["model"].endswith("guo") else ("moussaid" if spec["model"].endswith("moussaid") else "helbing")
        pair_flops = PAIR_FLOPS[soc] * W * n * (n - 1) * n_sub   # the reference's count: every ordered pair
        out["fp32_tflops_equiv"] = pair_flops / (k_avg * 1e-3) / 1e12
        out["fp32_frac"] = out["fp32_tflops_equiv"] / FP32_PEAK_TFLOPS
    out["valu"] = pm.get("valu")          # measured VALU issue figures (the true bound of this kernel), tagged with their source
    # a kernel's instruction count follows the crowd's state (cfg5: how many polygons the wave vote skips): where the counters were
    # also taken over THIS timed window (warmup, steps), those are the ones paired with this kernel time
    pmw = summary.get(f"{spec_key(spec)}@w{window[0]}s{window[1]}") if window else None
    if pmw and pmw.get("valu"):
        out["valu"] = pmw["valu"]
        out["valu_window"] = f"Gym steps {window[0]}..{window[0] + window[1]} (the timed window)"
    out["valu_frac"] = out["valu_frac_floor"] = None
    if out["valu"] and out["valu"].get("valu_insts_per_wave_substep"):
        scale = out["valu"]["waves_per_launch"] * n_sub * (W / float(pm.get("worlds", W)))
        insts = out["valu"]["valu_insts_per_wave_substep"] * scale
        simd_cycles = SIMDS * k_avg * 1e-3 * CLOCK_HZ
        mix = out["valu"].get("mix_per_wave_substep")
        if mix:   # SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 of the same passes: the classes the microbenchmark priced
            fma = (mix.get("add_f32", 0) + mix.get("mul_f32", 0) + mix.get("fma_f32", 0)) * scale
            trans = mix.get("trans_f32", 0) * scale
            other = max(0.0, insts - fma - trans)
            cyc = fma * VALU_CLASS_CYCLES["fma"] + trans * VALU_CLASS_CYCLES["trans"] + other * VALU_CLASS_CYCLES["other"]
            out["valu_issue_cycles_per_inst_model"] = cyc / insts
            out["valu_frac"] = cyc / simd_cycles
            out["valu_frac_floor"] = ((fma + other) * VALU_CLASS_CYCLES["fma"] + trans * VALU_CLASS_CYCLES["trans"]) / simd_cycles
            # the same model against the cycles the wavefronts really lived (SQ_WAVE_CYCLES of the same passes) instead of kernel time x
            # 2.4 GHz: independent of the shader clock, which drops under a long vector-heavy kernel (ORCA: ~1.8 GHz by this very ratio)
            wq, wps = out["valu"].get("wave_quadcycles_per_substep"), out["valu"].get("waves_per_simd")
            if wq and wps and wps <= 2.0:     # (every wavefront of the launch resident at once: every build here holds two per SIMD)
                per_wave = cyc / (out["valu"]["waves_per_launch"] * n_sub * (W / float(pm.get("worlds", W))))
                out["valu_frac_wave_cycles"] = per_wave * float(wps) / (4.0 * wq)
                out["shader_clock_ghz_est"] = 4.0 * wq * n_sub / (k_avg * 1e-3) / 1e9   # (a lower bound: the launch is longer than its wavefronts live)
        else:     # no class counters for this configuration yet: every instruction at the 4.1 of the compare / select / min-max class
            out["valu_frac"] = insts * VALU_CLASS_CYCLES["other"] / simd_cycles
        out["valu_cycles_per_inst_achieved"] = simd_cycles / insts
        out["valu_frac_meaning"] = ("SIMD issue cycles the launch's vector instructions need (PMC instruction counts by class x the MEASURED cost of the class at >= 2 "
                                    "wavefronts per SIMD: fma/mul/add 2.2, transcendental 8.2, everything else 4.1; profiles/archive/r5_valu_issue_ceiling.txt) / (1024 SIMDs x kernel "
                                    "time x 2.4 GHz).  valu_frac_floor prices every non-transcendental instruction at 2.2 (no SIMD can do better).  Without class "
                                    "counters: all at 4.1")
    return out


# --------------------------------------------------------------------------------------------------------------------------
# the stdout line: compact by construction.  BENCH_r03 lost its headline because the ONE line had grown to 22.6 KB (eight
# other_configs entries with a full roofline block and two prose strings each): the driver keeps an 8 KB tail.  Everything the
# contract names stays in the line; what explains it lives in the full file.
# --------------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def _r(x, sig=5):
    """floats to `sig` significant digits (a bench line is read by people and parsed by the driver: 17 digits help neither)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _wg_of(variant: str) -> int:
    """one-wavefront blocks per workgroup as cs_step_variant names it (" wg=4"); 1 where the block is the workgroup"""
    import re

    m = re.search(r"\bwg=(\d+)", variant or "")
    return int(m.group(1)) if m else 1


def short_variant(v: str) -> str:
    """'k_sfm_step<SOC=0,HEADED=1,PEQ=1,MAXT=64,OCC=1,ROWS_CT=25,LEAN=1> grid=2048 block=64 wpb=2' -> 'sfm<0,1,1,64,1,25,1>g2048'"""
    import re

    if not v:
        return v
    m = re.match(r"k_(\w+?)(?:_step)?(_row16)?<([^>]*)>(?:\s+grid=(\d+))?", v)
    if not m:
        return v[:40]
    vals = ",".join(kv.split("=")[-1] for kv in m.group(3).split(","))
    math = re.search(r"math=(\w+)", v)            # the ORCA build's arithmetic (exact / fast / fma)
    return f"{m.group(1)}{m.group(2) or ''}<{vals}>" + (f"g{m.group(4)}" if m.group(4) else "") + (f":{math.group(1)}" if math else "")


def compact_roofline(rl: dict) -> dict:
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu_frac", "kernel", "kernel_avg_ms", "algorithmic_bytes_per_launch",
            "bytes_per_agent_substep", "frac_vs_copy", "pmc_build_matches")
    out = {k: rl.get(k) for k in keep}
    out["variant"] = short_variant(rl.get("variant"))
    v = rl.get("valu") or {}
    out["valu_per_wave_substep"] = v.get("valu_insts_per_wave_substep")
    out["lds_per_wave_substep"] = v.get("lds_insts_per_wave_substep")
    return out


def compact_line(full: dict) -> str:
    """The ONE stdout line from the full result: headline keys, config, ONE roofline (the headline's), cpu_baseline, ranks, the Gym
    step figures and other_configs as a table.  Raises if the line would not fit LINE_LIMIT bytes or is not strict JSON."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "timing_repeats", "kernel_us", "kernel_le_wall", "gym_steps_per_s", "finite_fraction", "build_id")
    out = {k: full.get(k) for k in keys}
    cfg = full.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in ("workload", "worlds_per_gpu", "worlds_total", "agents", "substeps_per_step", "parallelism", "launch", "stationary")}
    out["roofline"] = compact_roofline(full["roofline"])
    if full.get("cpu_baseline"):
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "single_core_value", "cpu_model", "nproc")}
        out["gpu_over_cpu"] = full.get("gpu_over_cpu")
    out["dist"] = full.get("dist")
    out["ranks"] = full.get("ranks")
    if full.get("gym_step"):
        out["gym_step"] = full["gym_step"]
    cols = ("name", "ms_per_step", "kernel_us", "frac", "valu_frac", "variant", "pmc_build_matches", "worlds_total", "steps", "warmup", "cpu_value")
    rows = []
    for o in full.get("other_configs") or []:
        rl = o.get("roofline") or {}
        rows.append([o.get("name"), _r(o.get("ms_per_step"), 4), _r(o.get("kernel_us"), 4), _r(o.get("frac"), 3), _r(o.get("valu_frac"), 3), short_variant(rl.get("variant")),
                     rl.get("pmc_build_matches"), o.get("worlds_total"), o.get("steps"), o.get("warmup"), _r(o.get("cpu_value"), 3)])
    out["other_configs"] = {"columns": list(cols), "rows": rows, "workloads": "BASELINE.json configs[1],[3],[4] + cfg5's per-GPU shard + the shapes around configs[2]; cpu_value = the C oracle on all host cores (agent-substeps/s); titles in full_json"}
    out["full_json"] = full.get("full_json")
    line = json.dumps(_r(out), allow_nan=False, separators=(",", ":"))
    if len(line.encode()) >= LINE_LIMIT:
        # never print a line the driver cannot parse: drop the optional blocks, largest first
        for k in ("other_configs", "gym_step", "ranks"):
            if k == "ranks" and out.get("ranks"):
                out["ranks"] = [{kk: r.get(kk) for kk in ("rank", "device", "kernel_us")} for r in out["ranks"]]
            elif k in out:
                out[k] = {"dropped": "line limit; see full_json"}
            line = json.dumps(_r(out), allow_nan=False, separators=(",", ":"))
            if len(line.encode()) < LINE_LIMIT:
                break
    if len(line.encode()) >= LINE_LIMIT:
        raise RuntimeError(f"bench line is {len(line)} bytes (limit {LINE_LIMIT})")
    json.loads(line)
    return line


# --------------------------------------------------------------------------------------------------------------------------
# `--gpus N` without a launcher: start the N ranks here
# --------------------------------------------------------------------------------------------------------------------------
def visible_gpus() -> int:
    """Devices a fresh process would see, counted in a CHILD: this process must not initialise HIP before it starts its ranks."""
    import subprocess

    code = ("import sys; sys.path.insert(0, %r); from social_navigation_pyenvs_amd import _lib; print(_lib.device_count())" % ROOT)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


RANK_EXIT_TIMEOUT_S = 60.0   # how long the other ranks may take to exit after rank 0 has (teardown, the final barrier)


def free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv, worker=None, n_visible=None) -> int:
    """Start `args.gpus` ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, one GPU each), relay
    rank 0's line to stdout and return the worst exit code.  Nothing here imports torch or calls HIP.  `worker` / `n_visible`: the CPU
    test's stub (tests/test_bench_line_cpu.py)."""
    import subprocess

    n = args.gpus
    have = visible_gpus() if n_visible is None else n_visible
    need = 1 if args.same_device else n
    if have < need:
        print(f"bench.py: --gpus {n} needs {need} visible GPU(s), this box shows {have}: not measuring fewer ranks than asked for", file=sys.stderr)
        return 3
    port = free_port()
    cmd = [sys.executable, worker or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), CROWDSTEP_BENCH_LAUNCHER="self")
        # rank 0's stdout is the line; the other ranks print nothing there (and whatever they do print goes to stderr)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # a rank that dies (no GPU left, a fault) would leave the others waiting in the rendezvous: when one exits non-zero the rest of
    # OUR ranks (the exact processes started above) are ended instead of hanging until a collective times out
    out0 = None
    while out0 is None:
        try:
            out0, _ = procs[0].communicate(timeout=1.0)
        except subprocess.TimeoutExpired:
            if any(p.poll() not in (None, 0) for p in procs[1:]):
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                out0, _ = procs[0].communicate()
    # rank 0 is through: its line goes out NOW -- a rank that hangs in its teardown must not cost the measurement
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    if lines:
        print(lines[-1], flush=True)
    codes = [procs[0].returncode]
    deadline = time.monotonic() + RANK_EXIT_TIMEOUT_S
    for p in procs[1:]:      # the exact processes started above, each waited for within ONE overall deadline, then ended
        try:
            codes.append(p.wait(timeout=max(0.1, deadline - time.monotonic())))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            codes.append(-9)
            print(f"bench.py: a rank did not exit within {RANK_EXIT_TIMEOUT_S} s of rank 0 and was ended", file=sys.stderr)
    worst = max((abs(c) for c in codes), default=0)
    if worst != 0:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
    if not lines and worst == 0:
        worst = 4
    return worst


def gym_step_figures(W, n, steps=150):
    """us per batched step of the device-resident Gym loop (BatchedSocialNavGym.step_device: swept collision + reward + bookkeeping, 20
    fused substeps + observation, regeneration of the worlds whose episode ended) at W hybrid worlds x n humans, from inside the
    library's stream: without resets / same-step auto-reset / NEXT_STEP mode (tools/device_loop_bench.py prints the longer table)."""
    import configparser

    import torch

    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        "env": {"time_limit": 50, "time_step": 0.0125, "robot_time_step": 0.25, "val_size": 100, "test_size": 500, "randomize_attributes": "false"},
        "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
        "sim": {"train_val_sim": "hybrid_scenario", "test_sim": "hybrid_scenario", "square_width": 10, "circle_radius": 7, "human_num": n,
                "traffic_length": 14, "traffic_height": 3},
        "humans": {"visible": "true", "policy": "hsfm_farina", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
        "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
    })
    out = {"worlds": W, "humans": n, "unit": "us per batched Gym step", "steps": steps}
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    for key, mode in (("no_reset", False), ("same_step", True), ("next_step", "next_step")):
        env = BatchedSocialNavGym(cfg, W)
        env.reset(phase="train", first_case=0, device=True)
        buf = env.action_buffer()
        buf.copy_(torch.randn(W, 2, device="cuda", generator=gen) * 0.5)
        with torch.cuda.stream(env.device_stream()):
            for _ in range(60):                      # past the first episode ends: resets are part of the steady state
                env.step_device(buf, auto_reset=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                env.step_device(buf, auto_reset=mode)
            torch.cuda.synchronize()
            out[key] = (time.perf_counter() - t0) / steps * 1e6
        if mode is not False and hasattr(env, "failed_resets"):
            try:
                out[key + "_failed_resets"] = int(env.failed_resets())
            except Exception:
                pass
        del env
    try:    # the drop-in single-environment facade: what ONE SocialNavGym.step() costs at W = 1 (tools/facade_latency.py has the split)
        out.update(facade_w1_latency(cfg, n))
    except Exception as e:
        out["facade_w1_error"] = f"{type(e).__name__}: {e}"[:120]
    return out


def facade_w1_latency(cfg, n, steps=200):
    """us per SocialNavGym.step(ActionXY) and per MotionModelManager.get_next_human_observable_states() (the SARL-style peek) of the W = 1 facade
    -- host Python, the uploads / downloads of the host mirrors and the launches included (social_nav_gym.py:227-250; BASELINE.md: 227 ms per
    reference Gym step at N = 25)."""
    import types

    from social_navigation_pyenvs_amd.crowd_nav.utils.action import ActionXY
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import SocialNavGym
    from social_navigation_pyenvs_amd.social_gym.src.robot_agent import RobotAgent

    env = SocialNavGym()
    env.configure(cfg)
    robot = RobotAgent(env)
    robot.visible, robot.desired_speed, robot.radius, robot.sensor = False, 1.0, 0.3, "coordinates"
    robot.policy = types.SimpleNamespace(multiagent_training=True, with_theta_and_omega_visible=False, kinematics="holonomic", name="bench", query_env=True, time_step=None)
    robot.kinematics = "holonomic"
    env.set_robot(robot)
    env.reset(phase="test", test_case=1)
    rng = np.random.default_rng(n)
    t_step = t_peek = 0.0
    for k in range(steps + 20):
        if k % 60 == 0:
            env.reset(phase="test", test_case=1 + k // 60)
        a = rng.uniform(-0.5, 0.5, 2)
        t0 = time.perf_counter()
        env.motion_model_manager.get_next_human_observable_states(env.robot_time_step)
        t1 = time.perf_counter()
        env.step(ActionXY(float(a[0]), float(a[1])))
        t2 = time.perf_counter()
        if k >= 20:
            t_peek += t1 - t0; t_step += t2 - t1
    return {"facade_w1_step": t_step / steps * 1e6, "facade_w1_peek": t_peek / steps * 1e6}


def gym_step_child(W, n, timeout_s=240.0) -> dict:
    """gym_step_figures in a CHILD process with a deadline: a hang or a GPU fault in the side figure cannot take the headline with it
    (a process that has initialised the GPU may start children; it must not exec)."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gym-step-child", str(W), str(n)], capture_output=True, text=True, timeout=timeout_s)
        last = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        if r.returncode != 0 or not last:
            return {"error": f"child rc={r.returncode}: {(r.stderr or '').strip().splitlines()[-1] if (r.stderr or '').strip() else 'no output'}"[:160]}
        return json.loads(last[-1])
    except subprocess.TimeoutExpired:
        return {"error": f"gym_step child exceeded {timeout_s:.0f} s and was ended"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:160]}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 3 and argv[0] == "--gym-step-child":
        print(json.dumps(gym_step_figures(int(argv[1]), int(argv[2]))), flush=True)
        return
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher.  BEFORE torch is imported or HIP touched -- the ranks are fresh processes.
        sys.exit(launch_ranks(args, argv))
    # stdout carries ONE JSON line and nothing else: native libraries (RCCL prints a version banner from C when NCCL_DEBUG
    # asks for it) write to file descriptor 1 directly, so everything but the final line is sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_size:
        # a line whose n_gpus differs from what was asked for would be read as an N-GPU measurement: refuse
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}: start it as `python bench.py --gpus {args.gpus}` (it launches its own ranks) "
              f"or under torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    import torch

    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd.batched import HUMAN_MODELS as SFMS

    _lib.require_gpu()
    n_dev = _lib.device_count()
    dev_index = 0 if args.same_device else local_rank
    if dev_index >= n_dev:
        print(f"bench.py: rank {rank} wants GPU {dev_index}, {n_dev} visible", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(dev_index)
    _lib.set_device(dev_index)
    dist = None
    dist_info = {"backend": None, "world_size": 1, "launcher": os.environ.get("CROWDSTEP_BENCH_LAUNCHER", "none" if world_size == 1 else "external")}
    if world_size > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.force_dist and world_size == 1:   # rehearsal of the RCCL barrier path without a launcher
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
        dist_info.update(backend=dist.get_backend(), world_size=dist.get_world_size())
        if args.dist_backend == "nccl":
            try:
                dist_info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                dist_info["rccl_version"] = None

    stream = _lib.stream_create()
    run = Runner(torch, dist, stream, reduce_device="cuda" if args.dist_backend == "nccl" else None)
    spec = workload_spec(args)
    cw, host, W = build_worlds(spec, rank, world_size)
    wall, kern = run.measure(cw, spec, args.steps, args.warmup, args.repeats, eager=args.eager, restore=not args.no_restore)
    per_rank_kernel_us = run.gather(float(np.mean(kern)) * 1e3)
    total_worlds = args.total_worlds if args.total_worlds is not None else world_size * args.worlds
    # which card every rank sat on
    me = {"rank": rank, "device": dev_index, "pci": _lib.device_pci_bus_id(dev_index), "worlds": W, "kernel_us": float(np.mean(kern)) * 1e3}
    ranks = run.gather_objects(me)

    # sanity: the state is finite and moved
    S_end = cw.get_states()
    finite = float(np.mean(np.isfinite(S_end[..., :8])))

    copy_gbs = run.copy_bandwidth() if rank == 0 else None
    others = []
    if not args.no_other_configs:
        for ospec in other_config_specs(args):
            ocw, _, oW = build_worlds(ospec, rank, world_size)
            k_o = ospec["steps"]   # every row over its OWN window, whatever --steps says: the driver's line and profiles/ show the same figures
            r_o = 7
            owall, okern = run.measure(ocw, ospec, k_o, ospec["warmup"], r_o, restore=not args.no_restore)
            ous = run.gather(float(np.mean(okern)) * 1e3)
            if rank == 0:
                tot = ospec["total_worlds"] if ospec["total_worlds"] is not None else world_size * ospec["worlds"]
                med = float(np.median(owall))
                rl = roofline_block(ospec, oW, okern, copy_gbs, ocw, window=(ospec["warmup"], k_o))
                others.append({"name": ospec["name"], "workload": ospec["title"], "scaling": "strong" if ospec["total_worlds"] else "weak",
                               "worlds_this_rank": oW, "worlds_total": tot, "warmup": ospec["warmup"], "steps": k_o, "timing_repeats": r_o,
                               "ms_per_step": med / k_o * 1e3,
                               "value": tot * ospec["agents"] * ospec["substeps"] * k_o / med, "unit": "agent-substeps/s",
                               "kernel_us": rl["kernel_avg_ms"] * 1e3, "kernel_us_median": rl["kernel_median_ms"] * 1e3,
                               "kernel_le_wall": bool(rl["kernel_median_ms"] <= med / k_o * 1e3),
                               "per_rank_kernel_us": ous, "frac": rl["frac"], "valu_frac": rl["valu_frac"], "roofline": rl})
            del ocw

    if rank == 0:
        n_sub = args.substeps
        med, lo, hi = float(np.median(wall)), float(np.min(wall)), float(np.max(wall))
        agent_substeps = total_worlds * args.agents * n_sub * args.steps
        value = agent_substeps / med
        g, b, wpb = cw.launch_geometry()
        rl = roofline_block(spec, W, kern, copy_gbs, cw, window=(args.warmup, args.steps))
        out = {
            "metric": "env-steps/sec (worlds x agents) for HSFM 25-agent crowd",
            "value": value,
            "unit": "agent-substeps/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": med / args.steps * 1e3,
            "ms_per_step_min": lo / args.steps * 1e3,
            "ms_per_step_max": hi / args.steps * 1e3,
            "timing_repeats": int(len(wall)),
            "higher_is_better": True,
            "scaling": "strong" if args.total_worlds is not None else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "build_id": build_id(),
            "config": {
                "workload": f"{W} worlds/GPU x {args.agents}-agent {args.model} {args.scenario} scenario"
                            f"{' + 3 polygon walls' if args.walls else ''}{f' + {args.static} immobile humans' if args.static else ''}"
                            f"{' + a visible robot' if args.robot else ''}{', per-agent parameters' if args.per_agent_params else ''}, "
                            f"{n_sub} fused substeps of {args.dt} s per step (one Gym step), state resident in HBM ({args.layout})",
                "worlds_per_gpu": W, "worlds_total": total_worlds, "agents": args.agents, "substeps_per_step": n_sub,
                "motion_model": args.model, "scenario": args.scenario, "parallelism": f"worlds sharded x{world_size}, no collective",
                # (blocks of one wavefront go to the dispatcher several to a workgroup -- wg= in the variant string: grid / wg workgroups of block x wg threads)
                "launch": {"grid": g, "block": b, "worlds_per_block": wpb, "blocks_per_workgroup": _wg_of(cw.step_variant())},
                "timed_region": "eager launches, one HIP event pair per launch" if args.eager else
                                "[untimed: worlds restored to the post-warm-up snapshot] barrier + sync | ONE replay of a HIP graph holding exactly K "
                                "cs_step launches | sync + barrier; repeated R times on the SAME trajectory segment (Gym steps W .. W+K), median reported",
                "stationary": not args.no_restore,
            },
            "world_substeps_per_s": value / args.agents,
            "gym_steps_per_s": value / args.agents / n_sub,
            "finite_fraction": finite,
            "kernel_us": rl["kernel_avg_ms"] * 1e3,
            "kernel_us_median": rl["kernel_median_ms"] * 1e3,
            "kernel_le_wall": bool(rl["kernel_median_ms"] <= med / args.steps * 1e3),
            "per_rank_kernel_us": per_rank_kernel_us,
            "dist": dist_info,
            "ranks": ranks,
            "roofline": rl,
            "other_configs": others,
        }
        if not args.no_cpu_baseline and world_size == 1 and not args.robot and not args.per_agent_params:
            out["cpu_baseline"] = cpu_baseline(dict(spec, warmup=args.warmup, steps=args.steps), args.cpu_seconds)   # rank 0, N = 1 only
            out["gpu_over_cpu"] = value / world_size / out["cpu_baseline"]["value"]
            # the other configurations the oracle covers, each beside its own row (3 s of CPU work per row)
            for o in others:
                if o["name"].endswith("_fma") and o["name"][:-4] in CPU_ROWS:   # same worlds, same window: the CPU figure of the exact row
                    twin = next((x for x in others if x["name"] == o["name"][:-4] and x.get("cpu_value")), None)
                    if twin:
                        o["cpu_value"] = twin["cpu_value"]; o["gpu_over_cpu"] = o["value"] / twin["cpu_value"]
                elif o["name"] in CPU_ROWS:
                    try:
                        ospec = next(x for x in other_config_specs(args) if x["name"] == o["name"])
                        cb = cpu_baseline(ospec, seconds=3.0, single_seconds=0.0)
                        o["cpu_baseline"] = cb
                        o["cpu_value"] = cb["value"]
                        o["gpu_over_cpu"] = o["value"] / cb["value"]
                    except Exception as e:   # a side figure never costs the headline
                        o["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:160]}
        def write_full():
            try:
                os.makedirs(os.path.dirname(args.full_json), exist_ok=True)
                with open(args.full_json, "w") as f:
                    json.dump(out, f, indent=1)
                out["full_json"] = os.path.relpath(args.full_json, ROOT)
            except OSError as e:
                out["full_json"] = None
                print(f"bench.py: full result not written ({e})", file=sys.stderr)

        write_full()   # the measurement is on disk before the side figure below runs
        if not args.no_gym_step and not args.no_other_configs and world_size == 1 and args.model != "orca" and not args.robot and not args.per_agent_params and not args.walls:
            out["gym_step"] = gym_step_child(args.worlds, args.agents)
            write_full()
        print(json.dumps(out), file=sys.stderr, flush=True)      # the full blocks: stderr + the side file
        line = compact_line(out)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
