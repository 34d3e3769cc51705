#!/usr/bin/env python3
"""bench.py -- throughput of the crowd-step hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch: ONE Gym step (= `substeps` fused Euler substeps
of dt = 0.0125 s, social_nav_gym.py:240-245 / env.config:2-4) of every world resident on the GPU,
i.e. one cs_step launch.  Workload at N=1 = BASELINE.json configs[2] (the config the metric names):
4096 worlds x 25-agent Headed-SFM (hsfm_farina) hybrid scenario (half circular crossing R=7, half
14x3 m parallel traffic with respawn), synthetic random-goal crowds, state resident in HBM.
Worlds are independent: each rank owns its own 4096 worlds (weak scaling, no collective on the data
path); the only collectives are the timing barrier and the MAX over ranks.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (algorithmic
bytes per launch / average kernel duration measured with HIP events on the launch stream, against
the 8 TB/s HBM3E peak) and `cpu_baseline` (the C oracle timed on the host cores of this box).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per agent-substep, f32 (SURVEY.md §8d / BASELINE.md §4)
ALG_BYTES = {"sfm": 52, "hsfm": 76, "orca": 48}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
# HBM-side bytes per agent per LAUNCH from rocprofv3 PMC passes of this very command (profiles/r1i_pmc_traffic.txt (first measured in r1g),
# method in profiles/r1c_pmc_traffic.md): (FETCH_SIZE 5511.1 KB + WRITE_SIZE 4505.9 KB) / (4096 x 25 agents);
# independent of the number of fused substeps.  FETCH_SIZE is uncalibrated for 4-byte-per-lane loads on gfx950
# (reads 0.87x the 64 B/agent the code loads).
MEASURED_TRAFFIC_B_PER_AGENT_LAUNCH = {("hsfm_farina", "hybrid", False): (5511.07 + 4505.90) * 1024 / (4096 * 25)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--worlds", type=int, default=4096, help="worlds per GPU")
    ap.add_argument("--agents", type=int, default=25)
    ap.add_argument("--model", default="hsfm_farina")
    ap.add_argument("--scenario", default="hybrid", choices=["hybrid", "circle", "traffic"])
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--dt", type=float, default=0.0125)
    ap.add_argument("--layout", default="soa", choices=["aos", "soa"])
    ap.add_argument("--walls", action="store_true", help="add 3 shared polygon walls")
    ap.add_argument("--eager", action="store_true", help="launch every step from Python (default: one HIP graph of K steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the N > 1 code path on a one-GPU box")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def build_worlds(args, rank):
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    from social_navigation_pyenvs_amd.sharding import shard_seed, world_shard

    W, n = args.worlds, args.agents
    first, _ = world_shard(rank, int(os.environ.get("WORLD_SIZE", "1")), W)
    seed0 = shard_seed(first)
    respawn_bounds = None
    respawn_worlds = None
    if args.scenario == "hybrid":
        S, goals, P, respawn_bounds = sc.hybrid_worlds(W, n, "sfm_helbing" if args.model == "orca" else args.model, seed0=seed0)
        respawn_worlds = (np.arange(W) % 2 == 1).astype(np.int32)
    elif args.scenario == "circle":
        radius = 7.0 if n <= 30 else 7.0 * n / 25.0
        pos, yaw, g = sc.circular_crossing(W, n, radius, seed0)
        S, goals = sc.make_states(pos, yaw, g), g
        P = None if args.model == "orca" else np.tile(sc.default_params(args.model), (n, 1))
    else:
        pos, yaw, g = sc.parallel_traffic(W, n, seed0=seed0)
        S, goals = sc.make_states(pos, yaw, g), g
        P = None if args.model == "orca" else np.tile(sc.default_params(args.model), (n, 1))
        respawn_bounds = (7.0, 1.5)
    walls = sc.polygon_walls() if args.walls else None
    margin = None
    if args.model == "orca":  # cfg4: RVO2 agents, radius + 0.01, preferred velocity in columns 5:7 (unit vector to the goal)
        P = None
        d = goals[:, :, 0] - S[:, :, 0:2]
        S[:, :, 5:7] = d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
        margin = np.full(S.shape[:2], 0.01)
    cw = CrowdWorlds(S, goals, P, margin, walls, type=args.model, all_params_equal=True,
                     respawn_bounds=respawn_bounds, respawn_worlds=respawn_worlds, layout=args.layout)
    host = dict(S=S, goals=goals, P=P, walls=walls, respawn_bounds=respawn_bounds, respawn_worlds=respawn_worlds)
    return cw, host


def effective_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants 16 through cpu.max; more OpenMP threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(args, host, type_id):
    """The C oracle (port of the reference's f64 array kernel) on this box's host cores, on a bounded
    sample of the same workload: the first `sample_worlds` worlds, blocks of `substeps` substeps, stepped in
    place (no host copies in the timed loop).  Run on all cores (OpenMP over worlds) and on one."""
    from oracle import crowd_oracle as orc

    orc.build()
    cores = min(orc.num_threads(), effective_cores())
    respawn = host["respawn_bounds"] is not None
    rp = (host["respawn_bounds"][0], host["respawn_bounds"][1], 0.0) if respawn else (0.0, 0.0, 0.0)

    def timed(sample_worlds, threads, seconds):
        idx = np.arange(sample_worlds)
        groups = [(idx, respawn)] if host["respawn_worlds"] is None else \
            [(idx[host["respawn_worlds"][idx] == 1], True), (idx[host["respawn_worlds"][idx] == 0], False)]
        runners = [orc.StepBlockRunner(type_id, host["S"][sel], host["goals"][sel], host["walls"], host["P"],
                                       np.zeros((sel.size, host["S"].shape[1])), True, respawn=rs, respawn_par=rp,
                                       dtype=np.float64, threads=threads) for sel, rs in groups if sel.size]
        done, reps = 0, 0
        t0 = time.perf_counter()
        while True:
            for r in runners:
                r.run(args.dt, args.substeps)
                done += r.W * args.substeps
            reps += 1
            if time.perf_counter() - t0 >= seconds:
                break
        el = time.perf_counter() - t0
        return done * args.agents / el, reps, el

    sw_all = min(args.worlds, 32 * max(1, cores))
    v_all, reps, el = timed(sw_all, cores, args.cpu_seconds)
    v_one, reps1, el1 = timed(min(args.worlds, 64), 1, min(4.0, args.cpu_seconds))
    return {"value": v_all, "unit": "agent-substeps/s", "cores": cores, "kind": "port",
            "single_core_value": v_one,
            "sample": f"{sw_all} worlds x {args.agents} agents x {reps * args.substeps} substeps, f64 C oracle "
                      f"(oracle/), OpenMP over worlds on {cores} threads, {el:.1f} s; single-core leg: 64 worlds x "
                      f"{reps1 * args.substeps} substeps, {el1:.1f} s"}


def main():
    args = parse()
    # stdout carries ONE JSON line and nothing else: native libraries (RCCL prints a version banner from C when NCCL_DEBUG
    # asks for it) write to file descriptor 1 directly, so everything but the final line is sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd.batched import HUMAN_MODELS as SFMS

    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    _lib.set_device(local_rank)
    dist = None
    if world_size > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.force_dist and world_size == 1:   # rehearsal of the RCCL barrier path without a launcher
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world_size and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world_size}", file=sys.stderr)

    cw, host = build_worlds(args, rank)
    stream = _lib.stream_create()
    cw.stream = stream
    n_sub = args.substeps

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        cw.step(args.dt, n_sub)
    barrier()
    if args.eager:
        # one launch per step from Python, a HIP event pair around every launch
        starts = [_lib.Event() for _ in range(args.steps)]
        stops = [_lib.Event() for _ in range(args.steps)]
        t0 = time.perf_counter()
        for k in range(args.steps):
            starts[k].record(stream)
            cw.step(args.dt, n_sub)
            stops[k].record(stream)
        _lib.stream_sync(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = np.array([starts[k].elapsed_ms(stops[k]) for k in range(args.steps)])
    else:
        # the K steps are captured once into a HIP graph (untimed) and replayed with ONE launch in the timed region:
        # no per-launch host gap; HIP events on the launch stream bracket the K kernels
        with _lib.Graph.capture(stream) as graph:
            for k in range(args.steps):
                cw.step(args.dt, n_sub)
        e0, e1 = _lib.Event(), _lib.Event()
        barrier()
        t0 = time.perf_counter()
        e0.record(stream)
        graph.launch()
        e1.record(stream)
        _lib.stream_sync(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = np.array([e0.elapsed_ms(e1) / args.steps])
    from social_navigation_pyenvs_amd.sharding import max_over_ranks

    elapsed = max_over_ranks(elapsed, dist, device="cuda")

    # sanity: the state is finite and moved
    S_end = cw.get_states()
    finite = float(np.mean(np.isfinite(S_end[..., :8])))

    if rank == 0:
        agent_substeps = world_size * args.worlds * args.agents * n_sub * args.steps
        value = agent_substeps / elapsed
        family = "orca" if args.model == "orca" else ("hsfm" if args.model.startswith("hsfm") else "sfm")
        alg_bytes_launch = ALG_BYTES[family] * args.worlds * args.agents * n_sub
        if args.walls:
            alg_bytes_launch += 16 * 15 * 0  # walls are shared by all worlds: 0 B per world-substep
        k_avg = float(np.mean(kernel_ms))
        achieved = alg_bytes_launch / (k_avg * 1e-3) / 1e9
        g, b, wpb = cw.launch_geometry()
        out = {
            "metric": "env-steps/sec (worlds x agents) for HSFM 25-agent crowd",
            "value": value,
            "unit": "agent-substeps/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.worlds} worlds/GPU x {args.agents}-agent {args.model} {args.scenario} scenario"
                            f"{' + 3 polygon walls' if args.walls else ''}, {n_sub} fused substeps of "
                            f"{args.dt} s per step (one Gym step), state resident in HBM ({args.layout})",
                "worlds_per_gpu": args.worlds, "agents": args.agents, "substeps_per_step": n_sub,
                "model": args.model, "scenario": args.scenario, "parallelism": f"worlds sharded x{world_size}, no collective",
                "launch": {"grid": g, "block": b, "worlds_per_block": wpb},
            },
            "world_substeps_per_s": value / args.agents,
            "gym_steps_per_s": value / args.agents / n_sub,
            "finite_fraction": finite,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (MEASURED_TRAFFIC_B_PER_AGENT_LAUNCH[(args.model, args.scenario, args.walls)] * args.worlds * args.agents
                            if (args.model, args.scenario, args.walls) in MEASURED_TRAFFIC_B_PER_AGENT_LAUNCH else None),
                "traffic_note": "HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/)",
                "kernel": "k_orca_step" if args.model == "orca" else "k_sfm_step", "kernel_avg_ms": k_avg, "kernel_min_ms": float(np.min(kernel_ms)),
                "launch_mode": "eager, one HIP event pair per launch" if args.eager else "HIP graph of K launches, events around the graph",
                "algorithmic_bytes_per_launch": alg_bytes_launch,
                "bytes_per_agent_substep": ALG_BYTES[family],
            },
        }
        if not args.no_cpu_baseline and args.model != "orca" and world_size == 1:  # rank 0, N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, host, SFMS.index(args.model))
            out["gpu_over_cpu"] = value / world_size / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
