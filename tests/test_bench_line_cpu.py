"""bench.py's stdout contract on CPU: the ONE line stays under 4 KB and is strict JSON whatever the measurements were, and
`--gpus N` without a launcher starts N ranks of its own (stub worker: no GPU, no torch)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned_full(n_ranks=1):
    """A full result assembled from committed measurements: profiles/r3r_main_bench.json is the 22.6 KB line of round 3 -- the one
    the driver could not parse -- with the blocks this round added beside it."""
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", "r3r_main_bench.json")))
    full["n_gpus"] = n_ranks
    full["dist"] = {"backend": "nccl" if n_ranks > 1 else None, "world_size": n_ranks, "launcher": "self", "rccl_version": "2.26.6"}
    full["ranks"] = [{"rank": r, "device": r, "pci": f"0000:{0x05 + 0x10 * r:02x}:00.0", "worlds": 4096, "kernel_us": 31.134706640243532 + r} for r in range(n_ranks)]
    full["gym_step"] = {"worlds": 4096, "humans": 25, "unit": "us per batched Gym step", "steps": 150, "no_reset": 44.41234567, "same_step": 61.2345678,
                       "next_step": 52.3456789, "same_step_failed_resets": 0, "next_step_failed_resets": 0}
    full["full_json"] = "gpurun_out/bench_full.json"
    # round 5: cfg5's per-GPU shard and a Moussaid row, a CPU figure beside the rows the oracle covers, CPU model + nproc in the block
    for name in ("cfg5_shard", "moussaid"):
        full["other_configs"].append(dict(full["other_configs"][0], name=name))
    for o in full["other_configs"]:
        if o["name"] in bench.CPU_ROWS or o["name"] == "cfg4_dense":
            o["cpu_value"] = 12345678.912345
    if full.get("cpu_baseline"):
        full["cpu_baseline"].update(cpu_model="AMD EPYC 9575F 64-Core Processor", nproc=256)
    return full


@pytest.mark.parametrize("n_ranks", [1, 8])
def test_line_is_compact_strict_json_with_everything_the_contract_names(n_ranks):
    full = canned_full(n_ranks)
    assert len(json.dumps(full)) > 20000          # the input really is the oversized one
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) < 4096, len(line)
    d = json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["config"]["workload"].startswith("4096 worlds/GPU x 25-agent hsfm_farina")
    rl = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert rl["bound"] == "hbm" and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-4
    assert abs(rl["frac"] - full["roofline"]["frac"]) < 1e-4 and rl["valu_frac"] is not None
    if n_ranks == 1:
        cb = d["cpu_baseline"]
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] == "port"
    assert abs(d["value"] / full["value"] - 1) < 1e-4 and abs(d["ms_per_step"] / full["ms_per_step"] - 1) < 1e-4
    assert len(d["ranks"]) == n_ranks and d["ranks"][0]["pci"] and d["dist"]["world_size"] == n_ranks
    oc = d["other_configs"]
    assert oc["columns"][:5] == ["name", "ms_per_step", "kernel_us", "frac", "valu_frac"]
    assert [r[0] for r in oc["rows"]] == [o["name"] for o in full["other_configs"]] and len(oc["rows"]) == 10
    assert oc["columns"][-1] == "cpu_value" and any(r[-1] for r in oc["rows"])
    if n_ranks == 1:
        assert d["cpu_baseline"]["cpu_model"] and d["cpu_baseline"]["nproc"]
    assert d["gym_step"]["no_reset"] and d["full_json"]


def test_line_survives_missing_pmc_entries_and_nan():
    full = canned_full()
    full["roofline"].update(traffic=None, valu=None, valu_frac=None, pmc_build_matches=None)
    full["other_configs"][0]["valu_frac"] = float("nan")      # never reaches the line as a bare NaN
    full["gym_step"] = {"error": "x" * 160}
    d = json.loads(bench.compact_line(full))
    assert d["roofline"]["traffic"] is None and d["other_configs"]["rows"][0][4] is None


def test_oversized_optional_blocks_are_dropped_not_printed():
    full = canned_full(8)
    full["other_configs"] = full["other_configs"] * 6          # 48 rows: would not fit
    line = bench.compact_line(full)
    assert len(line.encode()) < 4096
    d = json.loads(line)
    assert "dropped" in d["other_configs"] and d["roofline"]["frac"] and d["value"]


def test_short_variant():
    assert bench.short_variant("k_sfm_step<SOC=0,HEADED=1,PEQ=1,MAXT=64,OCC=1,ROWS_CT=25,LEAN=1> grid=2048 block=64 wpb=2") == "sfm<0,1,1,64,1,25,1>g2048"
    assert bench.short_variant("k_sfm_step_row16<SOC=0,HEADED=0,ROWS=10> grid=1024 block=64 wpb=4") == "sfm_row16<0,0,10>g1024"
    assert bench.short_variant("k_orca_step<FAST10=1,MAXT=64> grid=2048 block=64 wpb=2 math=fma") == "orca<1,64>g2048:fma"
    assert bench.short_variant(None) is None


STUB = textwrap.dedent("""
    import json, os, sys
    # a rank of the launcher: must find the torchrun-style environment, a distinct rank, and the arguments passed through
    rank, ws = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    open(os.path.join(os.environ["STUB_DIR"], f"rank{rank}"), "w").write(" ".join(sys.argv[1:]))
    print(f"noise from rank {rank}")                 # only rank 0's LAST stdout line is relayed
    if rank == 0:
        print(json.dumps({"n_gpus": ws, "argv": sys.argv[1:]}))
    sys.exit(int(os.environ.get("STUB_FAIL_RANK", "-1")) == rank and 7 or 0)
""")


def run_launcher(tmp_path, n, visible, extra_env=None, extra_args=()):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; a = ['--gpus', '{n}', '--steps', '3'] + {list(extra_args)!r}; "
            f"rc = bench.launch_ranks(bench.parse(a), a, worker={str(stub)!r}, n_visible={visible}); "
            "assert 'torch' not in sys.modules, 'the launcher must not import torch'; sys.exit(rc)")
    env = dict(os.environ, STUB_DIR=str(tmp_path), **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)


def test_launcher_starts_n_ranks_and_relays_rank0(tmp_path):
    r = run_launcher(tmp_path, 4, visible=8)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # ONE line on stdout: rank 0's last
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["argv"] == ["--gpus", "4", "--steps", "3"]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("rank")) == ["rank0", "rank1", "rank2", "rank3"]
    assert "noise from rank 1" in r.stderr                  # the other ranks' stdout goes to stderr


def test_launcher_refuses_when_fewer_gpus_are_visible(tmp_path):
    r = run_launcher(tmp_path, 8, visible=1)
    assert r.returncode == 3 and r.stdout.strip() == "" and "needs 8 visible" in r.stderr
    assert not [f for f in os.listdir(tmp_path) if f.startswith("rank")]
    # the one-GPU rehearsal: every rank on GPU 0
    r = run_launcher(tmp_path, 2, visible=1, extra_args=("--same-device", "--dist-backend", "gloo"))
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 2


def test_launcher_reports_a_failed_rank(tmp_path):
    r = run_launcher(tmp_path, 2, visible=2, extra_env={"STUB_FAIL_RANK": "1"})
    assert r.returncode == 7 and "exit codes [0, 7]" in r.stderr


def test_rank_refuses_a_world_size_it_was_not_asked_for():
    """`python bench.py --gpus 8` under a launcher that made ONE rank must not print a line that says n_gpus = 1."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == "" and "WORLD_SIZE=1" in r.stderr


def test_launcher_ends_the_other_ranks_when_one_dies(tmp_path):
    """A rank that exits non-zero while the others wait (here: sleep) must not leave the launcher hanging in a rendezvous."""
    import time

    stub = tmp_path / "stub_hang.py"
    stub.write_text("import os, sys, time\nr = int(os.environ['RANK'])\nif r == 1:\n    sys.exit(9)\ntime.sleep(120)\n")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; a = ['--gpus', '3']; "
            f"sys.exit(bench.launch_ranks(bench.parse(a), a, worker={str(stub)!r}, n_visible=3))")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=60)
    assert time.time() - t0 < 30 and r.returncode == 9 and r.stdout.strip() == ""


def test_launcher_with_eight_ranks(tmp_path):
    """The shape of the driver's 8-GPU run, with stub ranks: eight processes, eight distinct ranks, one port, ONE relayed line."""
    r = run_launcher(tmp_path, 8, visible=8, extra_args=("--total-worlds", "65536"))
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 8
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("rank")) == [f"rank{k}" for k in range(8)]
    assert all(open(tmp_path / f"rank{k}").read() == "--gpus 8 --steps 3 --total-worlds 65536" for k in range(8))


def test_line_is_relayed_at_once_and_a_rank_hanging_in_its_teardown_is_ended(tmp_path):
    """Rank 0 prints its line and exits 0; rank 1 hangs (a stuck final barrier).  The line must reach stdout without waiting for rank 1,
    and the launcher must end rank 1 within its deadline instead of hanging."""
    import time

    stub = tmp_path / "stub_teardown.py"
    stub.write_text("import json, os, sys, time\nr = int(os.environ['RANK'])\nif r == 0:\n    print(json.dumps({'n_gpus': 2}))\n    sys.exit(0)\ntime.sleep(300)\n")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.RANK_EXIT_TIMEOUT_S = 3.0; a = ['--gpus', '2']; "
            f"sys.exit(bench.launch_ranks(bench.parse(a), a, worker={str(stub)!r}, n_visible=2))")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=60)
    assert time.time() - t0 < 30
    assert json.loads(r.stdout.strip())["n_gpus"] == 2          # the measurement survived
    assert r.returncode == 9 and "did not exit within" in r.stderr


def test_cpu_baseline_runner_covers_the_robot_and_per_agent_rows():
    """bench.py times the oracle through StepBlockRunner (in place, copy-free): with a visible robot row and with per-agent parameters
    it must step exactly what orc.step_block steps (the function the parity tests use)."""
    import numpy as np

    import bench
    from oracle import crowd_oracle as orc
    from social_navigation_pyenvs_amd.batched import HUMAN_MODELS

    args = bench.parse([])
    for name in ("robot26", "peragent"):
        spec = next(s for s in bench.other_config_specs(args) if s["name"] == name)
        host = bench.host_worlds(dict(spec, worlds=6, total_worlds=None, device_generator=False), 0, 1)
        sel = np.nonzero(host["respawn_worlds"] == 0)[0] if host["respawn_worlds"] is not None else np.arange(6)
        t = HUMAN_MODELS.index(spec["model"])
        P = host["P"] if np.asarray(host["P"]).ndim == 2 else np.asarray(host["P"])[sel]
        peq = bool(host["all_params_equal"])
        rob = None if host["robot"] is None else host["robot"][sel]
        act = None if host["action"] is None else host["action"][sel]
        r = orc.StepBlockRunner(t, host["S"][sel], host["goals"][sel], host["walls"], P, np.zeros((sel.size, host["S"].shape[1])), peq,
                                threads=1, robot=rob, action=act)
        r.run(spec["dt"], 7)
        S, g, _ = orc.step_block(t, host["S"][sel], host["goals"][sel], host["walls"], P, spec["dt"], 7, np.zeros((sel.size, host["S"].shape[1])), peq,
                                 robot_visible=rob is not None, robot=rob, action=act, threads=1)
        np.testing.assert_array_equal(r.S, S)
        assert not np.array_equal(r.S, host["S"][sel])
