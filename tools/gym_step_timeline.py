#!/usr/bin/env python3
"""The device-resident Gym step as the GPU sees it.
   run:      rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python3 tools/gym_step_timeline.py run [same_step|next_step|none] [gaps]
   analyse:  python3 tools/gym_step_timeline.py show gpurun_out/tl
   -> per kernel mean duration, and the idle time of the step stream between consecutive launches (step -> consume -> next step)"""
import csv, glob, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import torch
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym
    mode = {"same_step": True, "next_step": "next_step", "none": False}[sys.argv[2] if len(sys.argv) > 2 else "same_step"]
    import configparser
    cfg = configparser.RawConfigParser()
    cfg.read_dict({
        "env": {"time_limit": 50, "time_step": 0.0125, "robot_time_step": 0.25, "val_size": 100, "test_size": 500, "randomize_attributes": "false"},
        "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
        "sim": {"train_val_sim": "hybrid_scenario", "test_sim": "hybrid_scenario", "square_width": 10, "circle_radius": 7, "human_num": 25, "traffic_length": 14, "traffic_height": 3},
        "humans": {"visible": "true", "policy": "hsfm_farina", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
        "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
    })
    env = BatchedSocialNavGym(cfg, 4096)
    env.reset(phase="train", first_case=0, device=True)
    buf = env.action_buffer()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    buf.copy_(torch.randn(4096, 2, device="cuda", generator=g) * 0.5)
    with torch.cuda.stream(env.device_stream()):
        gaps = len(sys.argv) > 3 and sys.argv[3] == "gaps"      # a host synchronisation after every step: every kernel starts on an idle GPU
        for _ in range(360):
            env.step_device(buf, auto_reset=mode)
            if gaps:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    sys.exit(0)

f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    k = r["Kernel_Name"]
    tag = next((t for t in ("k_consume_staged", "k_refill_staged", "k_sfm_step", "k_generate") if t in k), None)
    if tag:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag))
ev.sort()
main = [e for e in ev if e[2] in ("k_sfm_step", "k_consume_staged")]
main = main[len(main) // 6:]          # warm-up off
dur, gap = {}, {}
for i, (s, e, t) in enumerate(main):
    dur.setdefault(t, []).append((e - s) / 1e3)
    if i + 1 < len(main):
        gap.setdefault(t + " -> " + main[i + 1][2], []).append((main[i + 1][0] - e) / 1e3)
for t, v in dur.items():
    v = np.array(v); print(f"{t:20s} calls {len(v):4d}  mean {v.mean():6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
for t, v in gap.items():
    v = np.array(v); print(f"idle {t:36s} n {len(v):4d}  mean {v.mean():6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
steps = [e for e in main if e[2] == "k_sfm_step"]
per = np.diff([s for s, _, _ in steps]) / 1e3
print(f"step period (start to start): mean {per.mean():.2f}  median {np.median(per):.2f} us over {len(per)} steps")
ref = [e for e in ev if e[2] == "k_refill_staged"]
print(f"refill passes: {len(ref)}, mean {np.mean([(e - s) / 1e3 for s, e, _ in ref]) if ref else 0:.1f} us")
