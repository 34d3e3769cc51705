#!/usr/bin/env python3
"""What cs_consume_staged_worlds costs against the number of worlds that take over a staged episode in the launch (0, 1, 8, 44, 256):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cp -- python3 tools/consume_probe.py run ; python3 tools/consume_probe.py show gpurun_out/cp"""
import csv, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COUNTS = [0, 1, 8, 44, 256]
REPS = 12
if sys.argv[1] == "run":
    import ctypes as C
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import configparser
    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym
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
    dl = env._device_loop_state()
    c = env._step_pieces(dl, 0, "same_step")
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    with torch.cuda.stream(env.device_stream()):
        for rep in range(REPS):
            for k in COUNTS:
                dl["mask"].zero_()
                if k:
                    idx = torch.randperm(4096, device="cuda", generator=g)[:k]
                    dl["mask"][idx] = 1
                    dl["seeds"][idx] += dl["stride"]          # what the bookkeeping of the step does for a finished world
                torch.cuda.synchronize()
                _lib.check(lib.cs_consume_staged_worlds(*c["tail"]))
                torch.cuda.synchronize()
    assert env.failed_resets() == 0
    sys.exit(0)
f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(f)) if "k_consume_staged" in r["Kernel_Name"]]
d = np.array([x for _, x in sorted(d)]).reshape(REPS, len(COUNTS))[2:]
for j, k in enumerate(COUNTS):
    print(f"{k:4d} worlds take over their staged episode: k_consume_staged mean {d[:, j].mean():6.2f}  min {d[:, j].min():6.2f}  max {d[:, j].max():6.2f} us")
