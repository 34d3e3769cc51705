"""The policy seam end to end (SURVEY.md §8 rows b + f1), pinned on what the REFERENCE's own CADRL / SARL decided.

Golden group g16_policy (tests/golden/make_golden.py gen_g16_policy): the reference's `CADRL.predict` (crowd_nav/policy/cadrl.py:235-291) and
`SARL.predict` (multi_human_rl.py:12-88, sarl.py) run in the reference's Gym with seeded weights; per decision the crowd rows the env held, the
robot's full state, what `get_next_human_observable_states` returned to the policy, the value network's 81 outputs, the 81 action values
and the arg-max.  Only arrays travel: weights, states, values.

The value networks below are TEST-LOCAL restatements of the two published architectures (an MLP; SARL's pairwise MLP + attention over the
humans + MLP), built from the fixture's weight arrays: the policies themselves are the user's code, the framework's
promise is that its env, its look-ahead arrays and ONE batched forward of such a network reproduce the reference's decisions.
  CPU: the restated networks on the REFERENCE's own look-ahead inputs give the reference's values (the networks are the right ones).
  GPU: the repo's batched env holding the recorded crowds -> lookahead_device (cs_peek + cs_lookahead) -> one batched forward ->
       arg-max == the reference's choice."""
import numpy as np
import pytest

from golden_io import load_cases


def _weights(cases):
    out = {}
    for c in cases:
        if c["kind"] == "weights":
            keys = [str(k) for k in c["weights_keys"]]
            out[str(c["wkey"])] = {k: np.asarray(c[f"w_{i}"], dtype=np.float64) for i, k in enumerate(keys)}
    return out


def _mlp(x, w, prefix, last_relu=False):
    """crowd_nav/policy/cadrl.py mlp(): Linear / ReLU chain stored as <prefix>.<2 i>.weight / .bias"""
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in w if k.startswith(prefix + ".")})
    for j, i in enumerate(idx):
        x = x @ w[f"{prefix}.{i}.weight"].T + w[f"{prefix}.{i}.bias"]
        if j != len(idx) - 1 or last_relu:
            x = np.maximum(x, 0.0)
    return x


def value_cadrl(rot, w):
    """rot [..., N, 13] -> [...]: the value network on every (robot, human) row, minimum over the humans (cadrl.py:264-270)"""
    return _mlp(rot, w, "value_network")[..., 0].min(axis=-1)


def value_sarl(rot, w, self_dim=6):
    """rot [..., N, 13] -> [...]: sarl.py ValueNetwork.forward (with_global_state): mlp1 per human, attention scores from (mlp1, mean mlp1),
    masked softmax, weighted mlp2 features, mlp3 on (self state, weighted feature)"""
    m1 = _mlp(rot, w, "mlp1", last_relu=True)
    m2 = _mlp(m1, w, "mlp2")
    g = np.broadcast_to(m1.mean(axis=-2, keepdims=True), m1.shape)
    scores = _mlp(np.concatenate([m1, g], axis=-1), w, "attention")[..., 0]
    e = np.exp(scores) * (scores != 0)
    wts = e / e.sum(axis=-1, keepdims=True)
    feat = (wts[..., None] * m2).sum(axis=-2)
    joint = np.concatenate([rot[..., 0, :self_dim], feat], axis=-1)
    return _mlp(joint, w, "mlp3")[..., 0]


VALUE = {"cadrl": value_cadrl, "sarl": value_sarl}


def _groups():
    cases = load_cases("g16_policy")
    w = _weights(cases)
    groups = {}
    for c in cases:
        if c["kind"] == "decision":
            groups.setdefault(str(c["wkey"]), []).append(c)
    return groups, w


def test_fixture_and_restated_networks_reproduce_the_reference_values_cpu():
    """CPU (oracle side): from the reference's own peeked next states, the host restatement of compute_rotated_states_and_reward
    (oracle/crowd_oracle.py) + the restated networks give the reference's 81 action values and its choice, decision by decision."""
    from oracle import crowd_oracle as orc

    groups, w = _groups()
    assert sum(len(g) for g in groups.values()) >= 100 and set(k.split("_")[0] for k in groups) == {"cadrl", "sarl"}
    worst = 0.0
    for key, cs in groups.items():
        for c in cs:
            rot, rew = orc.lookahead(c["action_space"], c["next_humans"], c["obs"], c["robot"], float(c["dt"]))
            np.testing.assert_allclose(rew, c["rewards"], atol=1e-12)
            net = VALUE[str(c["policy"])](rot, w[key])
            # the reference evaluated its network in float32 (torch): a relative bar on outputs of magnitude up to ~150
            scale = max(1.0, float(np.max(np.abs(c["net_outputs"]))))
            worst = max(worst, float(np.max(np.abs(net - c["net_outputs"]))) / scale)
            values = rew + float(c["gamma"]) ** (float(c["dt"]) * float(c["robot"][7])) * net
            assert int(np.argmax(values)) == int(c["chosen"]), (key, c["test_case"], c["step"])
            np.testing.assert_allclose(c["action_space"][int(c["chosen"])], c["action"], atol=1e-12)
    assert worst < 5e-5, worst        # (measured 2.1e-5: four float32 layers at outputs of magnitude ~100 against this float64 evaluation)


@pytest.mark.gpu
def test_batched_env_lookahead_and_one_forward_pick_the_reference_actions():
    """GPU: every recorded decision becomes one world of a batched env (its crowd rows, goal lists and robot row as the reference's env
    held them); lookahead_device (cs_peek: one Euler step of the robot's time step with the crowd's own model; cs_lookahead: the 81
    rotated joint states and rewards) and ONE batched forward over [W, 81, N, 13] pick the reference's action in >= 99 % of the
    decisions, the rest within 1e-4 (relative) of the best value; the action values agree to float32."""
    torch = pytest.importorskip("torch")
    import configparser

    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.social_gym.social_nav_gym import BatchedSocialNavGym

    groups, w = _groups()
    total = same = 0
    worst_val = worst_next = 0.0
    for key, cs in groups.items():
        c0 = cs[0]
        W, n = len(cs), int(c0["mm_states"].shape[0])     # (the humans the reference's env really held: its test phase places ten whatever human_num says)
        scen = {"circular_crossing": "circle_crossing"}.get(str(c0["scenario"]), str(c0["scenario"]))
        cfg = configparser.RawConfigParser()
        cfg.read_dict({
            "env": {"time_limit": 50, "time_step": float(c0["substep"]), "robot_time_step": float(c0["dt"]), "val_size": 100, "test_size": 500, "randomize_attributes": "false"},
            "reward": {"success_reward": 1, "collision_penalty": -0.25, "discomfort_dist": 0.2, "discomfort_penalty_factor": 0.5},
            "sim": {"train_val_sim": scen, "test_sim": scen, "square_width": 10, "circle_radius": 7, "human_num": n, "traffic_length": 14, "traffic_height": 3},
            "humans": {"visible": "true", "policy": str(c0["model"]), "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
            "robot": {"visible": "false", "policy": "none", "radius": 0.3, "v_pref": 1, "sensor": "coordinates"},
        })
        env = BatchedSocialNavGym(cfg, W)
        env.reset(phase="test", first_case=0, device=True)
        cw = env.cw
        # the recorded crowds: state rows, goal lists (NaN-padded to the batch's slots), the robot's full state as a 13-column row
        S = np.stack([c["mm_states"][:n] for c in cs]).astype(np.float32)
        G = np.full((W, n, cw.G, 2), np.nan, np.float32)
        for k, c in enumerate(cs):
            g = np.asarray(c["mm_goals"], np.float32)[:n]
            G[k, :, :min(cw.G, g.shape[1])] = g[:, :cw.G]
            # (the reference env ran the model's default parameters: so does the batch -- BatchedSocialNavGym has no others)
        R = np.zeros((W, 13), np.float32)
        for k, c in enumerate(cs):
            r = c["robot"]                      # px, py, vx, vy, radius, gx, gy, v_pref, theta
            R[k, [0, 1, 3, 4, 8, 10, 11, 12, 2]] = r
            R[k, 9] = 80.0
        cw.set_states(S); cw.set_goals(G); cw.set_robot(R)
        acts = np.asarray(c0["action_space"], np.float64)
        rot, rew = env.lookahead_device(acts)
        rot, rew = rot.cpu().numpy().astype(np.float64), rew.cpu().numpy().astype(np.float64)
        # what the policy was handed by the reference's env: the peeked next human states sit in the rotated rows' inputs; check the peek itself
        nxt = cw.peek(float(c0["dt"]))[:, :, [0, 1, 3, 4]]
        worst_next = max(worst_next, max(float(np.max(np.abs(nxt[k] - cs[k]["next_humans"]))) for k in range(W) if not str(c0["model"]).endswith("moussaid")))
        net = VALUE[str(c0["policy"])](rot, w[key])                       # ONE batched forward: [W, 81, N, 13] -> [W, 81]
        disc = np.array([float(c["gamma"]) ** (float(c["dt"]) * float(c["robot"][7])) for c in cs])[:, None]
        values = rew + disc * net
        for k, c in enumerate(cs):
            ref = np.asarray(c["action_values"])
            scale = max(1.0, float(np.max(np.abs(ref))))
            worst_val = max(worst_val, float(np.max(np.abs(values[k] - ref))) / scale)
            pick = int(np.argmax(values[k]))
            total += 1
            if pick == int(c["chosen"]):
                same += 1
            else:                                                          # a tie at float32 resolution: the reference's best value is ours too
                assert abs(ref[pick] - ref[int(c["chosen"])]) <= 1e-4 * scale, (key, k, pick, int(c["chosen"]))
        env.close()
    assert total >= 100 and same >= 0.99 * total, (same, total)
    assert worst_next < 1e-5 and worst_val < 1e-4, (worst_next, worst_val)     # (measured: 2.5e-6 on the peeked states, 2.2e-5 relative on the action values)
    import parity_util

    parity_util.REPORT["g16 policy seam: reference CADRL / SARL decisions reproduced (GPU env + look-ahead + one batched forward)"] = {
        "decisions": total, "same_action": same, "worst_action_value_rel": worst_val, "worst_peek_abs": worst_next}
