"""GPU tests at BASELINE.json's full sizes, through size-independent properties: batch composition must not
change any world (bitwise "checksum of checksums"), speeds respect the clamp, sampled worlds equal the oracle."""
import numpy as np
import pytest

from oracle import crowd_oracle as orc
from parity_util import fused_substeps_vs_oracle, record, row_errors, single_call_bar

pytestmark = pytest.mark.gpu


def _worlds(cfg):
    from social_navigation_pyenvs_amd import scenarios as sc

    if cfg == "cfg2":      # 4096 worlds x 10-agent SFM circle crossing
        W, n, model = 4096, 10, "sfm_helbing"
        pos, yaw, g = sc.circular_crossing(W, n, 7.0, 1000)
        S, goals, P, rb, rw, walls = sc.make_states(pos, yaw, g), g, np.tile(sc.default_params(model), (n, 1)), None, None, None
    elif cfg in ("cfg3", "cfg3x4"):    # 4096 (16384: four waves per SIMD) worlds x 25-agent HSFM hybrid scenario
        W, n, model = (4096 if cfg == "cfg3" else 16384), 25, "hsfm_farina"
        S, goals, P, rb = sc.hybrid_worlds(W, n, model)
        rw, walls = (np.arange(W) % 2 == 1).astype(np.int32), None
    else:                  # one GPU's shard of cfg5: 8192 worlds x 50-agent HSFM + static obstacles (both flavours)
        # THE worlds bench.py's cfg5 entry steps (generators.static_obstacle_crossing: one function for both, first 8192 of the 65536)
        from social_navigation_pyenvs_amd import generators as gen

        W, n, model = 8192, 50, "hsfm_farina"  # (hsfm_new* blows up |omega| to 1e108 in the f64 reference itself here)
        # "cfg5shard" = the worlds rank 0 owns; "cfg5shard@3" / "@7" = the worlds ranks 3 and 7 of the 8-GPU run own (global ids 24576.. and 57344..:
        # worlds are functions of (seed, GLOBAL id), sharding.world_shard) -- no 8-GPU node has been available to any round, so the other ranks'
        # worlds are stepped under the parity check here
        first = 8192 * int(cfg.split("@")[1]) if "@" in cfg else 0
        cw0 = gen.static_obstacle_crossing(W, n, model, first_world=first, radius=14.0, n_static=3, walls=True)
        S, goals, P, walls = cw0.get_states(), cw0.get_goals(), np.tile(sc.default_params(model), (n, 1)), sc.polygon_walls()
        rb, rw = None, None
    return W, n, model, S.astype(np.float32), goals.astype(np.float32), P.astype(np.float32), rb, rw, walls


# the kernel build behind each published number (crowdstep.hip select_variant); asserted so that a dispatch change cannot
# silently un-test a build
VARIANT = {"cfg2": "k_sfm_step_row16<SOC=0,HEADED=0,ROWS=10>", "cfg3": "MAXT=64,OCC=1,ROWS_CT=25,LEAN=1",
           "cfg3x4": "MAXT=64,OCC=4,ROWS_CT=25,LEAN=1", "cfg5shard": "MAXT=64,OCC=3,ROWS_CT=50,LEAN=2"}
VARIANT["cfg5shard@3"] = VARIANT["cfg5shard@7"] = VARIANT["cfg5shard"]


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg3x4", "cfg5shard"])
def test_full_size_properties(cfg):
    from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

    W, n, model, S, goals, P, rb, rw, walls = _worlds(cfg)
    peq = True
    t = SFMS.index(model)
    w64 = None if walls is None else walls.astype(np.float32).astype(np.float64)

    def make(sel, layout="soa"):
        return CrowdWorlds(S[sel], goals[sel], P, None, walls, type=model, all_params_equal=peq, respawn_bounds=rb,
                           respawn_worlds=None if rw is None else rw[sel], layout=layout)

    def run(sel, nsub=20, steps=3):
        cw = make(sel)
        for _ in range(steps):
            cw.step(0.0125, nsub)
        return cw.get_states(), cw.get_goals()

    def oracle_block(w, S0, g0, nsub):
        respawn = rw is not None and bool(rw[w])
        rp = (rb[0], rb[1], 0.0) if respawn else (0.0, 0.0, 0.0)
        return orc.step_block(t, S0[w].astype(np.float64), g0[w].astype(np.float64), w64, P.astype(np.float64), 0.0125, nsub,
                              np.zeros(n), peq, respawn=respawn, respawn_par=rp)

    rng = np.random.default_rng(0)
    # (0) THE BUILD THE PUBLISHED NUMBER COMES FROM, through cs_step (mode == commit goals in place), against the f64 oracle on
    #     sampled worlds: one substep and one Gym step (20 fused substeps) from the initial state
    cw = make(np.arange(W))
    assert VARIANT[cfg] in cw.step_variant(), cw.step_variant()
    sample = rng.choice(W, 12, replace=False)
    for nsub, tol in ((1, 1e-5), (20, 1e-5)):      # (round 6: the 20-substep END state at 1e-5 too -- measured worst 3.6e-6; round 5 allowed 5e-5)
        cw = make(np.arange(W))
        cw.step(0.0125, nsub)
        out, gout = cw.get_states(), cw.get_goals()
        for w in sample:
            ref, ref_goals, _ = oracle_block(w, S, goals, nsub)
            err = np.max(np.abs(out[w][:, [0, 1, 3, 4]] - ref[:, [0, 1, 3, 4]]))
            assert err < tol, (cfg, "cs_step", nsub, int(w), err)
            record(f"full size {cfg} {nsub} substep(s) through cs_step (GPU vs f64 oracle)", err)
            assert np.max(np.abs(np.nan_to_num(gout[w]) - np.nan_to_num(ref_goals))) < 1e-4
    full, gfull = run(np.arange(W))
    assert np.all(np.isfinite(full[..., :8]))
    # (1) composition invariance, bitwise: a permuted half-batch gives the same rows for the same worlds
    sel = rng.permutation(W)[: W // 2 + 3]
    part, gpart = run(sel)
    np.testing.assert_array_equal(part, full[sel])
    np.testing.assert_array_equal(gpart, gfull[sel])
    # (2) the speed clamp holds for every agent (body speed for HSFM == |v| after R(theta))
    sp = np.linalg.norm(full[..., 3:5], axis=-1)
    assert np.all(sp <= S[..., 12] * (1 + 1e-5) + 1e-6)
    # (3) things that must not change: radius, mass, desired speed
    np.testing.assert_array_equal(full[..., [8, 9, 12]], S[..., [8, 9, 12]])
    # (4) sampled worlds against the f64 oracle from the EVOLVED state (three Gym steps in): one more Gym step through cs_step
    #     (the same lean build), and one substep through the array seam (the generic build)
    cw = CrowdWorlds(full, gfull, P, None, walls, type=model, all_params_equal=peq, respawn_bounds=rb, respawn_worlds=rw, layout="soa")
    assert VARIANT[cfg] in cw.step_variant()
    cw.step(0.0125, 20)
    out20 = cw.get_states()
    cw = CrowdWorlds(full, gfull, P, None, walls, type=model, all_params_equal=peq, layout="aos")
    out = cw.get_states(cw.update_humans_parallel(0.0125, in_place=False))
    for w in rng.choice(W, 12, replace=False):
        ref, _, _ = orc.update_humans(t, full[w].astype(np.float64), gfull[w].astype(np.float64), w64,
                                      P.astype(np.float64), 0.0125, np.zeros(n), peq, False)
        err = np.max(np.abs(out[w][:, [0, 1, 3, 4]] - ref[:, [0, 1, 3, 4]]))
        assert err < 1e-5, (cfg, int(w), err)
        ref20, _, _ = oracle_block(w, full, gfull, 20)
        err = np.max(np.abs(out20[w][:, [0, 1, 3, 4]] - ref20[:, [0, 1, 3, 4]]))
        respawned = rw is not None and bool(rw[w])
        assert err < 1e-5, (cfg, "evolved + 20", int(w), err, respawned)      # (round 6: 1e-5 for respawning worlds too -- measured worst 4.4e-6; round 5: 5e-5 / 3e-4)
        record(f"full size {cfg} 20 substeps from the evolved state (GPU vs f64 oracle)", err)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg3x4", "cfg5shard", "cfg5shard@3", "cfg5shard@7"])
def test_full_size_every_substep_of_the_fused_launch(cfg):
    """north_star's 1e-5 per SUBSTEP inside the fused 20-substep launch at BASELINE.json's full sizes, for the very kernel
    builds the published numbers come from: cs_step_trace records every row after every fused substep of ALL worlds; 48 sampled
    worlds are checked substep by substep against the oracle's single substep (respawn rule included) restarted from the GPU's
    own previous rows -- from the initial state and from the evolved one (three Gym steps in: contacts, goal switches, respawns)."""
    from parity_util import fused_substeps_vs_oracle
    from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

    W, n, model, S, goals, P, rb, rw, walls = _worlds(cfg)
    t = SFMS.index(model)
    rng = np.random.default_rng(11)

    def make(S_, g_):
        return CrowdWorlds(S_, g_, P, None, walls, type=model, all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw, layout="soa")

    cw = make(S, goals)
    assert VARIANT[cfg] in cw.step_variant(), cw.step_variant()
    S_k, g_k = S, goals
    for phase in ("initial state", "evolved state"):
        sample = rng.choice(W, 48, replace=False)
        res = fused_substeps_vs_oracle(cw, t, S_k, g_k, P, None, walls, 0.0125, 20, True, respawn=rw, respawn_bounds=rb, worlds=sample,
                                       group=f"full size {cfg} per substep inside the fused launch, {phase}", what=f"{cfg} {phase}")
        assert res["within"] >= res["substeps"] - res["ill_conditioned"], (cfg, phase, res)
        if phase == "initial state" and "@" in cfg:
            # another rank's shard really holds other worlds than rank 0's
            W0, _, _, S0, _, _, _, _, _ = _worlds("cfg5shard")
            assert not np.array_equal(S0[:64], S[:64])
        if phase == "initial state":
            # the traced launch is cs_step's launch: same rows as an untraced batch, bit for bit
            ref = make(S, goals)
            ref.step(0.0125, 20)
            np.testing.assert_array_equal(cw.get_states(), ref.get_states())
            for _ in range(3):
                cw.step(0.0125, 20)
            S_k, g_k = cw.get_states(), cw.get_goals()


def test_fused_block_equals_repeated_single_substeps_bitwise():
    """n fused substeps == n launches of one substep (same arithmetic, state round-trips through HBM)."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n, model, S, goals, P, rb, rw, walls = _worlds("cfg3")
    sel = np.arange(512)
    a = CrowdWorlds(S[sel], goals[sel], P, None, None, type=model, all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw[sel])
    b = CrowdWorlds(S[sel], goals[sel], P, None, None, type=model, all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw[sel])
    a.step(0.0125, 20)
    for _ in range(20):
        b.step(0.0125, 1)
    np.testing.assert_array_equal(a.get_states(), b.get_states())
    np.testing.assert_array_equal(a.get_goals(), b.get_goals())


@pytest.mark.parametrize("n,model,robot", [(80, "hsfm_farina", False), (150, "sfm_guo", True), (64, "sfm_helbing", True), (300, "hsfm_new_guo", False)])
def test_large_worlds_one_block_per_world(n, model, robot):
    """rows > 64: one world per block of up to 1024 threads (multi-wave barriers, block-wide contact / respawn votes)."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

    W = 5
    rng = np.random.default_rng(n)
    pos, yaw, g = sc.circular_crossing(W, n + int(robot), 0.28 * n, 7)
    S = sc.make_states(pos, yaw, g).astype(np.float32)
    S[..., 3:5] = rng.normal(0, 0.4, S[..., 3:5].shape)
    S[..., 5:7] = rng.normal(0, 0.4, S[..., 5:7].shape)
    S[:, ::7, 0:2] = S[:, 1::7, 0:2][:, : S[:, ::7].shape[1]] + 0.45   # some overlapping pairs -> contact pass
    goals = g[:, :n].astype(np.float32)
    P = np.tile(sc.default_params(model), (n, 1)).astype(np.float32)
    t = SFMS.index(model)
    cw = CrowdWorlds(S, goals, P, None, None, type=model, all_params_equal=True, robot_row=robot)
    grid, block, wpb = cw.launch_geometry()
    assert wpb == 1 and block >= n + int(robot) and block % 64 == 0
    out = cw.get_states(cw.update_humans_parallel(0.0125, in_place=False))
    for w in range(W):
        args = (t, S[w].astype(np.float64), goals[w].astype(np.float64), None, P.astype(np.float64), 0.0125, np.zeros(n + int(robot)), True, robot)
        ref, _, _ = orc.update_humans(*args)
        ref32, _, _ = orc.update_humans(*args, dtype=np.float32)
        # 1e-5, or 3 x the float32 oracle's own error where the overlapping pairs (k1 = 120 kN/m) put float32 itself beyond it
        tol, e32 = single_call_bar(ref32[:n], ref[:n], S[w, :n, 7], 0.0125, t >= 3)
        err = float(row_errors(out[w][:n], ref[:n], S[w, :n, 7], 0.0125, t >= 3)[0].max())
        assert err < tol, (n, model, w, err, e32)
        record(f"one world per block (MAXT = 1024), one substep through the array seam (GPU vs f64 oracle)", err)
    # every substep of a fused launch of the one-world-per-block build (cs_step: multi-wave barriers, block-wide contact votes)
    cwt = CrowdWorlds(S, goals, P, None, None, type=model, all_params_equal=True, robot_row=robot)
    assert "MAXT=1024" in cwt.step_variant(), cwt.step_variant()
    res = fused_substeps_vs_oracle(cwt, t, S, goals, P, None, None, 0.0125, 5, True, robot_row=robot,
                                   group="one world per block (MAXT = 1024) per substep inside the fused launch", what=f"MAXT=1024 {model} n={n}")
    assert res["within"] >= res["substeps"] - res["ill_conditioned"], (n, model, res)
    # fused block on the big variant == repeated single substeps, bitwise
    a = CrowdWorlds(S, goals, P, None, None, type=model, all_params_equal=True, robot_row=robot)
    b = CrowdWorlds(S, goals, P, None, None, type=model, all_params_equal=True, robot_row=robot)
    a.step(0.0125, 5)
    for _ in range(5):
        b.step(0.0125, 1)
    np.testing.assert_array_equal(a.get_states(), b.get_states())


@pytest.mark.parametrize("model", ["hsfm_farina", "sfm_guo", "hsfm_new_moussaid"])
def test_long_rollout_soak(model):
    """250 Gym steps (5000 substeps, 62 s of simulated time) of 1024 hybrid worlds: the crowd crosses, touches (contact
    pass), switches goals and respawns many times.  Everything stays finite and inside the speed clamp, a sub-batch run on
    its own ends bit-identical (whatever the wavefront packing), and the evolved state still steps like the f64 oracle."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

    W, n = 1024, 25
    S, goals, P, rb = sc.hybrid_worlds(W, n, model)
    rw = (np.arange(W) % 2 == 1).astype(np.int32)
    S, goals, P = S.astype(np.float32), goals.astype(np.float32), P.astype(np.float32)
    sel = np.r_[3:40, 900:921]                      # 58 worlds: odd count, both kinds
    full = CrowdWorlds(S, goals, P, None, None, type=model, all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw, layout="soa")
    part = CrowdWorlds(S[sel], goals[sel], P, None, None, type=model, all_params_equal=True, respawn_bounds=rb,
                       respawn_worlds=rw[sel], layout="aos")
    g_before = goals.copy()
    for _ in range(250):
        full.step(0.0125, 20)
        part.step(0.0125, 20)
    A, B = full.get_states(), part.get_states()
    assert np.all(np.isfinite(A[..., :8]))
    np.testing.assert_array_equal(A[sel], B)
    np.testing.assert_array_equal(full.get_goals()[sel], part.get_goals())
    sp = np.linalg.norm(A[..., 3:5], axis=-1)
    assert np.all(sp <= S[..., 12] * (1 + 1e-5) + 1e-6)
    assert np.any(full.get_goals()[::2, :, 0] != g_before[::2, :, 0])       # circle worlds: goals were switched
    assert np.any(A[1::2, :, 0] > 7.0 - 1e-3)                                # traffic worlds: somebody respawned at the bound
    # one more substep from the evolved state against the f64 oracle (sampled worlds, no respawn in this call)
    cw = CrowdWorlds(A, full.get_goals(), P, None, None, type=model, all_params_equal=True, layout="aos")
    out = cw.get_states(cw.update_humans_parallel(0.0125, in_place=False))
    gA = full.get_goals()
    rng = np.random.default_rng(5)
    t = SFMS.index(model)
    flips = 0
    for w in rng.choice(W, 16, replace=False):
        args = (t, A[w].astype(np.float64), gA[w].astype(np.float64), None, P.astype(np.float64), 0.0125, np.zeros(n), True, False)
        ref, _, _ = orc.update_humans(*args)
        with np.errstate(over="ignore", invalid="ignore"):
            ref32, _, _ = orc.update_humans(*args, dtype=np.float32)
        tol, e32 = single_call_bar(ref32, ref, A[w][:, 7], 0.0125, t >= 3)     # 1e-5, or 3 x the float32 oracle's own error on this world
        e_rows = row_errors(out[w], ref, A[w][:, 7], 0.0125, t >= 3)[0]
        if t % 3 == 2 and np.any(e_rows >= tol):                                # Moussaid: rows of a pair on sign(theta ~ 0) are coin flips
            from parity_util import moussaid_sign_ambiguous
            assert moussaid_sign_ambiguous(A[w].astype(np.float64), P[0].astype(np.float64), n).any(), (model, int(w), float(e_rows.max()))
            flips += int((e_rows >= tol).sum())
            e_rows = np.where(e_rows >= tol, 0.0, e_rows)
        assert e_rows.max() < tol, (model, int(w), float(e_rows.max()), e32)
        record(f"soak: one substep from the state 250 Gym steps in ({'Moussaid' if t % 3 == 2 else 'Helbing / Guo'})", float(e_rows.max()))
    assert flips <= 4, flips


def _big_world(rows, n, model, rng, spacing=0.8):
    from social_navigation_pyenvs_amd import scenarios as sc

    side = int(np.ceil(np.sqrt(rows)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    pos = (np.stack([gx.ravel(), gy.ravel()], -1)[:rows] - side / 2) * spacing + rng.uniform(-0.12, 0.12, (rows, 2))
    S = np.zeros((1, rows, 13))
    S[0, :, 0:2] = pos
    S[0, :, 2] = rng.uniform(-np.pi, np.pi, rows)
    S[0, :, 5:7] = rng.normal(0, 0.4, (rows, 2))
    c, s = np.cos(S[0, :, 2]), np.sin(S[0, :, 2])
    S[0, :, 3] = c * S[0, :, 5] - s * S[0, :, 6]; S[0, :, 4] = s * S[0, :, 5] + c * S[0, :, 6]
    if not model.startswith("hsfm"):
        S[0, :, 3:5] = rng.normal(0, 0.4, (rows, 2)); S[0, :, 5:8] = 0
    else:
        S[0, :, 7] = rng.normal(0, 0.3, rows)
    S[0, :, 8] = rng.uniform(0.25, 0.35, rows); S[0, :, 9] = rng.uniform(60, 90, rows); S[0, :, 12] = rng.uniform(0.8, 1.3, rows)
    goals = np.full((1, n, 3, 2), np.nan)
    goals[0, :, 0] = -pos[:n] + rng.normal(0, 0.5, (n, 2)); goals[0, :, 1] = pos[:n]
    goals[0, ::17, 0] = pos[:n][::17] + 0.1                       # some humans stand on their goal: the list rotates
    S[0, :n, 10:12] = goals[0, :, 0]
    return S.astype(np.float32), goals.astype(np.float32), np.tile(sc.default_params(model), (n, 1)).astype(np.float32)


@pytest.mark.parametrize("model,peq,robot,walls", [("sfm_helbing", True, False, False), ("hsfm_farina", True, True, True),
                                                   ("hsfm_new_guo", False, False, True), ("sfm_moussaid", True, False, False)])
def test_world_of_4096_rows_through_the_grid(model, peq, robot, walls):
    """a world far beyond one block (4096 rows; the LDS kernel ends at 1024).  Rows stay in HBM, partners come
    from the 3 x 3 cells of a uniform grid whose edge is the force's reach; one substep through the array seam (in place and out of
    place, with the reference's in-place side effects) and a fused block of substeps, against the f64 oracle on all 4096 rows."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

    rows = 4096
    n = rows - int(robot)
    rng = np.random.default_rng(rows + len(model))
    S, goals, P = _big_world(rows, n, model, rng)
    if not peq:
        P = (P * rng.uniform(0.9, 1.1, P.shape)).astype(np.float32)
    W = sc.polygon_walls().astype(np.float32) if walls else None
    t = SFMS.index(model)
    cw = CrowdWorlds(S, goals, P, None, W, type=model, all_params_equal=peq, robot_row=robot)
    assert "k_bw_sfm_step" in cw.step_variant("update"), cw.step_variant("update")
    out = cw.get_states(cw.update_humans_parallel(0.0125, in_place=False))[0]
    ref, s_after, g_after = orc.update_humans(t, S[0].astype(np.float64), goals[0].astype(np.float64), None if W is None else W.astype(np.float64),
                                              P.astype(np.float64), 0.0125, np.zeros(rows), peq, robot)
    # a dense lattice with ~200 body contacts: 1e-5, or 3 x the float32 oracle's own error on this very world where float32 itself is
    # beyond it (hsfm_new_guo: 2.5e-5); Moussaid rows on sign(theta ~ 0) are identified as such, not tolerated
    with np.errstate(over="ignore", invalid="ignore"):
        ref32, _, _ = orc.update_humans(t, S[0].astype(np.float64), goals[0].astype(np.float64), None if W is None else W.astype(np.float64),
                                        P.astype(np.float64), 0.0125, np.zeros(rows), peq, robot, dtype=np.float32)
    headed = t >= 3
    tol, e32 = single_call_bar(ref32[:n], ref[:n], S[0, :n, 7], 0.0125, headed)
    e_rows = row_errors(out[:n], ref[:n], S[0, :n, 7], 0.0125, headed)[0]
    if model.endswith("moussaid") and np.any(e_rows >= tol):
        from parity_util import moussaid_sign_ambiguous
        amb = moussaid_sign_ambiguous(S[0].astype(np.float64), P[0].astype(np.float64), n)
        assert amb.any(), "rows beyond the bar in a world without any sign(theta ~ 0) pair"   # (the partner's reaction -f flips with it)
        assert int((e_rows >= tol).sum()) <= max(2, 0.002 * n)
        e_rows = np.where(e_rows >= tol, 0.0, e_rows)
    err = float(e_rows.max())
    assert err < tol, (model, err, e32)
    record(f"4096-row world through the grid, {model} (GPU vs f64 oracle)", err)
    np.testing.assert_array_equal(cw.get_goals()[0], g_after.astype(np.float32))                  # rotated goal lists, exact
    s_in = cw.get_states()[0]
    np.testing.assert_array_equal(s_in[:n, 10:12], s_after[:n, 10:12].astype(np.float32))        # goal columns written back
    if model.startswith("hsfm"):
        assert np.max(np.abs(s_in[:n, 3:5] - s_after[:n, 3:5])) < 1e-5                            # refreshed linear velocity
    if robot:
        np.testing.assert_array_equal(out[n], S[0, n])
    # a fused block (double-buffered in HBM, the grid rebuilt every substep) == repeated single substeps, bitwise
    a = CrowdWorlds(S, goals, P, None, W, type=model, all_params_equal=peq, robot_row=robot)
    b = CrowdWorlds(S, goals, P, None, W, type=model, all_params_equal=peq, robot_row=robot)
    a.step(0.0125, 3)
    for _ in range(3):
        b.step(0.0125, 1)
    np.testing.assert_array_equal(a.get_states(), b.get_states())
    # every substep of the fused block on the grid path (cs_step_trace records from HBM after every substep) at the 1e-5 / F32_SLACK rule
    c = CrowdWorlds(S, goals, P, None, W, type=model, all_params_equal=peq, robot_row=robot)
    res = fused_substeps_vs_oracle(c, t, S, goals, P, None, W, 0.0125, 3, peq, robot_row=robot,
                                   group=f"grid path per substep inside the fused block ({'Moussaid' if model.endswith('moussaid') else 'Helbing / Guo'})",
                                   what=f"4096-row world {model}")
    assert res["within"] >= res["substeps"] - res["ill_conditioned"], (model, res)
    np.testing.assert_array_equal(c.get_states(), a.get_states())          # the traced launch IS cs_step's


def test_grid_path_on_small_worlds_equals_the_lds_kernel():
    """The grid path forced onto small worlds (CROWDSTEP_BIGWORLD_MIN_ROWS=1): several worlds per launch, SoA layout, per-world
    parameters -- within float32 rounding of the LDS kernel's result and of the oracle."""
    import os

    from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

    rng = np.random.default_rng(3)
    W, n = 7, 90
    parts = [_big_world(n, n, "hsfm_guo", rng, spacing=0.9) for _ in range(W)]
    S = np.concatenate([p[0] for p in parts]); goals = np.concatenate([p[1] for p in parts]); P = parts[0][2]
    res = {}
    for mode in ("grid", "lds"):
        if mode == "grid":
            os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
        try:
            cw = CrowdWorlds(S, goals, P, None, None, type="hsfm_guo", all_params_equal=True, layout="soa")
            assert ("k_bw_sfm_step" in cw.step_variant()) == (mode == "grid")
            cw.step(0.0125, 4)
            res[mode] = cw.get_states()
        finally:
            os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    # both paths substep by substep against the oracle at 1e-5 (grid path: CROWDSTEP_BIGWORLD_MIN_ROWS=1 while its launches are made)
    for mode in ("grid", "lds"):
        if mode == "grid":
            os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
        try:
            cw = CrowdWorlds(S, goals, P, None, None, type="hsfm_guo", all_params_equal=True, layout="soa")
            r = fused_substeps_vs_oracle(cw, SFMS.index("hsfm_guo"), S, goals, P, None, None, 0.0125, 4, True,
                                         group=f"grid path forced onto 90-row worlds vs the LDS kernel, per substep ({mode})", what=f"90-row worlds, {mode} path")
            assert r["within"] >= r["substeps"] - r["ill_conditioned"], (mode, r)
            np.testing.assert_array_equal(cw.get_states(), res[mode])
        finally:
            os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    # secondary: the two paths sum their partners in different orders -- float32 rounding of four stiff substeps apart
    assert np.max(np.abs(res["grid"][..., [0, 1, 3, 4]] - res["lds"][..., [0, 1, 3, 4]])) < 2e-5


def test_grid_path_respawn_robot_through_d_robot_and_peek_equal_the_lds_kernel():
    """SURVEY.md §8 row f3 remainder on the grid path (forced onto small worlds, CROWDSTEP_BIGWORLD_MIN_ROWS=1): the parallel-traffic
    respawn rule, a visible robot handed over AND moved through cs_worlds.d_robot by an action, and cs_peek -- against the LDS kernel
    on the same hybrid batch: rows within float32 rounding, goal lists and robot rows exact, peek commits nothing."""
    import os

    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 9, 25
    S, goals, P, rb = sc.hybrid_worlds(W, n, "hsfm_farina", seed0=3)
    rw = (np.arange(W) % 2 == 1).astype(np.int32)
    rng = np.random.default_rng(1)
    for w in range(1, W, 2):                               # traffic worlds: three humans at the edge of the respawn zone
        for k, i in enumerate(rng.choice(n, 3, replace=False)):
            S[w, i, 0] = goals[w, i, 0, 0] + 3.0 + 0.004 * (k + 1)
            S[w, i, 3] = -0.9; S[w, i, 5] = 0.9
    R = np.zeros((W, 13), np.float32)
    R[:, 0:2] = rng.uniform(-3, 3, (W, 2)); R[:, 8] = 0.3; R[:, 9] = 80; R[:, 10:12] = -R[:, 0:2]; R[:, 12] = 1.0
    A = rng.uniform(-0.8, 0.8, (W, 2)).astype(np.float32)
    St = np.concatenate([S, R[:, None, :]], axis=1).astype(np.float32)
    res = {}
    for mode in ("grid", "lds"):
        if mode == "grid":
            os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
        try:
            cw = CrowdWorlds(St, goals, P, None, None, type="hsfm_farina", all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw,
                             robot_row=True, robot=R, layout="soa" if mode == "grid" else "aos")
            assert ("k_bw_sfm_step" in cw.step_variant()) == (mode == "grid")
            before = cw.get_states().copy()
            pk = cw.peek(0.25)
            np.testing.assert_array_equal(cw.get_states(), before)              # nothing committed
            np.testing.assert_array_equal(cw.get_goals(), goals.astype(np.float32))
            cw.step(0.0125, 20, A)
            res[mode] = (cw.get_states(), cw.get_goals(), cw.get_robot(), pk)
        finally:
            os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    g, l = res["grid"], res["lds"]
    # the grid path itself, every substep of the fused launch against the oracle (respawn rule, robot moved by its action before every
    # substep and handed over as the last row): north_star's bar, not a comparison of two float32 kernels
    os.environ["CROWDSTEP_BIGWORLD_MIN_ROWS"] = "1"
    try:
        from social_navigation_pyenvs_amd.batched import SFMS
        cw = CrowdWorlds(St, goals, P, None, None, type="hsfm_farina", all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw,
                         robot_row=True, robot=R, layout="soa")
        r = fused_substeps_vs_oracle(cw, SFMS.index("hsfm_farina"), St, goals, P, None, None, 0.0125, 20, True, respawn=rw, respawn_bounds=rb,
                                     robot_row=True, robot=R, action=A, group="grid path per substep: respawn rule + robot through d_robot",
                                     what="grid path, hybrid batch with robot")
        assert r["within"] >= r["substeps"] - r["ill_conditioned"], r
        np.testing.assert_array_equal(cw.get_states(), g[0])
    finally:
        os.environ.pop("CROWDSTEP_BIGWORLD_MIN_ROWS", None)
    # secondary (20-substep end state of two float32 kernels): respawned humans land at contact distance (25 kN/m)
    assert np.max(np.abs(g[0][..., [0, 1, 3, 4]] - l[0][..., [0, 1, 3, 4]])) < 3e-4
    moved = np.abs(l[0][:, :n, 0] - St[:, :n, 0]) > 1.0
    assert moved.any() and np.array_equal(moved, np.abs(g[0][:, :n, 0] - St[:, :n, 0]) > 1.0)
    assert np.max(np.abs(np.nan_to_num(g[1]) - np.nan_to_num(l[1]))) < 1e-4
    np.testing.assert_array_equal(g[2], l[2])                                             # the robot rows moved by the action
    np.testing.assert_allclose(g[2][:, 0:2], R[:, 0:2] + 20 * 0.0125 * A, atol=5e-6)
    np.testing.assert_array_equal(g[0][:, n, 0:5], g[2][:, 0:5])                          # ... and are the last state row
    assert np.max(np.abs(g[3] - l[3])) < 1e-4                                             # peek: one Euler step of 0.25 s


def test_world_of_4096_rows_with_the_respawn_rule_matches_the_oracle():
    """A parallel-traffic world far beyond one block (4096 humans in a 74 m x 96 m field, everybody walking to x = -38) with the
    respawn rule: the humans inside the 3 m zone of their goal -- several in the same substep -- are respawned behind the rightmost
    one in index order (k_bw_respawn); five substeps against the f64 oracle, the respawned rows and the goal lists included."""
    from social_navigation_pyenvs_amd import scenarios as sc
    from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

    n, model = 4096, "hsfm_farina"
    rng = np.random.default_rng(77)
    cols, lanes = 32, 128
    gx_, gy_ = np.meshgrid(np.arange(cols), np.arange(lanes))
    pos = np.stack([gx_.ravel() * 2.3 - 35.0, (gy_.ravel() - lanes / 2) * 0.75], -1) + rng.uniform(-0.05, 0.05, (n, 2))
    S = np.zeros((1, n, 13))
    S[0, :, 0:2] = pos
    S[0, :, 2] = np.pi; S[0, :, 3] = -0.9; S[0, :, 5] = 0.9
    S[0, :, 8] = 0.3; S[0, :, 9] = 75; S[0, :, 12] = 1.0
    goals = np.zeros((1, n, 1, 2))
    goals[0, :, 0, 0] = -38.0; goals[0, :, 0, 1] = pos[:, 1]
    near = rng.choice(np.flatnonzero(gx_.ravel() == 0), 40, replace=False)   # forty humans of the first column on the edge of the zone
    S[0, near, 0] = -35.0 + rng.uniform(0.0, 0.05, 40)
    S[0, :, 10:12] = goals[0, :, 0]
    P = np.tile(sc.default_params(model), (n, 1))
    bounds = (40.0, 50.0)
    S32, g32, P32 = S.astype(np.float32), goals.astype(np.float32), P.astype(np.float32)
    cw = CrowdWorlds(S32, g32, P32, None, None, type=model, all_params_equal=True, respawn_bounds=bounds)
    assert "k_bw_sfm_step" in cw.step_variant(), cw.step_variant()
    # every one of the five substeps (the respawn substeps included) at the 1e-5 / F32_SLACK rule, through cs_step_trace on the grid path
    r = fused_substeps_vs_oracle(cw, SFMS.index(model), S32, g32, P32, None, None, 0.0125, 5, True, respawn=True, respawn_bounds=bounds,
                                 group="grid path per substep: 4096-human traffic world with the respawn rule", what="4096-human traffic world")
    assert r["within"] >= r["substeps"] - r["ill_conditioned"], r
    got, gg = cw.get_states()[0], cw.get_goals()[0]
    ref, rg, _ = orc.step_block(SFMS.index(model), S32[0].astype(np.float64), g32[0].astype(np.float64), None, P32.astype(np.float64), 0.0125, 5,
                                np.zeros(n), True, respawn=True, respawn_par=(bounds[0], bounds[1], 0.0))
    moved = np.abs(ref[:, 0] - S32[0, :, 0]) > 30.0
    assert moved.sum() >= 20 and np.array_equal(moved, np.abs(got[:, 0] - S32[0, :, 0]) > 30.0)
    # the c-th respawned human stands at x_0 + c * 2 max_r, accumulated in float32 as the rule is sequential: c ulps of 88 m at most
    assert np.max(np.abs(got[moved][:, 0] - ref[moved][:, 0])) < 40 * 8e-6
    # secondary: the five-substep END state (rows 0.75 m apart: contacts amplify float32 rounding from substep to substep)
    assert np.max(np.abs(got[:, [1, 3, 4]] - ref[:, [1, 3, 4]])) < 5e-5
    assert np.max(np.abs(got[~moved][:, 0] - ref[~moved][:, 0])) < 5e-5
    assert np.max(np.abs(gg - rg)) < 1e-4
    assert np.all(got[moved][:, 0] >= bounds[0] - 1e-4)           # behind everybody else, never in front of the bound
