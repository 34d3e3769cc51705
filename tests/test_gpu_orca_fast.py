"""GPU: the ORCA kernel's OPT-IN arithmetics ("fma": v_rcp / v_sqrt / v_rsq, determinants as mul + fma; "fast": the first three only), per substep
from re-synchronised state against the exact restatement, every agent-substep beyond north_star's 1e-5 accounted for
(tests/orca_fast_parity.py; the full-size run is tools/orca_fast_parity.py -> profiles/archive/r5c_orca_fast_parity.txt).
The bit-identity suite of the exact arithmetic is tests/test_gpu_orca.py.  ORCA's parity with rvo2 itself is UNPINNED (library absent)."""
import numpy as np
import pytest

import orca_fast_parity as ofp
import parity_util

pytestmark = pytest.mark.gpu


def test_the_arithmetic_is_a_field_of_the_context_and_exact_is_the_default():
    import os

    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    lib = _lib.load()
    S, g, margin = ofp.crossing(8, 25, 7.0, 5)
    cw = CrowdWorlds(S, g, None, margin, None, type="orca")
    if os.environ.get("CROWDSTEP_ORCA_MATH", "exact") == "exact":
        assert lib.cs_orca_default_math() == _lib.CS_ORCA_MATH_EXACT and "math=exact" in cw.step_variant(), cw.step_variant()
    # two environments of one process differ, and neither changes the other (ABI 4: cs_worlds.orca_math; there is no process-wide switch)
    others = {name: CrowdWorlds(S, g, None, margin, None, type="orca", orca_math=name) for name in ("exact", "fast", "fma")}
    for name, c in others.items():
        assert f"math={name}" in c.step_variant()
    for c in (cw, *others.values()):
        c.step(0.0125, 3)
    assert np.array_equal(cw.get_states(), others["exact"].get_states())
    assert "math=exact" in cw.step_variant() and "math=fma" in others["fma"].step_variant()
    cw.orca_math = "bogus"
    with pytest.raises(ValueError):
        cw.step(0.0125, 1)
    # an out-of-range field is refused by the library itself
    cw.orca_math = "default"
    d = cw.descriptor(); d.orca_math = 7
    import ctypes as C
    assert lib.cs_step(C.byref(d), C.c_float(0.0125), C.c_int(1), None, None) != 0
    # the generic builds (other maxNeighbors, per-agent parameters, obstacles) are always exact
    cw5 = CrowdWorlds(S, g, None, margin, None, type="orca", orca_math="fma")
    cw5.orca_params = dict(cw5.orca_params, max_neighbors=5)
    assert "FAST10=0" in cw5.step_variant() and "math=exact" in cw5.step_variant(), cw5.step_variant()


@pytest.mark.parametrize("mode,name", [(2, "fma"), (1, "fast")])
@pytest.mark.parametrize("W,n,R,nsub", [(512, 25, 7.0, 700), (256, 10, 3.0, 300), (128, 40, 6.0, 300)])
def test_fast_arithmetic_per_substep_against_the_exact_restatement(mode, name, W, n, R, nsub):
    """cfg4's crossing (and a 10- and a 40-agent one: 3 and 1 worlds per wavefront) through the dense phase and out again.  Asserted:
    the share beyond 1e-5 is small, EVERY such agent-substep is one float32 does not resolve (classes f64 / edge1 / edge4 / edge16 / op4,
    none unexplained) and the flipping decision of RVO2's programme is named for each, the build is no farther from the algorithm in double than the exact float32 restatement is, goal switches differ only on
    the switch radius, the next preferred velocity agrees wherever the velocity does."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    S, g, margin = ofp.crossing(W, n, R, 4242 + n)
    cw = CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa", orca_math=name)
    assert f"math={name}" in cw.step_variant()
    res = ofp.substeps_vs_restatement(cw, S, g, margin, 0.0125, nsub, seed=mode)
    parity_util.REPORT[f"ORCA {name} build per substep, {W}x{n}"] = {k: res[k] for k in (
        "agent_substeps", "bit_identical_agent_substeps", "beyond_bar", "beyond_bar_share", "class_f64", "class_edge1", "class_edge4", "class_edge16", "class_op4", "decisions", "probe_reproduces_build", "unexplained",
        "unexplained_share", "worst_unexplained", "disagree_build_closer_to_f64", "disagree_exact_closer_to_f64", "beyond_bar_vs_f64_build_share",
        "beyond_bar_vs_f64_exact_share", "goal_column_flips", "pref_velocity_worst", "p99", "p9999", "mean_displacement_m")}
    assert res["mean_displacement_m"] > 0.25 * R                     # the crowds really crossed
    assert res["bit_identical_agent_substeps"] > 0.5 * res["agent_substeps"]
    assert res["p99"] < 2e-6 and res["beyond_bar_share"] < 1e-3, res
    assert res["not_examined"] == 0
    # EVERY agent-substep beyond the bar is one float32 does not determine: the double evaluation is as far from the exact restatement (f64), or the
    # restatement's own answer moves that far under 1 / 4 / 16 ulps of input noise (edge*) or under <= 4 ulps of rounding in its divisions, roots and
    # product sums (op4: oracle/orca_oracle_probe.c, which also names the decision that flips -- `decisions`)
    assert res["unexplained"] == 0, res
    examined = res["beyond_bar"] - res["class_f64"]
    assert sum(res["decisions"].values()) == examined and res["decisions"].get("not sensitive", 0) == 0 and res["decisions"].get("trace diverged", 0) == 0, res["decisions"]
    assert res["probe_reproduces_build"] >= 0.97 * examined, res      # ... and a perturbed restatement lands on the build's own answer
    # as close to real arithmetic as float32 RVO2: the build's share beyond the bar from the double evaluation vs the exact restatement's
    assert res["beyond_bar_vs_f64_build"] <= 1.05 * res["beyond_bar_vs_f64_exact"] + 5, res
    # where the two float32 answers disagree the build is not the one that is usually wrong
    assert res["disagree_build_closer_to_f64"] >= 0.8 * res["disagree_exact_closer_to_f64"] - 5, res
    assert res["pref_velocity_worst"] < 1e-5


def test_fast_arithmetic_free_running_crowd_is_as_healthy_as_the_exact_one():
    """No re-synchronisation: 700 substeps of the cfg4 crossing under each arithmetic.  The rows diverge (a chaotic system), the crowd
    must not: overlaps, speeds and progress agree."""
    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    W, n = 512, 25
    S, g, margin = ofp.crossing(W, n, 7.0, 777)
    h = {}
    for mode, name in enumerate(("exact", "fast", "fma")):
        h[mode] = ofp.free_run_health(lambda: CrowdWorlds(S, g, None, margin, None, type="orca", layout="soa", orca_math=name), S, g, margin, 0.0125, 700)
    for mode in (1, 2):
        assert h[mode]["worst_overlap_m"] < h[0]["worst_overlap_m"] + 5e-3, h
        assert h[mode]["max_speed_over_vmax"] < 5e-2, h
        assert abs(h[mode]["mean_displacement_m"] - h[0]["mean_displacement_m"]) < 0.02, h
        assert abs(h[mode]["mean_goal_distance_m"] - h[0]["mean_goal_distance_m"]) < 0.02, h
        assert abs(h[mode]["overlapping_pairs_sampled"] - h[0]["overlapping_pairs_sampled"]) <= 0.1 * h[0]["overlapping_pairs_sampled"] + 20, h
