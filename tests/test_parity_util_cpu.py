"""CPU check of the per-substep parity harness itself (tests/parity_util.fused_substeps_vs_oracle): a stand-in batch whose
`step_trace` is produced by the float32 instantiation of the oracle, substep by substep, must pass the harness on the golden
blocks (goal switches, respawns, walls, per-agent parameters) -- and a stand-in with a deliberate 3e-5 error in one substep
must fail it.  The GPU tests hand the harness cs_step_trace's records instead."""
import numpy as np
import pytest

from golden_io import load_cases
from oracle import crowd_oracle as orc
from parity_util import f32, fused_substeps_vs_oracle


class OracleTraced:
    """Looks like CrowdWorlds.step_trace for ONE world; the rows come from orc.step_block(dtype=float32)."""

    def __init__(self, c, poke=None):
        self.c, self.poke = c, poke

    def step_trace(self, dt, nsub):
        c = self.c
        S = f32(c["in_states"]).astype(np.float64)
        goals = f32(c["in_goals"]).astype(np.float64)
        up = lambda key: None if c.get(key) is None else f32(c[key]).astype(np.float64)
        rp = (c["respawn_bounds"] + [0.0]) if c["respawn"] else (0.0, 0.0, 0.0)
        n = S.shape[0]
        trace = np.zeros((nsub, 1, n, 12), np.float32)
        for k in range(nsub):
            with np.errstate(over="ignore", invalid="ignore"):
                S, goals, _ = orc.step_block(c["type"], S, goals, up("in_obstacles"), up("in_params"), dt, 1, up("in_safety"),
                                             c["all_params_equal"], respawn=c["respawn"], respawn_par=rp, dtype=np.float32)
            S = S.astype(np.float64); goals = goals.astype(np.float64)
            if self.poke is not None and k == self.poke:
                S[0, 3] += 3e-5
            trace[k, 0, :, 0:8] = S[:, 0:8]
            trace[k, 0, :, 8:10] = S[:, 10:12]
            trace[k, 0, :, 10:12] = goals[:, 0]
        return trace


def _check(c, poke=None):
    return fused_substeps_vs_oracle(OracleTraced(c, poke), c["type"], c["in_states"], c["in_goals"], c["in_params"], c["in_safety"],
                                    c.get("in_obstacles"), c["dt"], c["n_substeps"], c["all_params_equal"], respawn=c["respawn"],
                                    respawn_bounds=c["respawn_bounds"] if c["respawn"] else None, what="harness self-check")


def test_harness_accepts_the_float32_oracle_on_every_golden_block():
    total = within = 0
    for group in ("g2_block", "g13_block_sizes"):
        for c in load_cases(group):
            res = _check(c)
            total += res["substeps"]; within += res["within"]
            assert res["substeps"] == c["n_substeps"]
    assert total > 1000 and within > 0.95 * total, (total, within)


def test_harness_rejects_one_bad_substep():
    c = [c for c in load_cases("g2_block") if c["type"] == 0 and not c["respawn"]][0]
    assert _check(c)["within"] == c["n_substeps"]
    with pytest.raises(AssertionError, match="substep 8 of the fused launch"):
        _check(c, poke=7)
