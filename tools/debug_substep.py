"""Diagnostic (GPU box): reproduce one case of tests/test_gpu_parity.py::test_shape_specialised_builds_every_substep and print, for the
worst substep, the per-row errors of the GPU and of the float32 oracle against the float64 oracle, with the pair geometry."""
import sys, os
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests")); sys.path.insert(0, os.path.join(R_, "tests", "golden"))
from oracle import crowd_oracle as orc
from parity_util import f32, row_errors
from social_navigation_pyenvs_amd import scenarios as sc
from social_navigation_pyenvs_amd.batched import CrowdWorlds, SFMS

n, robot, W, model = int(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3]), sys.argv[4]
rng = np.random.default_rng(n + W)
for m in ["hsfm_farina", "sfm_guo", "hsfm_new_moussaid"]:
    S, goals, P, rb = sc.hybrid_worlds(W, n, m, seed0=31 + n)
    rw = (np.arange(W) % 2 == 1).astype(np.int32)
    R = A = None
    if robot:
        R = np.zeros((W, 13), np.float32)
        R[:, 0:2] = rng.uniform(-3, 3, (W, 2)); R[:, 8] = 0.3; R[:, 9] = 80; R[:, 10:12] = -R[:, 0:2]; R[:, 12] = 1.0
        A = rng.uniform(-0.8, 0.8, (W, 2)).astype(np.float32)
        S = np.concatenate([S, R[:, None, :]], axis=1)
    if m != model:
        continue
    S32, g32, P32 = f32(S), f32(goals), f32(P)
    cw = CrowdWorlds(S32, g32, P32, None, None, type=model, all_params_equal=True, respawn_bounds=rb, respawn_worlds=rw, robot_row=robot, robot=R)
    print(cw.step_variant())
    for _ in range(2):
        cw.step(0.0125, 20, A)
    Sk, gk, Rk = cw.get_states().astype(np.float64), cw.get_goals().astype(np.float64), (cw.get_robot().astype(np.float64) if robot else None)
    tr = cw.step_trace(0.0125, 20, A).astype(np.float64)
    if robot:   # the robot row as substep 1 sees it
        Sk[:, n] = Rk
        Sk[:, n, 0:2] = (Rk[:, 0:2] + A.astype(np.float64) * float(np.float32(0.0125))).astype(np.float32)
        Sk[:, n, 3:5] = A
    t = SFMS.index(model)
    for k in range(20):
        worst = (0, None)
        nxt = []
        for w in range(W):
            kw = dict(robot_visible=robot, respawn=bool(rw[w]), respawn_par=(rb[0], rb[1], 0.0) if rw[w] else (0, 0, 0),
                      robot=None if Rk is None else Sk[w, n], action=None)
            r64, g64, rob = orc.step_block(t, Sk[w], gk[w], None, P32.astype(np.float64), 0.0125, 1, np.zeros(S.shape[1]), True, **kw)
            r32, _, _ = orc.step_block(t, Sk[w], gk[w], None, P32.astype(np.float64), 0.0125, 1, np.zeros(S.shape[1]), True, dtype=np.float32, **kw)
            eg, _, _ = row_errors(tr[k, w, :n], r64[:n], Sk[w, :n, 7], 0.0125, t >= 3)
            ef, _, _ = row_errors(r32[:n], r64[:n], Sk[w, :n, 7], 0.0125, t >= 3)
            if eg.max() > 5e-6:
                i = int(np.argmax(eg))
                p = Sk[w, :, 0:2]; d = np.linalg.norm(p - p[i], axis=1); d[i] = 9
                j = int(np.argmin(d))
                print(f"substep {k+1} world {w} row {i}: GPU {eg[i]:.2e} f32-oracle {ef[i]:.2e}; nearest row {j} at {d[j]:.4f} (r sum {Sk[w,i,8]+Sk[w,j,8]:.2f}); "
                      f"cols GPU-ref {np.round((tr[k,w,i,:8]-r64[i,:8])*1e6,2)} f32-ref {np.round((r32[i,:8]-r64[i,:8])*1e6,2)}; omega_in {Sk[w,i,7]:.3f} |F dv| {np.linalg.norm(r64[i,5:7]-Sk[w,i,5:7]):.3e}")
            nxt.append((g64, rob))
        for w in range(W):
            Sk[w, :n, 0:8] = tr[k, w, :n, 0:8]; Sk[w, :n, 10:12] = tr[k, w, :n, 8:10]
            gk[w] = nxt[w][0].reshape(gk[w].shape)
            keep = ~np.isnan(gk[w][:, 0, 0]); gk[w][:, 0][keep] = tr[k, w, :n, 10:12][keep]
            if robot:
                Sk[w, n, 0:8] = tr[k, w, n, 0:8]
