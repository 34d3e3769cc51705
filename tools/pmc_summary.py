#!/usr/bin/env python3
"""tools/pmc_summary.py DIR OUT.json -- per configuration of tools/profile_round6.sh: mean per dispatch of the rocprofv3 PMC
counters of the step kernel -> HBM-side bytes per agent and launch, VALU issue figures.  The result is committed as
profiles/pmc_summary.json (bench.py reads it back and tags every figure with its source file)."""
import csv
import glob
import json
import os
import sys

SHAPES = {"cfg3": (4096, 25, "hsfm_farina_25_hybrid"), "cfg2": (4096, 10, "sfm_helbing_10_circle"), "cfg4": (4096, 25, "orca_25_circle"),
          "cfg4_first20": (4096, 25, "orca_25_circle_first20"), "cfg4_dense": (4096, 25, "orca_25_circle_dense"),
          "cfg4_first20_fma": (4096, 25, "orca_25_circle_first20_fma"), "cfg4_dense_fma": (4096, 25, "orca_25_circle_dense_fma"),
          "n15": (4096, 15, "hsfm_farina_15_hybrid"), "n40": (4096, 40, "hsfm_farina_40_hybrid"),
          "cfg5": (8192, 50, "hsfm_farina_50_circle_walls_static"), "cfg3x4": (16384, 25, "hsfm_farina_25_hybrid_16384"),
          "moussaid": (4096, 25, "hsfm_new_moussaid_25_hybrid"), "cfg3_new_guo": (4096, 25, "hsfm_new_guo_25_hybrid"),
          "robot26": (4096, 25, "hsfm_farina_25_hybrid_robot"), "n30": (4096, 30, "hsfm_farina_30_hybrid"),
          "peragent": (4096, 25, "hsfm_farina_25_hybrid_peragent"), "cfg5_nowalls": (8192, 50, "hsfm_farina_50_circle_static"),
          # the windows of the DRIVER's protocol (--steps 20 --warmup 5): the headline over Gym steps 5-25, cfg5 over steps 20-40.  A kernel's
          # instruction count follows the crowd's state (cfg5: how many polygons the wave vote skips), so valu_frac pairs counters and
          # time of the SAME window: bench.py looks "<key>@w<warmup>s<steps>" up before "<key>"
          "cfg3_w5s20": (4096, 25, "hsfm_farina_25_hybrid@w5s20"), "cfg5_w20s20": (8192, 50, "hsfm_farina_50_circle_walls_static@w20s20")}
# (round 5: every other_configs row runs over its own window whatever --steps says, so only the headline needs its driver-protocol window)
SUBSTEPS = 20
SIMDS = 256 * 4


def means(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "k_sfm_step" not in k and "k_orca_step" not in k:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {c: sum(v) / len(v) for c, v in acc.items()}, (max(len(v) for v in acc.values()) if acc else 0)


def main(d, out, tag):
    res = {}
    for name, (W, n, key) in SHAPES.items():
        files = sorted(glob.glob(os.path.join(d, f"{name}_pmc_*.csv")))
        if not files:
            continue
        m, nd = {}, 0
        for f in files:
            mm, k = means(f)
            m.update(mm)
            nd = max(nd, k)
        e = {"dispatches_averaged": nd, "worlds": W, "source": f"profiles/{tag}_{name}_pmc_*.csv (rocprofv3 --pmc, separate passes)"}
        try:   # the library build the counters were collected on (bench.py: roofline.pmc_build_matches)
            e["build_id"] = json.load(open(os.path.join(d, f"{name.split('_w')[0] if '_w' in name and name.split('_w')[-1][0].isdigit() else name}_bench.json")))["build_id"]
        except Exception:
            e["build_id"] = None
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            # FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE reads 1/2 of the bytes of 16-B-per-lane streaming loads and is
            # uncalibrated for the 4-B-per-lane loads of this kernel (MI355X_MICROARCH.md, HBM): reported as measured, not corrected.
            e["fetch_KiB_per_launch"], e["write_KiB_per_launch"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
            e["hbm_bytes_per_agent_launch"] = (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0 / (W * n)
            e["traffic_source"] = f"profiles/{tag}_{name}_pmc_FETCH_SIZE.csv + _pmc_WRITE_SIZE.csv"
        if "SQ_WAVES" in m and m["SQ_WAVES"] > 0:
            waves = m["SQ_WAVES"]
            v = {"waves_per_launch": waves,
                 "valu_insts_per_wave_substep": m.get("SQ_INSTS_VALU", 0) / waves / SUBSTEPS,
                 "lds_insts_per_wave_substep": m.get("SQ_INSTS_LDS", 0) / waves / SUBSTEPS,
                 "wave_quadcycles_per_substep": m.get("SQ_WAVE_CYCLES", 0) / waves / SUBSTEPS,
                 "valu_active_frac_of_wave_cycles": m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
                 "wait_any_frac": m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
                 "wait_inst_any_frac": m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
                 "waves_per_simd": waves / SIMDS,
                 "source": f"profiles/{tag}_{name}_pmc_SQ_WAVES.csv"}
            if "SQ_INSTS_VALU_FMA_F32" in m:   # the instruction classes tools/valu_issue_ceiling.hip priced (bench.py VALU_CLASS_CYCLES)
                v["mix_per_wave_substep"] = {k: m.get(c, 0) / waves / SUBSTEPS for k, c in (
                    ("add_f32", "SQ_INSTS_VALU_ADD_F32"), ("mul_f32", "SQ_INSTS_VALU_MUL_F32"), ("fma_f32", "SQ_INSTS_VALU_FMA_F32"),
                    ("trans_f32", "SQ_INSTS_VALU_TRANS_F32"), ("int32", "SQ_INSTS_VALU_INT32"), ("cvt", "SQ_INSTS_VALU_CVT"),
                    ("add_f64", "SQ_INSTS_VALU_ADD_F64"), ("mul_f64", "SQ_INSTS_VALU_MUL_F64"))}
                v["mix_source"] = f"profiles/{tag}_{name}_pmc_SQ_INSTS_VALU_ADD_F32.csv"
            # VALU issue occupancy of a SIMD ~ (share of a wave's cycles its VALU instructions are active) x (waves resident per SIMD)
            v["simd_valu_busy_est"] = min(1.0, v["valu_active_frac_of_wave_cycles"] * max(1.0, v["waves_per_simd"]))
            v["bound"] = "VALU issue / single-wave latency (not HBM): see DESIGN.md §4.1"
            e["valu"] = v
        res[key] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    d = sys.argv[1]
    main(d, sys.argv[2], os.path.basename(os.path.normpath(d)))
