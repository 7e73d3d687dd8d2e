#!/usr/bin/env python3
"""Folds the PMC passes (tools/pmc_collect.sh -> tools/pmc_summary.py) and the in-kernel stamps (tools/persist_stamps.py
--json=...) of one kernel and mesh into the record bench.py prints as `roofline.onchip`.

    python tools/onchip_summary.py <key> <pmc summary json> <pmc key> [<stamps json>] [--steps-per-dispatch=1000]
                                   [--out=profiles/r03_onchip_summary.json]

Definitions (counters per dispatch, MI355X_MICROARCH.md "rocprofv3 PMC slots"; SQ_* cycle counters are quad-cycles
summed over waves, SQ_BUSY_CYCLES is cycles summed over the 32 shader engines, SQ_LDS_* are cycles summed over CUs):
  launch_cycles            = SQ_BUSY_CYCLES / 32
  valu_busy                = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * launch_cycles)     share of time a SIMD issues VALU
  lds_busy                 = SQ_LDS_IDX_ACTIVE / (256 CUs * launch_cycles)              share of time a CU's LDS is busy
  lds_bank_conflict_share  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wave_wait_share          = SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked at s_waitcnt / s_barrier
  wave_issue_stall_share   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   waves stalled at issue
  fp64_flop_per_step       = 64 lanes * (ADD_F64 + MUL_F64 + TRANS_F64 + 2 * FMA_F64) / steps per dispatch
                             (executed flops, the copies of elements on block borders included; idle lanes counted)
  barrier_wait_share       = cycles a wave spends between reaching a barrier and leaving it / cycles per step
                             (medians over waves, in-kernel stamps of a diagnostic build)
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    key, pmc_path, pmc_key = pos[0], pos[1], pos[2]
    stamps_path = pos[3] if len(pos) > 3 else None
    steps, out_path = 1000, os.path.join(REPO, "profiles", "r04_onchip_summary.json")
    for a in sys.argv[1:]:
        if a.startswith("--steps-per-dispatch="):
            steps = int(a.split("=", 1)[1])
        if a.startswith("--out="):
            out_path = os.path.join(REPO, a.split("=", 1)[1])
    with open(pmc_path) as fh:
        pmc = json.load(fh)

    def g(c):
        return pmc[f"{pmc_key}:{c}"]["mean_per_dispatch"]

    cyc = g("SQ_BUSY_CYCLES") / 32.0
    rec = {"launch_cycles": cyc, "steps_per_dispatch": steps,
           "valu_busy": 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * cyc),
           "lds_busy": g("SQ_LDS_IDX_ACTIVE") / (256.0 * cyc),
           "lds_bank_conflict_share": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
           "wave_wait_share": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
           "wave_issue_stall_share": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
           "valu_insts_per_step": g("SQ_INSTS_VALU") / steps,
           "source": f"{os.path.relpath(pmc_path, REPO)} [{pmc_key}]"}
    try:
        rec["fp64_flop_per_step"] = 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") +
                                            g("SQ_INSTS_VALU_TRANS_F64") + 2.0 * g("SQ_INSTS_VALU_FMA_F64")) / steps
        rec["fp64_insts_per_step"] = {c: g(f"SQ_INSTS_VALU_{c}_F64") / steps for c in ("ADD", "MUL", "FMA", "TRANS")}
    except KeyError:
        pass
    if stamps_path:
        with open(stamps_path) as fh:
            st = json.load(fh)
        rec["barrier_wait_share"] = st["barrier_cycles"] / st["cycles_per_step_median_total"]
        rec["cycles_per_step_stamped"] = st["cycles_per_step_median_total"]
        rec["phase_cycles"] = st["phases"]
        rec["source"] += f" + {os.path.relpath(stamps_path, REPO)}"
    try:
        with open(out_path) as fh:
            summary = json.load(fh)
    except OSError:
        summary = {}
    summary[key] = rec
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    print(key, json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
