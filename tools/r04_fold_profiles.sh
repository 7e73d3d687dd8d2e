#!/bin/bash
# Folds what tools/r04_profiles.sh left under gpurun_out/ into profiles/ (run here, after the gpurun calls).
src=gpurun_out/r04
for f in bench_n1 bench_n1_driver_flags; do
  [ -s $src/$f.json ] && grep '^{' $src/$f.json | tail -1 > profiles/r04_$f.json
done
if [ -s $src/stats/run_kernel_stats.csv ]; then
  cp $src/stats/run_kernel_stats.csv profiles/r04_bench_kernel_stats.csv
  python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04/stats/run_kernel_trace.csv")))
keep = ["Kernel_Name", "Dispatch_Id", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size",
        "VGPR_Count", "SGPR_Count", "Scratch_Size"]
with open("profiles/r04_bench_kernel_trace_saa.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(keep + ["Duration_ns"])
    short = 0
    for r in rows:
        if "saa::" in r["Kernel_Name"]:
            if "persistent_steps_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 1000000:
                short += 1  # (the headline's thousands of 20-step launches: the first 300 are kept, all 1000-step launches are)
                if short > 300:
                    continue
            w.writerow([r[k] if k != "Kernel_Name" else r[k][:90] for k in keep] + [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
PY
fi
for key in r04_resident19 r04_fused38 r04_resident19_delaunay; do
  d=gpurun_out/pmc_$key
  [ -d $d ] || continue
  if [ $key = r04_fused38 ]; then
    python tools/pmc_summary.py $key $d fused_step_kernel --grid=1048576 --out=profiles/r04_pmc_summary.json --plan-from=$d/g1.log > /dev/null
  else
    python tools/pmc_summary.py $key $d persistent_steps_kernel --full-only --out=profiles/r04_pmc_summary.json --plan-from=$d/g1.log > /dev/null
  fi
done
[ -s $src/resident_stamps.txt ] && grep -v "amdgpu.ids" $src/resident_stamps.txt > profiles/r04_resident_stamps.txt
if [ -s profiles/r04_pmc_summary.json ]; then
  [ -s $src/resident_stamps.json ] && cp $src/resident_stamps.json profiles/r04_resident_stamps.json
  stamps=""; [ -s profiles/r04_resident_stamps.json ] && stamps=profiles/r04_resident_stamps.json
  python tools/onchip_summary.py resident19 profiles/r04_pmc_summary.json r04_resident19 $stamps --steps-per-dispatch=1000 --out=profiles/r04_onchip_summary.json > /dev/null
  python tools/onchip_summary.py fused38 profiles/r04_pmc_summary.json r04_fused38 --steps-per-dispatch=1 --out=profiles/r04_onchip_summary.json > /dev/null
  [ -d gpurun_out/pmc_r04_resident19_delaunay ] && python tools/onchip_summary.py resident19_delaunay profiles/r04_pmc_summary.json r04_resident19_delaunay --steps-per-dispatch=1000 --out=profiles/r04_onchip_summary.json > /dev/null
fi
ls -la profiles/r04_*
