#!/bin/bash
# Folds what tools/r03_profiles.sh left under gpurun_out/ into profiles/ (run here, after the gpurun calls).
src=gpurun_out/r03_final
for f in bench_n1 bench_n1_driver_flags bench_n1_8Mtets bench_n1_jittered; do
  [ -s $src/$f.json ] && cp $src/$f.json profiles/r03_$f.json
done
[ -s $src/unstructured_1M_tets.txt ] && grep -v "amdgpu.ids" $src/unstructured_1M_tets.txt > profiles/r03_unstructured_1M_tets.txt
if [ -s $src/stats/run_kernel_stats.csv ]; then
  cp $src/stats/run_kernel_stats.csv profiles/r03_bench_kernel_stats.csv
  (head -1 $src/stats/run_kernel_trace.csv | cut -d, -f9,10,11,12,13,17,21,22,23,24 > /dev/null
   python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r03_final/stats/run_kernel_trace.csv")))
keep = ["Kernel_Name", "Dispatch_Id", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size",
        "VGPR_Count", "SGPR_Count", "Scratch_Size"]
with open("profiles/r03_bench_kernel_trace_saa.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(keep + ["Duration_ns"])
    for r in rows:
        if "saa::" in r["Kernel_Name"]:
            w.writerow([r[k] if k != "Kernel_Name" else r[k][:90] for k in keep] + [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
PY
  )
fi
for key in r03_resident19 r03_fused38; do
  d=gpurun_out/pmc_$key
  [ -d $d ] || continue
  if [ $key = r03_resident19 ]; then
    python tools/pmc_summary.py $key $d persistent_steps_kernel --full-only --out=profiles/r03_pmc_summary.json > /dev/null
  else
    python tools/pmc_summary.py $key $d fused_step_kernel --grid=1048576 --out=profiles/r03_pmc_summary.json > /dev/null
  fi
done
[ -s $src/resident_stamps.txt ] && grep -v "amdgpu.ids" $src/resident_stamps.txt > profiles/r03_resident_stamps.txt
[ -s $src/fused_8Mtets_stamps.txt ] && grep -v "amdgpu.ids" $src/fused_8Mtets_stamps.txt > profiles/r03_fused_8Mtets_stamps.txt
[ -s $src/fused_8Mtets_ablation.txt ] && grep -v "amdgpu.ids" $src/fused_8Mtets_ablation.txt > profiles/r03_fused_8Mtets_ablation.txt
if [ -s profiles/r03_pmc_summary.json ]; then
  [ -s $src/resident_stamps.json ] && cp $src/resident_stamps.json profiles/r03_resident_stamps.json
  python tools/onchip_summary.py resident19 profiles/r03_pmc_summary.json r03_resident19 profiles/r03_resident_stamps.json --steps-per-dispatch=1000 > /dev/null
  python tools/onchip_summary.py fused38 profiles/r03_pmc_summary.json r03_fused38 --steps-per-dispatch=1 > /dev/null
fi
ls -la profiles/r03_*
