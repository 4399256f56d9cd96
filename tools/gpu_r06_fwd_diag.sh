#!/bin/bash
# round 6, second GPU call: (1) the trajectory and self-launch tests again; (2) the forward's hand-over stamps (-DFF2_WAITSTAMP build) on the
# headline batch and on N(0,1); (3) same-box A/B of the packed latent arithmetic (-DFF2_PACKED=1) on DENSE latents -- round 5 measured
# it only on the over-fitted, 2 %-dense batch; (4) LDS counters of the forward; (5) the clock the chip holds inside the shipped kernels on the
# three batches (GRBM_GUI_ACTIVE / 8 / duration per dispatch): cycles equal, clock differs = the data sensitivity IS the power limit.
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r06_diag
mkdir -p "$OUT"; export TMPDIR=/tmp
export FREUD_SAE_ALLOW_OLD_LIB=1
B="python3 $ROOT/bench.py --no-cpu-baseline --no-pcie-sample"
timeout 1500 python -m pytest tests/test_trajectory_gpu.py -q -s -m gpu > "$OUT/trajectory.txt" 2>&1
echo "trajectory rc $?" >> "$OUT/trajectory.txt"
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "without_a_launcher or several_ranks" > "$OUT/selflaunch.txt" 2>&1
echo "selflaunch rc $?" >> "$OUT/selflaunch.txt"
for data in lowrank normal; do
  FREUD_SAE_LIB=build/ab/libfreud_sae_waitstamp.so FREUD_FF2_WAITSTAMP=1 $B --dbg 65 --steps 100 --warmup 20 --data $data > "$OUT/waitstamp_$data.json" 2> "$OUT/waitstamp_$data.txt"
  $B --dbg 65 --steps 100 --warmup 20 --data $data > "$OUT/stamp_$data.json" 2> "$OUT/stamp_$data.txt"
done
for i in 1 2 3; do
  for lib in "" build/ab/libfreud_sae_ff2packed.so; do
    for args in "--data normal" "--rotate 4" ""; do
      echo -n "${lib:-current} [$args] " >> "$OUT/ab_packed.txt"
      FREUD_SAE_LIB=$lib $B --steps 200 --warmup 20 --breakdown $args 2>&1 | grep -E "per-kernel|ms_per_step" | tr '\n' ' ' | sed -e 's/.*"fwd_fused_gemm": \([0-9.]*\).*"bwd_fused_gemm": \([0-9.]*\).*"reduce_grads": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/fwd \1 bwd \2 reduce \3 step \4/' >> "$OUT/ab_packed.txt"
      echo >> "$OUT/ab_packed.txt"
    done
  done
done
cd /tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" "$OUT/counters_list.txt" | sort -u > "$OUT/lds_counters_available.txt"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL \
  -d "$OUT/pmc_lds" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-pcie-sample --steps 5 --warmup 2 --spinup 0 > /dev/null 2> "$OUT/pmc_lds.log"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM \
  -d "$OUT/pmc_lds2" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-pcie-sample --steps 5 --warmup 2 --spinup 0 > /dev/null 2> "$OUT/pmc_lds2.log"
for data in lowrank normal; do
  for rot in 1 4; do
    timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
      -d "$OUT/pmc_clock_${data}_rot$rot" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-pcie-sample --data $data --rotate $rot --steps 200 --warmup 10 --spinup 0.5 \
      > "$OUT/bench_clock_${data}_rot$rot.json" 2> "$OUT/pmc_clock_${data}_rot$rot.log"
  done
done
cd $ROOT
python3 tools/parse_clock.py "$OUT" > "$OUT/clock_summary.txt" 2>&1
python3 - > "$OUT/lds_summary.txt" 2>&1 <<PY
import csv, glob, collections
for tag in ("pmc_lds", "pmc_lds2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print(tag, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "fused" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
for d in "$OUT"/pmc_*; do [ -d "$d" ] && rm -rf "$d"; done
cat "$OUT/clock_summary.txt" "$OUT/lds_summary.txt"; cat "$OUT/ab_packed.txt"; cat "$OUT"/waitstamp_*.txt | grep -v amdgpu.ids
