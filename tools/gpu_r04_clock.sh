#!/bin/bash
# Round 4, VERDICT item 2: the clock the chip holds inside the SHIPPED fused kernels, from counters instead of stamp builds.
#   effective clock = GRBM_GUI_ACTIVE / 8 / (End - Start) per dispatch (MI355X_MICROARCH.md "DVFS give-back"); the quotient reads
#   high on dispatches shorter than ~0.3 ms, so the same kernels are also run at 8 x the rows (2-2.5 ms dispatches).
# Interleaved on one box: fwd v1 / fwd v2 (FREUD_FWD) alternate, three rounds.  Run ON the GPU box from the repo root:
#   gpurun -- bash tools/gpu_r04_clock.sh
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/r04_clock
mkdir -p "$OUT"; export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-pcie-sample"
for i in 1 2; do
  $B --steps 20 --warmup 5 > "$OUT/bench_driver_style_$i.json" 2>> "$OUT/bench.err"
done
cd /tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
for round in 1 2 3; do
  for fv in 1 2; do
    export FREUD_FWD=$fv
    for rows in 65536 524288; do
      steps=200; [ $rows -gt 65536 ] && steps=40
      timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS \
        -d "$OUT/pmc_f${fv}_r${rows}_$round" -o pmc --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-pcie-sample --rows $rows --steps $steps --warmup 10 --spinup 0.1 \
        > "$OUT/bench_f${fv}_r${rows}_$round.json" 2> "$OUT/pmc_f${fv}_r${rows}_$round.log"
    done
  done
done
unset FREUD_FWD
cd $ROOT
python3 tools/parse_clock.py "$OUT" > "$OUT/clock_summary.txt" 2>&1
cat "$OUT/clock_summary.txt"
# keep one raw CSV as a sample, drop the rest (tens of MB of per-dispatch rows)
for d in "$OUT"/pmc_*; do case "$d" in *pmc_f2_r65536_1) ;; *) [ -d "$d" ] && rm -rf "$d";; esac; done
