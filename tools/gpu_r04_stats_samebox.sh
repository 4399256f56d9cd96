#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats of the default bench for THIS round's build and for round 3's final build
# (build/ab/libfreud_sae_r3final.so, commit b0676df) on ONE box, alternating twice -- boxes differ by +-3 %, so the per-round
# kernel_stats files of two rounds cannot be compared with each other; these can
set -u
O=$PWD/gpurun_out/r04_stats_samebox; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
cd /tmp
for i in 1 2; do
  for b in r04 r03; do
    if [ $b = r03 ]; then export FREUD_SAE_LIB=$R/build/ab/libfreud_sae_r3final.so; else unset FREUD_SAE_LIB; fi
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_${b}_$i -o stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pcie-sample --steps 3000 --warmup 20 --spinup 0.2 > $O/bench_${b}_$i.json 2> $O/stats_${b}_$i.log
    python3 $R/tools/trace_timeline.py $(find $O/stats_${b}_$i -name "*kernel_trace.csv" | head -1) 3000 > $O/timeline_${b}_$i.txt 2>&1
    rm -f $(find $O/stats_${b}_$i -name "*kernel_trace.csv")
    echo "== $b build, pass $i"; head -6 $O/timeline_${b}_$i.txt
  done
done
unset FREUD_SAE_LIB
