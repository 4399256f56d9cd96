#!/bin/bash
# copies what tools/profile_round.sh left under gpurun_out/prof_<tag>/ (scratch) into profiles/<tag>_* (tracked)
TAG=${1:-r02}
O=gpurun_out/prof_$TAG
for f in $O/bench_*.json; do
  b=$(basename $f .json)
  tail -1 $f > profiles/${TAG}_$b.json
done
for d in c3 c3auxk c4 c5fp8; do
  s=$(find $O/stats_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && cp $s profiles/${TAG}_kernel_stats_$d.csv
done
cp $O/timeline.txt profiles/${TAG}_timeline_timed_region.txt
for f in $O/${TAG}_*; do cp $f profiles/; done
ls profiles | grep "^${TAG}_" | wc -l
