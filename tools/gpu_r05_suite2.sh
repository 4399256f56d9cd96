#!/bin/bash
O=gpurun_out/r05_suite2; mkdir -p $O
timeout 3600 python -m pytest tests -m gpu -q > $O/gputest.txt 2>&1; echo "rc=$?" >> $O/gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
tail -5 $O/gputest.txt; tail -2 $O/smoke.txt; cut -c1-300 $O/bench_driver_style.json
