#!/bin/bash
# round 6: the whole -m gpu suite + smoke on the current build; then the statistics exchange of the fused d = 384 path inline (inside
# finalize_losses, on the critical path) against on the communication stream under the forward (FREUD_DP_STATS=stream), 8 and 2 ranks
# sharing this GPU, three alternations each
mkdir -p gpurun_out/r06_suite
timeout 3000 python -m pytest tests -q -x -m gpu > gpurun_out/r06_suite/gpu_suite.txt 2>&1
echo "suite rc $?" >> gpurun_out/r06_suite/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_suite/smoke.txt 2>&1
echo "smoke rc $?" >> gpurun_out/r06_suite/smoke.txt
export FREUD_BENCH_SHARE_GPU=1 FREUD_BENCH_OTHER_CARRIER=0
for i in 1 2 3; do
  for mode in inline stream; do
    for n in 8 2; do
      FREUD_DP_STATS=$mode timeout 600 python bench.py --gpus $n --steps 100 --warmup 10 --no-cpu-baseline --spinup 0.3 > gpurun_out/r06_suite/dp_${mode}_n${n}_$i.json 2> gpurun_out/r06_suite/dp_${mode}_n${n}_$i.err
    done
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_suite/dp_*.json")):
    try:
        d = json.load(open(f)); t = d["dp_timing"]
        print(f.split("/")[-1], "ms %.4f plain %.4f exposed %.4f exchange %.4f stats %.4f" % (d["ms_per_step"], t["plain_ms_per_step"], t["exposed_exchange_ms"], t["exchange_ms"], t["stats_exchange_ms"]))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids gpurun_out/r06_suite/gpu_suite.txt | tail -5; tail -3 gpurun_out/r06_suite/smoke.txt
