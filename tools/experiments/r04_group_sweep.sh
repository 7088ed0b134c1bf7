# Member groups again, with group streams that are known to sit on different hardware queues (csrc/stream_apart.hpp):
# ms per step by ensemble size and number of groups (PYSPEEDY_AMD_CHUNKS).  Usage: bash tools/experiments/r04_group_sweep.sh "8 12 16" "1 2 3 4"
mkdir -p gpurun_out/r4sweep
for m in ${1:-8 16}; do
  for c in ${2:-1 2 4}; do
    PYSPEEDY_AMD_CHUNKS=$c python bench.py --scaling strong --members $m --no-legs --no-cpu-baseline --min-seconds 1 > gpurun_out/r4sweep/m${m}_c$c.json 2>/dev/null
    python - <<PY
import json
a=json.loads(open("gpurun_out/r4sweep/m${m}_c$c.json").read().strip().splitlines()[-1])
print("members $m groups $c:  %.4f ms/step  (%s)" % (a["ms_per_step"], a["config"].get("plan", "?")), flush=True)
PY
  done
done
