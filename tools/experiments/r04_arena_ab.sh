set -e
mkdir -p gpurun_out/r4arena
for round in 1 2; do
for v in default perarray align4k skew; do
  if [ $v == default ]; then unset PYSPEEDY_AMD_LIB; else export PYSPEEDY_AMD_LIB=build_variants/lib_$v.so; fi
  python bench.py --no-legs --no-cpu-baseline --serial-plan --steps 360 --warmup 36 > gpurun_out/r4arena/bench_${v}_$round.json 2> gpurun_out/r4arena/bench_${v}_$round.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4arena/bench_${v}_$round.json").read().strip().splitlines()[-1])
print("$v $round", d["ms_per_step"], d.get("ms_per_step_min"), d["roofline"].get("serial_plan_ms_per_step"), [ (k["kernel"], round(k["avg_launch_us"],1)) for k in d["roofline"].get("kernels",[])])
PY
done
done
unset PYSPEEDY_AMD_LIB
python - <<PY
import sys; sys.path.insert(0,".")
import pyspeedy_amd
from pyspeedy_amd.model import EnsembleModel
sp=pyspeedy_amd.ModSpectral(0)
for M in (1,8,64):
    m=EnsembleModel(sp,M); r,u=m.memory(); print("M",M,"reserved MB",r/2**20,"used MB",u/2**20,"per member used",u/2**20/M); m.close()
PY
