#!/bin/bash
# Where in group 1's first step does group 2 start?  (a variant library that reads PYSPEEDY_AMD_OFFSET_AT: 1 behind the spectral ->
# grid launch, 2 behind the column kernel, 3 behind the grid -> spectral launch = the library's choice), for 20-step calls (the
# driver's command) and for the default regions, offset forced on (PYSPEEDY_AMD_GROUP_OFFSET=1) or off (0).
for rep in 1 2; do
for steps in 20 0; do
  for cfg in "0 3" "1 1" "1 2" "1 3"; do
    set -- $cfg
    if [ $steps = 0 ]; then args=""; else args="--steps 20 --warmup 5"; fi
    PYSPEEDY_AMD_LIB=build_variants/lib_offsetat.so PYSPEEDY_AMD_GROUP_OFFSET=$1 PYSPEEDY_AMD_OFFSET_AT=$2 python bench.py $args --no-legs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps/region %3s  offset %s at %s:  %.4f ms per step (min %.4f)' % (d['steps'], '$1', '$2', d['ms_per_step'], d['ms_per_step_min']))"
  done
done
done
