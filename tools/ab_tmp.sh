set -e
python -m pytest tests/test_step_gpu.py tests/test_run_gpu.py tests/test_physics_gpu.py tests/test_physics_replay_gpu.py tests/test_cfg5_gpu.py -m gpu -x -q 2>&1 | tail -3
run() { # label M
    python bench.py --scaling strong --members $2 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null
    python - "$1" $2 <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[1], 'ms/step %.4f'%d['ms_per_step'], ' '.join('%s %.1f'%(k['kernel'],k['avg_launch_us']) for k in d['roofline']['kernels']), flush=True)
PY
}
for M in 8 64; do
 for rep in 1 2 3; do
  for v in early2 batch batch2; do
  PYSPEEDY_AMD_LIB=$PWD/build_variants/lib_$v.so run $v $M
  done
 done
done
