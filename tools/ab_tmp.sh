set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in dynb pref2; do
    PYSPEEDY_AMD_LIB=$PWD/build_variants/lib_$v.so python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null
    python - "$v" cfg5 <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[1], 'ms/step %.4f'%d['ms_per_step'], ' '.join('%s %.1f'%(k['kernel'],k['avg_launch_us']) for k in d['roofline']['kernels']), flush=True)
PY
done
