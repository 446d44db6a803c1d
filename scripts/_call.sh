mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/c15_tests.log
bash scripts/final_prof.sh r05f > gpurun_out/c15.log 2>&1
cat gpurun_out/c15_tests.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['decode_step'], d['stage_seconds_per_batch'], d['roofline']['frac'], d['roofline']['avg_us'], d['roofline']['traffic'])
for k,v in d['secondary'].items(): print(k, v['images_per_s'], v['ms_per_step'], v.get('vq_decode_s'), v.get('vq_encode_s'))
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
