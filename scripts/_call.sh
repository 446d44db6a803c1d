mkdir -p gpurun_out
bash scripts/final_prof.sh r05b > gpurun_out/c9.log 2>&1
tail -3 gpurun_out/c9.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05b_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['decode_step'], d['stage_seconds_per_batch'], d['roofline']['frac'], d['roofline']['avg_us'])
for k,v in d['secondary'].items(): print(k, v['images_per_s'], v['ms_per_step'], v.get('vq_decode_s'), v.get('vq_encode_s'))
print(d['parity'])
PY
