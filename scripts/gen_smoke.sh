set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/gen_rar gpurun_out/gen_cham
timeout 600 python generate.py --model rar --synthetic true --outdir gpurun_out/gen_rar --conditioning 1,9,232 --num_samples_per_conditioning 2 --batch_size 6 --wm_method gentime --wm_seed_strategy linear --wm_split_strategy stratifiedrand --wm_context_size 1 --wm_delta 2.0 --wm_gamma 0.25 --temperature 1.0 --top_k 250 --top_p 0.92 --seed 1 --include_neural_compress false --include_diffpure false 2>&1 | tail -2
python - <<'PY'
import json
r=json.load(open('gpurun_out/gen_rar/results.json'))
print(len(r), r[0]['metrics'], r[-1]['conditioning'])
PY
