#!/bin/bash
# GPU side of scripts/fmac_experiment.sh: PASSES teacher-forced passes of the full-size Taming model per variant; prints mismatching passes.
#   usage: scripts/fmac_experiment_run.sh [passes]   -> gpurun_out/fmac_experiment.log
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
P=${1:-12}
echo "variant mismatching_passes_of_$((P-1))" > gpurun_out/fmac_experiment.log
WMAR_FAST_WEIGHTS=1 timeout 200 python scripts/stress_logits.py 1 $P > gpurun_out/fmac_control.txt 2>&1
echo "shipped $(tail -1 gpurun_out/fmac_control.txt)" >> gpurun_out/fmac_experiment.log
for d in build_alt/fmac_e*; do
  WMAR_FAST_WEIGHTS=1 WMAR_ROOT=$R/$d timeout 200 python scripts/stress_logits.py 1 $P > gpurun_out/fmac_tmp.txt 2>&1
  echo "$(basename $d) $(tail -1 gpurun_out/fmac_tmp.txt) | $(grep -c 'differ' gpurun_out/fmac_tmp.txt) first: $(grep differ gpurun_out/fmac_tmp.txt | head -1 | cut -c1-110)" >> gpurun_out/fmac_experiment.log
done
rm -f gpurun_out/fmac_tmp.txt
cat gpurun_out/fmac_experiment.log
