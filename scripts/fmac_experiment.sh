#!/bin/bash
# Round-4 experiment on the round-3 fp64 finding: the keeper workgroups' sums of squares in k_qkvx_bx as a chain of
# dependent v_fmac_f64 again (QX_FMAC_EXP = 1), with an s_nop 1 only between the v_cvt_f64_f32 producers and the chain (2) or only between
# the dependent fmacs (3); 4 = the round-2 REDUCTION (64-bit __shfl_xor(., 32), lanes 0..31 publish) with the unfused products, 5 = the
# round-2 code as it was (that reduction AND the fmac chain); at several code paddings in front of the kernel (QX_PAD s_nops: the failure rate of round 3 moved between 0 and
# 100 % with the padding alone).  Builds one library per variant into build_alt/<name>/ (CPU side; git-ignored, travels with gpurun);
# scripts/fmac_experiment_run.sh runs scripts/stress_logits.py against each on the GPU box.
#   usage: scripts/fmac_experiment.sh "<exp list>" "<pad list>"      e.g.  "1 2 3" "0 1 2 3 5 8"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python -m wmar_amd.build > /dev/null
EXPS=${1:-"1 2 3"}; PADS=${2:-"0 1 2 3 5 8"}
build_one() {
  e=$1; p=$2; d=build_alt/fmac_e${e}_p${p}
  mkdir -p $d/wmar_amd $d/objs
  for f in wmar_amd/*.py; do cp $f $d/wmar_amd/; done
  for sub in models watermarking utils augmentations assets; do [ -d wmar_amd/$sub ] && cp -r wmar_amd/$sub $d/wmar_amd/; done
  find $d -name __pycache__ -prune -exec rm -rf {} +
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -x hip \
      $( [ $e = 4 ] && echo "-DQX_OLD_RED" ) $( [ $e = 5 ] && echo "-DQX_OLD_RED -DQX_FMAC_EXP=1" ) $( [ $e -le 3 ] && echo "-DQX_FMAC_EXP=$e" ) -DQX_PAD=$p -c wmar_amd/csrc/gpt.hip -o $d/objs/gpt.hip.o
  objs=""
  for o in keytable.cpp watermark.hip gumbel.hip rar.hip cham.hip vqgan.hip; do objs="$objs wmar_amd/build/$o.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/wmar_amd/libwmar_hip.so $objs $d/objs/gpt.hip.o
  rm -rf $d/objs
  echo built $d
}
export -f build_one
for e in $EXPS; do for p in $PADS; do echo "$e $p"; done; done | xargs -P 4 -L 1 bash -c 'build_one $0 $1'
ls build_alt | grep fmac_ | wc -l
