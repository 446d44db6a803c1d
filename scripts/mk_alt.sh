#!/bin/bash
# dev: snapshot the current package (python + built libwmar_hip.so) under build_alt/<name>/ so that the perf scripts can A/B two
# builds on the same box:  WMAR_ROOT=build_alt/<name> python scripts/perf_gpt.py 64 256
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
python -m wmar_amd.build > /dev/null
d=build_alt/$1
rm -rf $d; mkdir -p $d/wmar_amd
for f in wmar_amd/*.py wmar_amd/libwmar_hip.so; do cp $f $d/wmar_amd/; done
for sub in models watermarking utils augmentations assets; do [ -d wmar_amd/$sub ] && cp -r wmar_amd/$sub $d/wmar_amd/; done
find $d -name __pycache__ -prune -exec rm -rf {} +
echo "snapshot in $d"
