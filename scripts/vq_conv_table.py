"""Dev tool: per-convolution times of one Taming VQGAN decode + encode at batch 64 (dev build: WMAR_VQ_TRACE=1 prints one line per conv).
   WMAR_ROOT=build_alt/dev WMAR_VQ_TRACE=1 python scripts/perf_vq.py 64 2> trace.txt ; python scripts/vq_conv_table.py trace.txt"""
import collections, sys
rows = [l.split() for l in open(sys.argv[1]) if l.startswith("conv")]
n = len(rows) // 3 if len(rows) >= 306 else len(rows)      # three repetitions: keep the last
rows = rows[-n:]
agg = collections.OrderedDict()
for r in rows:
    key = " ".join(r[1:6]); us = float(r[6]); tf = float(r[8])
    a = agg.setdefault(key, [0, 0.0, tf]); a[0] += 1; a[1] += us
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:26]:
    print(f"{k:34s} n={v[0]:3d}  {v[1] / 1e3:7.2f} ms  {v[2]:6.1f} TF/s")
print("convs:", n, "total", round(tot / 1e3, 2), "ms")
