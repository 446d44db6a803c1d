"""Per-kernel-role averages of the last full decode step in a rocprofv3 kernel trace (dev tool)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
roles = sys.argv[2].split(',') if len(sys.argv) > 2 else ['qkv', 'o', 'w13', 'w2']
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_sample_fused' in r['Kernel_Name'] or 'k_gumbel_sample' in r['Kernel_Name']]
seq = rows[idx[-3] + 1: idx[-2] + 1]
agg, cnt, g = collections.OrderedDict(), collections.Counter(), 0
for r in seq:
    n = r['Kernel_Name']; d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if 'k_bgemm<' in n and ', 0>' in n:
        key = 'gemm_' + roles[g % len(roles)]; g += 1
    else:
        key = n.split('(')[0][:44]
    agg[key] = agg.get(key, 0) + d; cnt[key] += 1
for k, v in agg.items():
    print(f"{k:46s} n={cnt[k]:4d} avg={v/cnt[k]/1e3:8.2f} us total={v/1e3:9.1f} us")
print("step wall", (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3, "us; sum of kernels", sum(agg.values()) / 1e3)
