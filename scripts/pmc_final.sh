cd $GRAFT_REPO_ROOT
bash scripts/pmc_passes.sh attn qkv proj fc1 fc2 > /dev/null 2>&1
python scripts/pmc_summarize.py attn qkv proj fc1 fc2
mkdir -p gpurun_out/pmc_out
cp profiles/pmc_*.json gpurun_out/pmc_out/
python - <<'PY'
import csv, glob, re
K = {"attn": "k_attn_decode", "qkv": "k_qkvx_bx", "proj": "k_bx_xr", "fc1": "k_fc1x", "fc2": "k_gemm"}
for role, kn in K.items():
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/pmc/{role}_{ctr}/*/*counter_collection.csv")[0]
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if re.search(r"(^|::)" + kn + r"[<(]", r["Kernel_Name"]) and r["Counter_Name"] == ctr][-48:]
        with open(f"gpurun_out/pmc_out/r06_pmc_{ctr}_{role}.csv", "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
rm -rf gpurun_out/pmc
ls -la gpurun_out/pmc_out | head -20
