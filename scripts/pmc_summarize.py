"""Dev tool: gpurun_out/pmc/<role>_{FETCH_SIZE,WRITE_SIZE}/.../*counter_collection.csv -> profiles/pmc_<role>.json
(HBM bytes per launch of the role's kernel: read = 2 x FETCH_SIZE KiB on gfx950, MI355X_MICROARCH.md HBM section; write = WRITE_SIZE KiB)."""
import csv, glob, json, re, sys

ALG = {"attn": ("k_attn_decode", 2.0 * 64 * 1536 * 4 * 128 + 64 * 3 * 1536 * 4 * 7, "K and V rows of 128 cached positions for 64 x 24 (sequence, head) pairs + the 7 QKV pieces of the new token"),
       "qkv": ("k_qkvx_bx", 3 * 1536 * 1536 * 4 + 7 * 64 * 1536 * 4 + 64 * 1536 * 4 + 7 * 64 * 4608 * 4, "QKV weights + x and 6 FC2 slabs (read once) + x' + 7 split-K pieces written"),
       "fc1": ("k_fc1x", 4 * 1536 * 1536 * 4 + 64 * 1536 * 4 + 64 * 6144 * 4, "FC1 weights (both packings of a tile are read: 24-column tiles) + x (read once; every workgroup re-reads it from L2) + the hidden activation written"),
       "fc2": ("k_gemm", 4 * 1536 * 1536 * 4 + 64 * 6144 * 4 + (16 * 6 + 32 * 5) / 48 * 64 * 1536 * 4, "FC2 weights + the hidden activation (read once) + 5.33 split-K slabs written"),
       # round 4: k_bx_xr = projection + XCD-local reduction + residual fold + LN2 statistics in one launch
       "proj": ("k_bx_xr", 1536 * 1536 * 4 + 64 * 1536 * 6 + 4 * 64 * 1536 * 4 + 2 * 64 * 1536 * 4, "proj weights + the attention output as bf16 pieces (read once) + 4 split-K slabs written (read back inside the XCD: L2) + the residual stream read and written")}
ROUND = 6
for role in sys.argv[1:]:
    kname, alg, what = ALG[role]
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/pmc/{role}_{ctr}/*/*counter_collection.csv")[0]
        # exact kernel (not the pack kernels whose names contain it), the replayed launches only (the last 48 dispatches)
        rows = [r for r in csv.DictReader(open(f)) if re.search(r"(^|::)" + kname + r"[<(]", r["Kernel_Name"]) and r["Counter_Name"] == ctr]
        rows = rows[-48:]
        vals[ctr] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
        n = len(rows)
    hbm = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    out = {"kernel": kname, "role": role, "round": ROUND, "launches_averaged": n, "FETCH_SIZE_KiB_per_launch": vals["FETCH_SIZE"],
           "WRITE_SIZE_KiB_per_launch": vals["WRITE_SIZE"], "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
           "algorithmic_bytes_are": what, "ratio": hbm / alg,
           "correction": "read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section); write bytes = WRITE_SIZE x 1024",
           "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python scripts/pmc_role.py {role} (separate passes; scripts/pmc_passes.sh)"}
    json.dump(out, open(f"profiles/pmc_{role}.json", "w"), indent=1)
    print(role, json.dumps({k: out[k] for k in ("FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "ratio")}))
