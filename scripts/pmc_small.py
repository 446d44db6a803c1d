"""Dev tool (round 6): gpurun_out/pmc_small/<role>_{FETCH_SIZE,WRITE_SIZE}/.../*counter_collection.csv of `scripts/pmc_role.py <role> 5`
-> profiles/r06_pmc_small_b5.json: HBM bytes per launch of the small-batch kernels (k_sgemv / k_sattn, wmar_amd/csrc/decode_small.h)
against their algorithmic bytes.  read = 2 x FETCH_SIZE KiB on gfx950 (MI355X_MICROARCH.md, HBM section), write = WRITE_SIZE KiB."""
import csv, glob, json, re

D, B = 1536, 5
ALG = {"qkv": ("k_sgemv<5, 3, 3, 0>", 3 * D * D * 4 + B * D * 4 + 3 * B * D * 4, "QKV weights + the residual rows (read once) + q / k / v rows written"),
       "proj": ("k_sgemv<5, 3, 1, 1>", D * D * 4 + 3 * B * D * 4, "projection weights + attention output and residual rows read + residual rows written"),
       "fc1": ("k_sgemv<5, 3, 4, 2>", 4 * D * D * 4 + B * D * 4 + 4 * B * D * 4, "FC1 weights + residual rows + the hidden rows written"),
       "fc2": ("k_sgemv<5, 3, 2, 3>", 4 * D * D * 4 + 4 * B * D * 4 + 2 * B * D * 4, "FC2 weights + hidden rows + residual rows read and written"),
       "attn": ("k_sattn", 2.0 * B * D * 4 * 128 + 2 * B * D * 4, "K and V rows of 128 cached positions for 5 x 24 (sequence, head) pairs + q read, output written")}
out = {"round": 6, "batch": B, "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python scripts/pmc_role.py <role> 5 (separate passes)",
       "correction": "read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section); write bytes = WRITE_SIZE x 1024", "roles": {}}
for role, (kname, alg, what) in ALG.items():
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"gpurun_out/pmc_small/{role}_{ctr}/*/*counter_collection.csv")[0]
        rows = [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"] and r["Counter_Name"] == ctr][-48:]
        vals[ctr] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
        n = len(rows)
    hbm = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    out["roles"][role] = {"kernel": kname, "launches_averaged": n, "FETCH_SIZE_KiB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": vals["WRITE_SIZE"],
                          "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_are": what, "ratio": round(hbm / alg, 4)}
    print(role, out["roles"][role]["ratio"], hbm, alg)
json.dump(out, open("gpurun_out/r06_pmc_small_b5.json", "w"), indent=1)
