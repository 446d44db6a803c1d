"""Dev tool: per-kernel SQ counters of a rocprofv3 --pmc run, normalised by GRBM_GUI_ACTIVE (busy cycles of the dispatch).
usage: python scripts/pmc_sq_summary.py <counter_collection.csv> [n]
The SQ counters of this rocprofv3 build are reported per XCD (1/8 of the chip): MFMA utilisation = 8 x SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x GRBM_GUI_ACTIVE); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles (MI355X_MICROARCH.md)."""
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
out = []
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:n]:
    gui = v.get("GRBM_GUI_ACTIVE", 1.0)
    wave = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    rec = {"kernel": k, "dispatches": cnt[k], "gui_cycles_per_dispatch": round(gui / max(cnt[k], 1)),
           "mfma_util": round(8 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui), 3),
           "waves_resident_avg": round(8 * 4 * wave / gui, 1),
           "wave_time_waiting_on_waitcnt_or_barrier": round(v.get("SQ_WAIT_ANY", 0.0) / wave, 3),
           "wave_time_issue_stalled": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
           "wave_time_issuing": round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3)}
    out.append(rec)
    print(json.dumps(rec))
