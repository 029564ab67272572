"""Per-launch means of the SQ counters collected by tools/pmc_sq.sh for the dominant kernel (sum over the counter's
dimension instances, mean over the launches), plus the ratios the round report quotes."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else "gi_batch_kernel"
out = {}
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, d in per.items():
        vals = list(d.values())
        out[name] = sum(vals) / len(vals)
g = out.get
d = {"kernel": match, "per_launch": out}
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    d["fractions_of_wave_cycles"] = {k: g(k, 0.0) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                                 "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")}
if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
    d["note_units"] = "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)"
if g("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_frac_of_lds_active"] = g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")
if g("SQ_WAVES"):
    d["insts_per_wave"] = {k: g(k, 0.0) / g("SQ_WAVES") for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH")}
print(json.dumps(d, indent=1))
