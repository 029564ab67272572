"""Print the top rows of a rocprofv3 --kernel-trace --stats CSV (tools/kstats.py DIR [rows])."""
import csv
import glob
import sys

d = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:rows]:
    name = r["Name"]
    name = name[name.find("jrlqp") :][:70] if "jrlqp" in name else name[:70]
    print(name.ljust(70), r["Calls"].rjust(7), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12), r["Percentage"].rjust(7))
