"""Time line of a few Mode A passes out of a rocprofv3 kernel trace (csv): start / end / duration per kernel relative to the solve's
x0 kernel, so that the idle gaps between dependent launches are visible.   python tools/c2_timeline.py trace.csv [first pass] [passes]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 200
count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NAMES = ['add_chain', 'rot_apply', 'jtn', 'select_partial', 'select_final', 'select_blk', 'select_comb', 'decision_gate', 'blk_comb', 'z_blk', 'step_tail', 'z_rows', 'remove_r', 'advance', 'jinv', 'x0', 'llt_panel',
         'llt_update', 'copyBuffer', 'fillBuffer', 'jrow', 'add_commit']
def nm(r):
    for s in NAMES:
        if s in r['Kernel_Name']:
            return s
    return r['Kernel_Name'][:30]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
start = [i for i, r in enumerate(rows) if nm(r) == 'x0'][-1]
t0 = int(rows[start]['Start_Timestamp'])
seg = rows[start:]
j = [i for i, r in enumerate(seg) if nm(r) == 'jtn']
for r in seg[j[first]:j[first + count] + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{nm(r):16s} queue {r['Queue_Id']:>2s}  start {s / 1e3:10.1f}  end {e / 1e3:10.1f}  dur {(e - s) / 1e3:6.1f} us")
print(f"{len(j)} passes in {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms after x0 started")
