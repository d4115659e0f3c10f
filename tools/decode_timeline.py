"""Kernel sequence of one decode step (between two logits_choose launches) from a
rocprofv3 --kernel-trace csv: names, durations, gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'logits_choose' in r['Kernel_Name']]
# a pair of consecutive launches well inside a generate call
k = len(idx) // 2
a, b = idx[k], idx[k + 1]
prev_end = int(rows[a]['End_Timestamp'])
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:8.2f}  gap {(s - prev_end) / 1e3:7.2f}  "
          f"{r['Kernel_Name'][:70]}")
    prev_end = e
print('step period', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'us')
