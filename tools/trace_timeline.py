"""Print the kernel timeline of the last FRI commit found in a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if ('leaf_hash_kernel' in r['Kernel_Name'] or 'leaf_subtree_kernel' in r['Kernel_Name'])]
s = idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
prev_end = t0
tot = 0
brief = len(sys.argv) > 2
for r in rows[s:]:
    st = int(r['Start_Timestamp']); en = int(r['End_Timestamp'])
    g = int(r.get('Grid_Size_X') or r.get('Grid_Size'))
    if not brief or (st - prev_end) > 3000 or (en - st) > 30000:
        print(f"{(st-t0)/1e3:9.1f} us gap {(st-prev_end)/1e3:7.1f} dur {(en-st)/1e3:8.1f} grid {g:9d} {r['Kernel_Name'][:44]}")
    prev_end = en; tot += en - st
print('total', (prev_end - t0) / 1e3, 'kernel time', tot / 1e3)
