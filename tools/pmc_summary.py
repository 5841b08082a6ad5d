"""Per-kernel mean of every counter in rocprofv3 counter_collection CSVs."""
import collections, csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:58]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if 'rocclr' in k:
            continue
        print(k)
        for c, vals in sorted(v.items()):
            print(f"    {c:32s} n={len(vals):3d} mean={sum(vals)/len(vals):16.1f}")
