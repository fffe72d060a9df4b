import csv, glob, collections, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for d in sorted(glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(d)) if pat in r['Kernel_Name'] and 'finish' not in r['Kernel_Name']]
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    if not rows: continue
    r0 = rows[0]
    print(d.split('/')[-3], r0['Kernel_Name'][:70], {k: r0[k] for k in ('Grid_Size', 'LDS_Block_Size', 'VGPR_Count', 'Scratch_Size') if k in r0})
    for k, v in acc.items():
        print(f"   {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
