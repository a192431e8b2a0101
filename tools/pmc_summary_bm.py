import csv, glob, sys, collections, re
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n[:40]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + '*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        res[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(res):
    if not ('_bm' in k): continue
    print(k)
    for c, v in sorted(res[k].items()):
        print(f'   {c:26s} {sum(v)/len(v):16.1f}  (n={len(v)})')
