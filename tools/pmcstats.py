import csv, collections, re, sys
rows = csv.DictReader(open(f'gpurun_out/pmc_{sys.argv[1]}/p_counter_collection.csv'))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen=set(); ctrs=[]
for r in rows:
    m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
    name = (m.group(1) if m else r['Kernel_Name'][:30]) + '/' + r['Grid_Size']
    if 'at::' in r['Kernel_Name']: continue
    c = r['Counter_Name']
    if c not in ctrs: ctrs.append(c)
    agg[name][c] += float(r['Counter_Value'])
    if (r['Dispatch_Id'],name) not in seen:
        seen.add((r['Dispatch_Id'],name)); cnt[name]+=1
print('%-36s %5s '%('kernel','n')+' '.join('%14s'%c[:14] for c in ctrs))
keys = sorted(agg, key=lambda k:-cnt[k])
for k in keys[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    n=cnt[k]
    print('%-36s %5d '%(k[:36],n)+' '.join('%14.0f'%(agg[k][c]/n) for c in ctrs))
