"""Timeline of one training step from a rocprofv3 kernel trace: phases of the
main queue, stalls of the reverse chain, side-stream kernels."""
import csv, re, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/prof_{tag}/r_kernel_trace.csv')))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
    r['n'] = m.group(1) if m else r['Kernel_Name'][:40]
rows.sort(key=lambda r: r['s'])
ce = [i for i, r in enumerate(rows) if r['n'] == 'softmax_ce_kernel']
seg = rows[ce[-2]:ce[-1]]
t0 = seg[0]['s']
print('step span ms', (seg[-1]['e'] - t0) / 1e6, 'kernels', len(seg))
mainq = seg[0]['Queue_Id']
bw = [r for r in seg if r['n'] == 'stack_bwd_kernel']
prev = None; slow = 0.0; base = None
per = sorted((b['s'] - a['s']) / 1e3 for a, b in zip(bw, bw[1:]))
med = per[len(per) // 2]
print('bwd step period: median %.1f us, sum of excess over median %.2f ms' % (med, sum(p - med for p in per if p > med and p < 3000) / 1e3))
fw = [r for r in seg if r['n'] == 'stack_step_fwd_kernel']
per = sorted((b['s'] - a['s']) / 1e3 for a, b in zip(fw, fw[1:]))
print('fwd step period: median %.1f us' % per[len(per) // 2])
side = [r for r in seg if r['Queue_Id'] != mainq and (r['e'] - r['s']) > 100000]
for r in side[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print('side %-22s start %.3f ms dur %.3f ms grid %s' % (r['n'], (r['s'] - t0) / 1e6, (r['e'] - r['s']) / 1e6, r['Grid_Size_X']))
