"""Timeline of the recurrence from a rocprofv3 kernel trace (csv): per-kernel
durations, gaps between dependent launches per queue, step periods, and a
listing of a few consecutive steps of the forward and the reverse chain."""
import csv, re, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    m = re.search(r'(\w+_kernel)(<[^>]*>)?', r['Kernel_Name'])
    r['n'] = (m.group(1) + (m.group(2) or '')) if m else r['Kernel_Name'][:40]
rows.sort(key=lambda r: r['s'])
ce = [i for i, r in enumerate(rows)
      if r['n'].startswith(('softmax_ce_kernel', 'output_ce_kernel'))]
lo = ce[-2] if len(ce) > 1 else 0
seg = rows[lo:ce[-1]] if len(ce) > 1 else rows
# keep the last training step: from the previous step's CE to this one's is
# backward(prev) + forward(this); simpler: take everything after the first CE
seg = rows[ce[0]:] if len(ce) == 1 else rows[ce[-2]:]
t0 = seg[0]['s']
print('segment span ms %.3f kernels %d' % ((seg[-1]['e'] - t0) / 1e6, len(seg)))
dur = collections.defaultdict(list)
for r in seg:
    dur[r['n']].append((r['e'] - r['s']) / 1e3)
print('%-44s %6s %9s %9s %9s' % ('kernel', 'calls', 'med us', 'mean us', 'total ms'))
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    d2 = sorted(d)
    print('%-44s %6d %9.2f %9.2f %9.3f' % (n[:44], len(d), d2[len(d) // 2], sum(d) / len(d), sum(d) / 1e3))
# gaps per queue between consecutive kernels
byq = collections.defaultdict(list)
for r in seg:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    gaps = sorted((b['s'] - a['e']) / 1e3 for a, b in zip(rs, rs[1:]))
    if len(gaps) > 10:
        busy = sum(r['e'] - r['s'] for r in rs) / 1e6
        print('queue %s: %d kernels, busy %.2f ms, gap median %.2f us, mean %.2f us, sum of gaps %.2f ms' % (
            q, len(rs), busy, gaps[len(gaps) // 2], sum(gaps) / len(gaps), sum(gaps) / 1e3))
# step markers: the forward stack step (or, with it fused into the cells, the
# layer-0 cell's XT instantiation), the stack adjoint (fused or its finish launch)
FWD_MARK = ('stack_step_fwd_kernel'
            if sum(r['n'].startswith('stack_step_fwd_kernel') for r in seg) > 100
            else 'gru_cell_fwd_kernel<false, true>')
BWD_MARK = ('stack_bwd_finish_kernel'
            if any(r['n'].startswith('stack_bwd_finish_kernel') for r in seg)
            else 'stack_bwd_kernel')
for name in (FWD_MARK, BWD_MARK):
    for q, rs in byq.items():
        ks = [r for r in rs if r['n'].startswith(name)]
        if len(ks) > 10:
            per = sorted((b['s'] - a['s']) / 1e3 for a, b in zip(ks, ks[1:]))
            print('%s on queue %s: step period median %.2f us (n=%d)' % (name, q, per[len(per) // 2], len(ks)))
def listing(name, skip, count):
    ks = [i for i, r in enumerate(seg) if r['n'].startswith(name)]
    if len(ks) <= skip + 3:
        return
    a, b = ks[skip], ks[skip + 3]
    base = seg[a]['s']
    print('--- three steps from the %dth %s' % (skip, name))
    for r in seg[a:b + 1]:
        print('  q%-3s %-40s start %8.2f us  dur %6.2f us' % (r['Queue_Id'], r['n'][:40], (r['s'] - base) / 1e3, (r['e'] - r['s']) / 1e3))
listing(FWD_MARK, 40, 3)
listing(BWD_MARK, 40, 3)
