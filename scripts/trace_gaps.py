"""Gaps between consecutive search kernels in a rocprofv3 kernel trace (csv): where an exchange-path step loses time."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = collections.Counter(r['Kernel_Name'][:60] for r in rows)
print(names.most_common(8))
srch = [i for i, r in enumerate(rows) if 'hx_lean_f32_kernel<100, 2>' in r['Kernel_Name'] or 'hx_lean_q8_kernel<2>' in r['Kernel_Name']]
srch = srch[-int(sys.argv[2]) if len(sys.argv) > 2 else len(srch) // 2:]  # the last K launches: the timed region
gaps = []
between = collections.Counter()
for a, b in zip(srch[:-1], srch[1:]):
    g = (int(rows[b]['Start_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3
    gaps.append(g)
    for k in range(a + 1, b):
        between[rows[k]['Kernel_Name'][:50]] += 1
import statistics
print('search kernels: %d, gap us: median %.1f mean %.1f p90 %.1f max %.1f' % (len(srch), statistics.median(gaps), sum(gaps) / len(gaps), sorted(gaps)[int(0.9 * len(gaps))], max(gaps)))
big = [g for g in gaps if g > 5]
print('gaps > 5 us: %d of %d, their mean %.1f us' % (len(big), len(gaps), sum(big) / max(1, len(big))))
print('kernels between search kernels:', between.most_common(8))
hist = collections.Counter(min(200, int(g // 10) * 10) for g in gaps)
print('gap histogram (us, floor to 10):', sorted(hist.items()))
