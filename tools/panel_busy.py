#!/usr/bin/env python3
"""Where the workgroups of a traced panel launch (GPX_PANEL_DEBUG=2, last launch of the log)
spend their time: per 100-us bin the share of the worker pool that runs factorisation tasks,
inverse tasks (I1: the sums, I2: the product with W_ss) and that waits inside a claimed task;
task durations of the inverse sums by K; when the factorisation ends and when the launch ends.
panel_busy.py LOG [T_plus_E]"""
import re, sys
import numpy as np
lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if l.startswith('panel trace T=')]
i0 = starts[-1]
T = int(re.search(r'T=(\d+)', lines[i0]).group(1))
TW = int(sys.argv[2]) if len(sys.argv) > 2 else T + 1
rows = []
for l in lines[i0 + 1:]:
    m = re.match(r'\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+) \|\s*(\d+) (\d+)\s+(-?\d+)\s+(-?\d+) \|', l)
    if not m:
        if l.startswith('  in-kernel'): break
        continue
    rows.append([float(x) for x in m.groups()])
r = np.array(rows)
claim, start, end, wg, op, K, sig = r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
cls = np.where(sig < TW * TW, 0, np.where(sig < TW * TW + T * T, 1, np.where(sig < TW * TW + 2 * T * T, 2, 3)))
chain = (op == 0) | (op == 4) | ((op == 3) & (wg < 9))
nwg = int(wg.max()) + 1
print('T=%d tasks=%d workgroups=%d; factorisation tasks end %.1f us, launch ends %.1f us' %
      (T, len(r), nwg, end[cls == 0].max(), end.max()))
work = ~chain
nworkers = len(set(wg[work]))
edges = np.arange(0, end.max() + 100, 100.0)
print('bin(us)  factor  I1    I2    waiting  idle   (share of %d workers)' % nworkers)
for a, b in zip(edges[:-1], edges[1:]):
    def share(s, e, m):
        return np.sum(np.clip(np.minimum(e[m], b) - np.maximum(s[m], a), 0, None)) / ((b - a) * nworkers)
    f = [share(start, end, work & (cls == c)) for c in (0, 1, 2)]
    w = share(claim, start, work)
    print('%5d    %.2f   %.2f  %.2f  %.2f     %.2f' % (a, f[0], f[1], f[2], w, 1 - sum(f) - w))
for c, name in ((1, 'I1'), (2, 'I2'), (0, 'factor products')):
    m = work & (cls == c) & (op == 2 if c else op >= 0)
    print(name, 'by K: ' + '  '.join('%d: n=%d %.1f us' % (k, np.sum(m & (K == k)), np.mean((end - start)[m & (K == k)]))
                                     for k in sorted(set(K[m]))[:12]))
print('sum of task time: factor %.1f ms, I1 %.1f ms, I2 %.1f ms' %
      tuple(np.sum((end - start)[work & (cls == c)]) / 1e3 for c in (0, 1, 2)))
