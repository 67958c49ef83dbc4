#!/usr/bin/env python3
"""In-situ behaviour of the diagonal-panel launches (GPX_PANEL_DEBUG=2 log of an evaluation
with the look-ahead on): per launch, when each workgroup claimed its first task (how long
the 157-KB workgroups waited for a free CU beside the products), the spine's tile period
and the launch's last task end. usage: panel_insitu.py <log>"""
import sys
runs = []
cur = None
for l in open(sys.argv[1], errors='replace'):
    if l.startswith('panel trace'):
        cur = []
        runs.append(cur)
        continue
    f = l.split()
    if cur is not None and len(f) >= 9 and f[0].isdigit() and f[4] == '|':
        cur.append(f)
for i, r in enumerate(runs):
    if not r:
        continue
    first = {}
    for f in r:
        wg = int(f[5])
        first[wg] = min(first.get(wg, 1e30), float(f[1]))
    starts = sorted(first.values())
    sp = sorted((f for f in r if f[6] in ('0', '3') and float(f[12]) > 0), key=lambda f: float(f[12]))
    piv = [float(f[12]) for f in sp]
    per = [b - a for a, b in zip(piv, piv[1:])]
    last = max(float(f[3]) for f in r)
    n = len(starts)
    print('launch %2d: %3d wgs, first claims at p50 %.0f p90 %.0f max %.0f us | tile period %s us | last end %.0f us'
          % (i, n, starts[n // 2], starts[int(n * 0.9)], starts[-1],
             ' '.join('%.0f' % p for p in per), last))
