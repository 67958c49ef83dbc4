#!/usr/bin/env python3
"""Summarise GPX_PANEL_DEBUG=2 traces (stderr of a run): per launch the spine tasks with
their phase stamps and the mean duration of the product tasks by K.
usage: panel_trace_summary.py <log> [--last]   (--last: the last launch of the log only)"""
import sys
runs = []
cur = None
for l in open(sys.argv[1]):
    if l.startswith('panel trace'):
        cur = []
        runs.append((l.strip(), cur))
        continue
    f = l.split()
    if cur is not None and len(f) >= 9 and f[0].isdigit():
        cur.append(f)
if '--last' in sys.argv[2:]:
    runs = runs[-1:]
for head, r in runs:
    print(head)
    ntask = len(r)
    # spine tasks: leaves (op 0), fused solve + update + leaf (op 3 with a pivots-done stamp), and
    # since round 5 the followers (op 4: rows-in, update-done, pivots-done, R-out) with the
    # spine's solves (op 3 on the spine's workgroups: strips-done only) between them
    nworkers = max(int(f[5]) for f in r) + 1
    uf = any(f[6] == '4' for f in r)
    sp = [f for f in r if f[6] in ('0', '4') or (f[6] == '3' and len(f) >= 13 and float(f[12]) > 0)]
    if uf:
        nsp = max(int(f[5]) for f in sp) + 2        # (spine workgroups come first in the grid)
        sp += [f for f in r if f[6] == '3' and int(f[5]) < nsp and f not in sp]
    sp.sort(key=lambda f: float(f[2]))
    print('  spine: id op start end | strips-done(rows-in) update-done pivots-done R-out')
    for f in sp:
        print('   ', f[0], f[6], f[2], f[3], '|', ' '.join(f[10:14]))
    for K in ('64', '128', '256', '512', '896'):
        s = [float(f[3]) - float(f[2]) for f in r if f[6] in ('1', '2') and f[7] == K]
        if s:
            print('  products K=%s: mean %.1f min %.1f max %.1f us (n=%d)' % (K, sum(s) / len(s), min(s), max(s), len(s)))
    xs = [float(f[3]) - float(f[2]) for f in r if f[6] == '3' and not (len(f) >= 13 and float(f[12]) > 0)]
    if xs:
        print('  XS tasks: mean %.1f us (n=%d)' % (sum(xs) / len(xs), len(xs)))
    ufs = sorted(float(f[12]) for f in r if f[6] == '4' and len(f) >= 13 and float(f[12]) > 0)
    if len(ufs) >= 6:
        n3 = len(ufs) // 3
        print('  per tile (pivots-done to pivots-done of the followers): whole launch %.1f us, '
              'first third %.1f, last third %.1f' % ((ufs[-1] - ufs[0]) / (len(ufs) - 1),
              (ufs[n3] - ufs[0]) / n3, (ufs[-1] - ufs[-1 - n3]) / n3))
    print('  last end %.1f us, %d tasks' % (max(float(f[3]) for f in r), ntask))
