#!/usr/bin/env python3
"""For every fused spine task XSF(t) of a GPX_PANEL_DEBUG=2 trace of a whole-matrix launch:
when did the tasks it waits for end (last signal on the counters of tile (t-1,t) and (t,t)),
when were they claimed / started, and when did XSF(t) itself start.
usage: r04_chain_deps.py <log> [launch index]"""
import sys
runs, cur = [], None
for l in open(sys.argv[1]):
    if l.startswith('panel trace'):
        cur = []
        runs.append((l.strip(), cur))
        continue
    f = l.split()
    if cur is not None and len(f) >= 9 and f[0].isdigit():
        cur.append(f)
head, r = runs[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
T = int(head.split('T=')[1].split()[0])
TW = T + 1
rows = [(int(f[0]), float(f[1]), float(f[2]), float(f[3]), int(f[5]), int(f[6]), int(f[7]), int(f[8]),
         [float(x) for x in f[10:14]]) for f in r]
spine = sorted([x for x in rows if x[5] == 3 and x[8][2] > 0], key=lambda x: x[2])
by_sig = {}
for x in rows:
    by_sig.setdefault(x[7], []).append(x)
print(head)
print(' t | XSF start | last update of X(t-1,t): claim start end wg K | of D(t,t): claim start end | prev pivots-done')
prev = None
for i, x in enumerate(spine):
    t = i + 1
    cx, cd = (t - 1) * TW + t, t * TW + t
    ux = [y for y in by_sig.get(cx, []) if y[5] in (1, 2)]
    ud = [y for y in by_sig.get(cd, []) if y[5] in (1, 2)]
    lx = max(ux, key=lambda y: y[3]) if ux else None
    ldd = max(ud, key=lambda y: y[3]) if ud else None
    print('%2d | %8.1f | %s | %s | %s' % (
        t, x[2],
        '%8.1f %8.1f %8.1f wg%3d K%4d' % (lx[1], lx[2], lx[3], lx[4], lx[6]) if lx else '-',
        '%8.1f %8.1f %8.1f' % (ldd[1], ldd[2], ldd[3]) if ldd else '-',
        '%8.1f' % prev if prev else '-'))
    prev = x[8][2]
