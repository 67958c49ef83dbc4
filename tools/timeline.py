#!/usr/bin/env python3
"""Flop-rate timeline of one evaluation: the launch log of the tile engine
(GPX_GEMM_LOG) joined with a rocprofv3 kernel trace, every launch's flops spread evenly
over its [start, end], summed per time bin. Shows where an evaluation falls below the
rate of its big launches (start-up, chain stalls, tails).
usage: timeline.py <log> <kernel_trace.csv> [bin_ms] [eval_index_from_end]"""
import collections
import csv
import sys

UPPER, KLO_M, KHI_M, KLO_N, KHI_N = 1, 2, 4, 8, 16


def live_flops(tile, M, N, K, flags, kshift):
    tot = 0
    for m0 in range(0, M, tile):
        for n0 in range(0, N, tile):
            if flags & UPPER and n0 + tile <= m0:
                continue
            klo, khi = 0, K
            if flags & KLO_M: klo = max(klo, m0 - kshift)
            if flags & KHI_M: khi = min(khi, m0 + tile)
            if flags & KLO_N: klo = max(klo, n0 - kshift)
            if flags & KHI_N: khi = min(khi, n0 + tile)
            tot += max(0, khi - klo)
    return 2.0 * tile * tile * tot


log = [l.split() for l in open(sys.argv[1])]
rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
binms = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
back = int(sys.argv[4]) if len(sys.argv) > 4 else 1
gem = [r for r in rows if 'gemm_f64_kernel' in r['Kernel_Name']]
byq = collections.defaultdict(list)
for r in gem:
    byq[r['Queue_Id']].append(r)
bys = collections.defaultdict(list)
for l in log:
    bys[l[0]].append(l)
launches = []          # (start ns, end ns, flops, queue)
for sid, ls in bys.items():
    seq = [int(l[10]) for l in ls]
    for q, rs in byq.items():
        qseq = [int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) //
                int(r['Workgroup_Size_X']) for r in rs]
        if qseq == seq:
            for l, r in zip(ls, rs):
                ta, tb, tile, M, N, K, flags, kshift, part, wgs = [int(x) for x in l[1:11]]
                fl = live_flops(tile, M, N, K, flags, kshift) if part == 0 else \
                    2.0 * tile * tile * K * wgs
                launches.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), fl, q))
            break
# evaluations are separated by kbuild_kernel launches
kb = [int(r['Start_Timestamp']) for r in rows if 'kbuild_kernel' in r['Kernel_Name']]
tr = [int(r['End_Timestamp']) for r in rows if 'trace_grad' in r['Kernel_Name']]
t0 = kb[-back]
t1 = min(t for t in tr if t > t0)
others = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:24], r['Queue_Id'])
          for r in rows if 'gemm_f64' not in r['Kernel_Name'] and t0 <= int(r['Start_Timestamp']) < t1]
nb = int((t1 - t0) / 1e6 / binms) + 1
fl = [0.0] * nb
busy = collections.defaultdict(lambda: [0.0] * nb)
for s, e, f, q in launches:
    if e <= t0 or s >= t1:
        continue
    for b in range(nb):
        lo, hi = t0 + b * binms * 1e6, t0 + (b + 1) * binms * 1e6
        ov = max(0.0, min(e, hi) - max(s, lo))
        if ov > 0:
            fl[b] += f * ov / max(1, e - s)
            busy[q][b] += ov / (binms * 1e6)
pan = [0.0] * nb
for s, e, name, q in others:
    if 'panel' in name:
        for b in range(nb):
            lo, hi = t0 + b * binms * 1e6, t0 + (b + 1) * binms * 1e6
            pan[b] += max(0.0, min(e, hi) - max(s, lo)) / (binms * 1e6)
qs = sorted(busy)
print('# evaluation span %.2f ms; bins of %.1f ms: TFLOP/s of the products | busy fraction per '
      'queue %s | panel kernel' % ((t1 - t0) / 1e6, binms, ' '.join(qs)))
tot = 0.0
for b in range(nb):
    tot += fl[b]
    print('%6.1f ms %6.1f TF | %s | %.2f' % (b * binms, fl[b] / (binms * 1e-3) * 1e-12,
                                         ' '.join('%.2f' % busy[q][b] for q in qs), pan[b]))
print('# total %.3e flop in %.2f ms = %.1f TF' % (tot, (t1 - t0) / 1e6, tot / (t1 - t0) * 1e-3))
