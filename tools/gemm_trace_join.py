#!/usr/bin/env python3
"""Join the launch log of the tile engine (GPX_GEMM_LOG=file) with a rocprofv3 kernel
trace of the same run: per launch its duration and TFLOP/s. Both are in program order
per stream / HW queue; the k-th big gemm launch of the log on a stream is the k-th
gemm kernel of the matching queue. usage: gemm_trace_join.py <log> <kernel_trace.csv> [min_us]"""
import csv, sys, collections
UPPER, KLO_M, KHI_M, KLO_N, KHI_N = 1, 2, 4, 8, 16
def live_flops(ta, tb, tile, M, N, K, flags, kshift, part, wgs):
    # count k-extent over live tiles of size `tile`
    tot = 0
    for m0 in range(0, M, tile):
        for n0 in range(0, N, tile):
            if flags & UPPER and n0 + tile <= m0: continue
            klo, khi = 0, K
            if flags & KLO_M: klo = max(klo, m0 - kshift)
            if flags & KHI_M: khi = min(khi, m0 + tile)
            if flags & KLO_N: klo = max(klo, n0 - kshift)
            if flags & KHI_N: khi = min(khi, n0 + tile)
            tot += max(0, khi - klo)
    return 2.0 * tile * tile * tot
log = [l.split() for l in open(sys.argv[1])]
rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
minus = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
gem = [r for r in rows if 'gemm_f64_kernel' in r['Kernel_Name']]
# per queue lists vs per stream lists: match by sequence of wg counts
byq = collections.defaultdict(list)
for r in gem: byq[r['Queue_Id']].append(r)
bys = collections.defaultdict(list)
for l in log: bys[l[0]].append(l)
out = []
for sid, ls in bys.items():
    seq = [int(l[10]) for l in ls]
    for q, rs in byq.items():
        qseq = [int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) //
                (int(r['Workgroup_Size_X'])) for r in rs]
        if qseq == seq:
            for l, r in zip(ls, rs):
                out.append((int(r['Start_Timestamp']), q, l, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
            break
    else:
        print('# stream', sid, 'with', len(ls), 'launches: no queue matches')
out.sort()
T0 = out[0][0] if out else 0
agg = collections.defaultdict(lambda: [0.0, 0.0])
for t, q, l, us in out:
    ta, tb, tile, M, N, K, flags, kshift, part, wgs = [int(x) for x in l[1:11]]
    if part == 0:
        fl = live_flops(ta, tb, tile, M, N, K, flags, kshift, part, wgs)
    else:
        fl = 2.0 * tile * tile * K * wgs
    key = 'ta%d tb%d flags%2d K%-5d %s' % (ta, tb, flags, K if not (flags & 30) else -1, 'big' if us > 1000 else 'mid' if us > 150 else 'small')
    agg[key][0] += fl; agg[key][1] += us
    if us >= minus:
        print('%9.2f ms q=%s ta%d tb%d tile%3d M=%5d N=%5d K=%5d fl=%2d ks=%5d part%d wg=%5d %8.1f us %6.1f TF' % (
            (t - T0) / 1e6, q, ta, tb, tile, M, N, K, flags, kshift, part, wgs, us, fl / us * 1e-6))
print()
tf = tu = 0
for k, (fl, us) in sorted(agg.items()):
    print('%-40s %8.2f ms %7.3e flop %6.1f TF' % (k, us / 1e3, fl, fl / us * 1e-6)); tf += fl; tu += us
print('total gemm %.2f ms, %.3e flop, %.1f TF' % (tu / 1e3, tf, tf / tu * 1e-6))
