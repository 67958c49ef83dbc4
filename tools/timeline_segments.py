#!/usr/bin/env python3
"""Busy segments of a tools/trace_timeline.py listing (gaps > 200 us between launches split
them): per segment the span and the time per kernel -- one batch of a tools/batch_small.py
run is one segment. usage: python tools/timeline_segments.py <timeline.txt> [min_span_us]"""
import collections
import re
import sys

pat = re.compile(r'\s*(\d+) q(\S+)\s+([\d.]+) us\s+\+\s*([\d.]+) us\s+wg\s+(\d+)\s+(.*)')
ev = []
for line in open(sys.argv[1]):
    m = pat.match(line)
    if m:
        ev.append((float(m.group(3)), float(m.group(4)), m.group(2), int(m.group(5)), m.group(6).strip()))
ev.sort()
minspan = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
segs, cur, end = [], [ev[0]], ev[0][0] + ev[0][1]
for e in ev[1:]:
    if e[0] - end > 200:
        segs.append(cur)
        cur = []
    cur.append(e)
    end = max(end, e[0] + e[1])
segs.append(cur)
for s in segs:
    span = max(e[0] + e[1] for e in s) - s[0][0]
    if span < minspan:
        continue
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for e in s:
        k = re.sub(r'<.*', '', e[4])
        if 'gemm' in e[4]:
            k = 'gemm G%s ta/tb %s' % (re.sub(r'.*G<(\d+).*', r'\1', e[4]),
                                       re.sub(r'.*kernel<(\d), (\d).*', r'\1\2', e[4]))
        tot[k] += e[1]
        cnt[k] += 1
    print('segment at %.0f us: span %.0f us, %d launches on queues %s, sum of durations %.0f us'
          % (s[0][0], span, len(s), sorted({e[2] for e in s}), sum(tot.values())))
    for k, v in sorted(tot.items(), key=lambda x: -x[1])[:14]:
        print('   %-36s n=%4d  %9.0f us  avg %7.1f' % (k, cnt[k], v, v / cnt[k]))
