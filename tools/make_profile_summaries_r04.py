#!/usr/bin/env python3
"""Round 4: turn the raw rocprofv3 outputs of tools/collect_profile_r04.sh (under
gpurun_out/<tag>/) into the committed summaries under profiles/:
  <tag>_bench_n1.json                       the bench line of that box (with configs, C4s and
                                            the CPU baseline)
  <tag>_configs.json                        the `configs` of that line on their own
  <tag>_rocprofv3_kernel_stats*.csv         rocprofv3 --stats tables (bench / sequential)
  <tag>_trace_summary.txt                   the bench trace: groups of six in lock-step + the
                                            sequential evaluations, per kernel and grid, and
                                            the breakdown of ONE group step
  <tag>_trace_sequential_summary.txt        six evaluations one at a time
  <tag>_gemm_launches_sequential.txt        every tile-engine launch of the sequential trace
  <tag>_timeline_sequential.txt             flop rate of the products over one evaluation
  <tag>_pmc_summary.txt                     FETCH_SIZE / WRITE_SIZE / MFMA passes over ONE
                                            group of six members (the unit of the timed region)
  <tag>_trace_batch_small.txt               256 thetas at N = 512 / 1024 / 2048: kernel stats
  <tag>_panel_trace_summary.txt             per-task trace of the 1024-block panel launch
  traffic.json                              HBM bytes per evaluation (gfx950 FETCH_SIZE
                                            correction applied), read by bench.py
usage: make_profile_summaries_r04.py <tag>"""
import csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
one = lambda pat: max(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
run = lambda *a: subprocess.run([sys.executable] + list(a), capture_output=True, text=True).stdout
T = os.path.join(ROOT, 'tools')

bench = json.load(open(os.path.join(src, 'bench_n1.json')))
json.dump(bench, open(os.path.join(dst, tag + '_bench_n1.json'), 'w'), indent=1)
json.dump(bench.get('configs', []), open(os.path.join(dst, tag + '_configs.json'), 'w'), indent=1)
shutil.copy(one('trace/*/*kernel_stats.csv'), os.path.join(dst, tag + '_rocprofv3_kernel_stats.csv'))

# ---- the bench trace: (1 + 5) steps x 6 thetas as groups of six + 7 sequential evaluations
trace = one('trace/*/*kernel_trace.csv')
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(r):
    return re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')


starts = [i for i, r in enumerate(rows) if short(r).startswith('kbuild_kernel<double, 4, true>')]
with open(os.path.join(dst, tag + '_trace_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 '
            '--no-cpu-baseline --no-configs: (1 + 5) steps of ONE group of six members in '
            'lock-step on one stream (pygp_amd/csrc/group.hip) + 7 sequential evaluations '
            '(look-ahead streams). Figures per evaluation over all 43.\n')
    f.write(run(os.path.join(T, 'trace_summary.py'), trace, '43'))
    if starts:
        i0 = starts[-1]
        end = next((j for j in range(i0, len(rows)) if short(rows[j]).startswith('trace_reduce')),
                   len(rows) - 1)
        seg = rows[i0:end + 1]
        t0 = int(seg[0]['Start_Timestamp'])
        t1 = max(int(r['End_Timestamp']) for r in seg)
        agg = {}
        for r in seg:
            k = short(r)
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        busy = sum(v[1] for v in agg.values())
        f.write('\n# ONE group step (the last of the run: six loglik+grad evaluations at '
                'N = 16384 in lock-step): span %.1f ms, %d launches, sum of kernel times %.1f ms '
                '(one stream: no overlap). Products on 128-tiles: %.1f ms for 6 x 4.474e12 flop '
                '= %.1f TFLOP/s.\n' % (
                    (t1 - t0) / 1e6, len(seg), busy,
                    sum(v[1] for k, v in agg.items() if 'gemm_f64_kernel' in k and 'Geo<128' in k),
                    6 * 4.474e12 / (sum(v[1] for k, v in agg.items()
                                        if 'gemm_f64_kernel' in k and 'Geo<128' in k) * 1e-3) * 1e-12))
        for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('%9.2f ms %4d launches  avg %9.1f us  %s\n' % (d, c, d / c * 1e3, k))

# ---- sequential evaluations (unchanged path: the optimize() pattern)
seq = one('trace_seq/*/*kernel_trace.csv')
shutil.copy(one('trace_seq/*/*kernel_stats.csv'),
            os.path.join(dst, tag + '_rocprofv3_kernel_stats_sequential.csv'))
with open(os.path.join(dst, tag + '_trace_sequential_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 tools/run_eval.py 16384 6: six '
            'sequential loglik+grad evaluations at N=16384 D=8 (look-ahead with diagonal blocks '
            'on the high-priority stream, trailing updates and inverse columns on two more, '
            'every stream over every CU); figures per evaluation. Kernels of the three streams '
            'overlap: per-kernel sums exceed the wall time.\n')
    f.write(run(os.path.join(T, 'trace_summary.py'), seq, '6'))
    f.write('\n# HW queues of the LAST evaluation in that trace (tools/trace_queues.py)\n')
    f.write('\n'.join(run(os.path.join(T, 'trace_queues.py'), seq, '1e9').splitlines()[:8]) + '\n')
with open(os.path.join(dst, tag + '_gemm_launches_sequential.txt'), 'w') as f:
    f.write('# tools/gemm_trace_join.py: launch log of the tile engine (GPX_GEMM_LOG) joined '
            'with the sequential kernel trace: every launch >= 400 us of the six evaluations '
            'and totals per shape class.\n')
    f.write(run(os.path.join(T, 'gemm_trace_join.py'), os.path.join(src, 'gemmlog_seq.txt'), seq, '400'))
with open(os.path.join(dst, tag + '_timeline_sequential.txt'), 'w') as f:
    f.write('# tools/timeline.py: flop rate of the products over the LAST evaluation of the '
            'sequential trace, 2-ms bins, with the busy fraction of each hardware queue and of '
            'the panel kernel\n')
    f.write(run(os.path.join(T, 'timeline.py'), os.path.join(src, 'gemmlog_seq.txt'), seq, '2', '1'))

# ---- counters over ONE group of six
tot = {}
with open(os.path.join(dst, tag + '_pmc_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --pmc <group> -- python3 tools/run_batch.py 16384 6 1: ONE group of six '
            'loglik+grad evaluations at N=16384 D=8 in lock-step (the unit of work of the timed '
            'region of bench.py); one pass per counter group; kernels with grid >= 1e6 threads. '
            'Per-dispatch figures are per launch over all six members.\n')
    for d in ('pmc_fetch', 'pmc_write', 'pmc_mfma'):
        path = one(d + '/*/*counter_collection.csv')
        f.write('## %s\n' % d)
        f.write(run(os.path.join(T, 'pmc_summary.py'), path, '1000000'))
        for r in csv.DictReader(open(path)):
            tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests
# as 64 B (MI355X_MICROARCH.md, HBM section) -> double it
members = 6
traffic = (2.0 * tot.get('FETCH_SIZE', 0.0) + tot.get('WRITE_SIZE', 0.0)) * 1024.0 / members
json.dump({'tag': tag, 'n': 16384, 'd': 8, 'hbm_bytes_per_eval': traffic,
           'fetch_size_kib_raw': tot.get('FETCH_SIZE'), 'write_size_kib': tot.get('WRITE_SIZE'),
           'mfma_busy_cycles': tot.get('SQ_VALU_MFMA_BUSY_CYCLES'),
           'members_in_the_pass': members,
           'note': 'ONE group of six N=16384 D=8 loglik+grad evaluations in lock-step '
                   '(tools/run_batch.py 16384 6 1), totals divided by six; FETCH_SIZE doubled '
                   'per the gfx950 correction'},
          open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)

# ---- 256 thetas at small sizes
with open(os.path.join(dst, tag + '_trace_batch_small.txt'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 tools/batch_small.py --b 256 --sizes N '
            '--reps 1: warm-up batches of 8, one value-only and one with-gradients batch of 256 '
            'thetas (two groups of 128 in flight; lock-step sweep: since round 5 sweep_xs_kernel = the '
            'dense row panels, sweep_kernel = last diagonal update + leaf, two launches a tile row) '
            'and ten single evaluations. Per kernel: calls, total, average. Launch by launch: '
            'profiles/r05_small_groups.txt.\n')
    for n in (512, 1024, 2048):
        f.write('\n## N = %d\n' % n)
        for r in list(csv.DictReader(open(one('small_%d/*/*kernel_stats.csv' % n))))[:14]:
            f.write('%-70s calls %5s total %9.3f ms avg %9.1f us  %5s %%\n' % (
                re.sub(r'\(.*', '', r['Name']).replace('void ', '')[:70], r['Calls'],
                float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage'][:5]))
        log = open(os.path.join(src, 'small_%d.log' % n)).read().strip().splitlines()
        for l in log:
            if l.startswith('{'):
                q = json.loads(l)
                f.write('(under the profiler: value-only %.0f evals/s, with gradients %.0f)\n'
                        % (q['value_only_evals_per_s'], q['with_grad_evals_per_s']))

with open(os.path.join(dst, tag + '_panel_trace_summary.txt'), 'w') as f:
    f.write('# 1024-block, one panel launch (GPX_PANEL_DEBUG=2 GPX_LOOKAHEAD=0 tools/panel_dbg.py 1024 x3;\n'
            '# the first launch is cold). Times in us from the first claim. Spine task = XS phase\n'
            '# (strips-done) + diagonal update (syrk-done) + leaf (pivots-done, R-out) of one tile.\n'
            '# (The whole-matrix launch at N = 4096: r04_panel_whole_4096_trace_summary.txt.)\n')
    f.write(run(os.path.join(T, 'panel_trace_summary.py'), os.path.join(src, 'panel_trace.log')))
# round 5: the whole-matrix launches traced task by task, and what their workers do
def have(name):
    return os.path.exists(os.path.join(src, name))
PTS, PBUSY = os.path.join(T, 'panel_trace_summary.py'), os.path.join(T, 'panel_busy.py')
if have('panel_whole_4096.log'):
    with open(os.path.join(dst, tag + '_panel_whole_4096_trace_summary.txt'), 'w') as f:
        f.write('# GPX_PANEL_DEBUG=2 python3 tools/run_value.py 4096 3, last launch: the ONE-launch factorisation of\n'
                '# C2\'s update (value-only), task by task; spine = followers (op 4: rows-in, update-done,\n'
                '# pivots-done, R-out) and the solves on the spine\'s workgroups (op 3); below it what the worker\n'
                '# pool does in 100-us bins (tools/panel_busy.py)\n')
        f.write(run(PTS, os.path.join(src, 'panel_whole_4096.log'), '--last'))
        f.write('\n# workers\n' + run(PBUSY, os.path.join(src, 'panel_whole_4096.log')))
if have('panel_whole_2048.log'):
    with open(os.path.join(dst, tag + '_panel_whole_2048_traces.txt'), 'w') as f:
        f.write('# GPX_PANEL_DEBUG=2 python3 tools/run_value.py 2048 3 (last launch): the whole-matrix launch at N = 2048, value-only\n'
                '# (a) the committed kernel (followers + folding solves, hand-offs polled as data)\n')
        f.write(run(PTS, os.path.join(src, 'panel_whole_2048.log'), '--last'))
        if have('panel_whole_2048_fused.log'):
            f.write('\n# (b) GPX_PANEL_SPLIT=0: the round-2..4 graph (fused spine task, every update a product task) on the same box\n')
            f.write(run(PTS, os.path.join(src, 'panel_whole_2048_fused.log'), '--last'))
if have('panel_whole_2048_grad.log'):
    with open(os.path.join(dst, tag + '_panel_whole_grad_traces.txt'), 'w') as f:
        f.write('# evaluations WITH gradients (python3 tools/run_value.py N 3 grad): up to np = 4096 the launch assembles\n'
                '# all of R^-1 beside R (chunked sums, gpx_grad_full_w). tools/panel_busy.py: what the worker pool does\n'
                '# in 100-us bins (factor = tasks of the factorisation, I1 = the sums, I2 = the products with W_ss,\n'
                '# waiting = inside a claimed task whose counters are not there yet). (a) N = 2048\n')
        f.write(run(PBUSY, os.path.join(src, 'panel_whole_2048_grad.log')))
        if have('panel_whole_4096_grad_fullw.log'):
            f.write('\n# (b) N = 4096: the launch is bound by its workers (about 345 ms of task time on 250 of them). The trace\n'
                    '# itself costs here: 1.8-2.3 ms traced against 1.66 ms untraced (profiles/r05_stage_time.txt), and\n'
                    '# the bins show what an in-order queue does then -- workers waiting inside claimed tasks whenever\n'
                    '# the queue order runs ahead of the chain\n')
            f.write(run(PBUSY, os.path.join(src, 'panel_whole_4096_grad_fullw.log')))
            f.write(run(PTS, os.path.join(src, 'panel_whole_4096_grad_fullw.log'), '--last').split('  products')[0][-3000:])
if have('seq_time.txt'):
    shutil.copy(os.path.join(src, 'seq_time.txt'), os.path.join(dst, tag + '_seq_time.txt'))
if have('stage_time.txt'):
    shutil.copy(os.path.join(src, 'stage_time.txt'), os.path.join(dst, tag + '_stage_time.txt'))
print('traffic per eval: %.3e B' % traffic)
