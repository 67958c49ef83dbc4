#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs of tools/collect_profile.sh (under
gpurun_out/<tag>/) into the committed summaries under profiles/:
  <tag>_bench_n1.json                       the bench line of that box
  <tag>_rocprofv3_kernel_stats*.csv         rocprofv3 --stats tables
  <tag>_trace_summary.txt                   batched run, per kernel and grid
  <tag>_trace_sequential_summary.txt        six evaluations one at a time
  <tag>_gemm_launches_sequential.txt        every tile-engine launch of the sequential
                                            trace with its shape, duration and TFLOP/s
  <tag>_pmc_summary.txt                     FETCH_SIZE / WRITE_SIZE / MFMA passes over
                                            one evaluation, MFMA utilisation per kernel
  <tag>_pmc_hbm_kernels.txt                 counter passes over the HBM-bound kernels
  <tag>_timeline_sequential.txt             flop rate of the products over one evaluation
  <tag>_configs.json                        C2..C5 records (tools/bench_configs.py)
  <tag>_panel_trace_summary.txt             per-task trace of the diagonal-panel kernel
                                            (GPX_PANEL_DEBUG=2): spine phases, task times
  traffic.json                              HBM bytes per evaluation (gfx950 FETCH_SIZE
                                            correction applied), read by bench.py
usage: make_profile_summaries.py <tag> <evals in the batched trace>"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, nev = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
one = lambda pat: max(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)  # newest run
run = lambda *a: subprocess.run([sys.executable] + list(a), capture_output=True, text=True).stdout
T = os.path.join(ROOT, 'tools')

shutil.copy(one('trace/*/*kernel_stats.csv'), os.path.join(dst, tag + '_rocprofv3_kernel_stats.csv'))
shutil.copy(os.path.join(src, 'bench_n1.json'), os.path.join(dst, tag + '_bench_n1.json'))
shutil.copy(os.path.join(src, 'configs.json'), os.path.join(dst, tag + '_configs.json'))
with open(os.path.join(dst, tag + '_trace_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 '
            '--no-cpu-baseline --no-configs; figures per evaluation (%s evaluations in the '
            'trace: (1+5) steps x 6 thetas, 3 in flight, + 7 sequential). Kernels of the 3 '
            'concurrent evaluations overlap, so their durations here are longer than in the '
            'sequential trace.\n' % nev)
    f.write(run(os.path.join(T, 'trace_summary.py'), one('trace/*/*kernel_trace.csv'), nev))
seq = one('trace_seq/*/*kernel_trace.csv')
shutil.copy(one('trace_seq/*/*kernel_stats.csv'),
            os.path.join(dst, tag + '_rocprofv3_kernel_stats_sequential.csv'))
with open(os.path.join(dst, tag + '_trace_sequential_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 tools/run_eval.py 16384 6: six '
            'sequential loglik+grad evaluations at N=16384 D=8 (the mode bench.py measures '
            'its roofline section in: look-ahead with diagonal blocks on the high-priority '
            'stream, trailing updates and inverse columns on two more, every stream over '
            'every CU -- no CU masks unless GPX_RESERVE_CUS is set); figures per evaluation. '
            'Kernels of the three streams overlap: per-kernel sums exceed the wall time.\n')
    f.write(run(os.path.join(T, 'trace_summary.py'), seq, '6'))
    f.write('\n# HW queues of the LAST evaluation in that trace (tools/trace_queues.py)\n')
    f.write('\n'.join(run(os.path.join(T, 'trace_queues.py'), seq, '1e9').splitlines()[:8]) + '\n')
with open(os.path.join(dst, tag + '_gemm_launches_sequential.txt'), 'w') as f:
    f.write('# tools/gemm_trace_join.py: launch log of the tile engine (GPX_GEMM_LOG) joined '
            'with the sequential kernel trace: every launch >= 400 us of the six evaluations '
            '(profiled run: ~4% slower than an unprofiled one) and totals per shape class. '
            'flags: 1 upper tiles only, 2/8 k >= row/col tile (- kshift), 4/16 k < row/col '
            'tile + 128; part 1/2: whole rounds of 128-tiles / remainder as 64-tiles.\n')
    f.write(run(os.path.join(T, 'gemm_trace_join.py'), os.path.join(src, 'gemmlog_seq.txt'), seq, '400'))
with open(os.path.join(dst, tag + '_timeline_sequential.txt'), 'w') as f:
    f.write('# tools/timeline.py: flop rate of the products over the LAST evaluation of the '
            'sequential trace (launch log joined with the kernel trace, every launch spread '
            'evenly over its duration), 2-ms bins, with the busy fraction of each hardware '
            'queue and of the panel kernel\n')
    f.write(run(os.path.join(T, 'timeline.py'), os.path.join(src, 'gemmlog_seq.txt'), seq, '2', '1'))
tot = {}
with open(os.path.join(dst, tag + '_pmc_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --pmc <group> -- python3 tools/run_eval.py 16384 1 (ONE evaluation, '
            'N=16384 D=8); one pass per counter group; kernels with grid >= 1e6 threads. '
            'MFMA utilisation is per SIMD of the whole GPU: the products run on every CU by '
            'default (with GPX_RESERVE_CUS=32 on 224 of them, where 0.78 here is 0.89 of the '
            'pipes they may use).\n')
    for d in ('pmc_fetch', 'pmc_write', 'pmc_mfma'):
        path = one(d + '/*/*counter_collection.csv')
        f.write('## %s\n' % d)
        f.write(run(os.path.join(T, 'pmc_summary.py'), path, '1000000'))
        for r in csv.DictReader(open(path)):
            tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
with open(os.path.join(dst, tag + '_pmc_hbm_kernels.txt'), 'w') as f:
    f.write('# tools/pmc_hbm.sh: counter passes over the HBM-bound kernels on their BASELINE '
            'sizes (tools/run_hbm.py: kbuild fp32 SE+Periodic and SE, N=32768 D=4 -- config 5; '
            'kbuild fp64 and trace_grad inside an N=16384 D=8 evaluation), one group per run\n')
    for i in range(1, 6):
        path = one('hbm/pmc%d/*/*counter_collection.csv' % i)
        f.write('## pass %d\n' % i)
        f.write(run(os.path.join(T, 'pmc_summary.py'), path, '1000000', 'kbuild|trace_grad'))
    f.write('## kernel trace of the same program (3 repetitions)\n')
    f.write(run(os.path.join(T, 'trace_summary.py'), one('hbm/trace/*/*kernel_trace.csv'), '1'))
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests
# as 64 B (MI355X_MICROARCH.md, HBM section) -> double it
traffic = (2.0 * tot.get('FETCH_SIZE', 0.0) + tot.get('WRITE_SIZE', 0.0)) * 1024.0
json.dump({'tag': tag, 'n': 16384, 'd': 8, 'hbm_bytes_per_eval': traffic,
           'fetch_size_kib_raw': tot.get('FETCH_SIZE'), 'write_size_kib': tot.get('WRITE_SIZE'),
           'mfma_busy_cycles': tot.get('SQ_VALU_MFMA_BUSY_CYCLES'),
           'note': 'one N=16384 D=8 loglik+grad evaluation; FETCH_SIZE doubled per the '
                   'gfx950 correction'},
          open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
with open(os.path.join(dst, tag + '_panel_trace_summary.txt'), 'w') as f:
    f.write('# 1024-block, one panel launch (GPX_PANEL_DEBUG=2 GPX_LOOKAHEAD=0 tools/panel_dbg.py 1024 x3;\n'
            '# the first launch is cold). Times in us from the first claim. Spine task = XS phase\n'
            '# (strips-done) + diagonal update (syrk-done) + leaf (pivots-done, R-out) of one tile.\n')
    f.write(run(os.path.join(T, 'panel_trace_summary.py'), os.path.join(src, 'panel_trace.log')))
    f.write('\n# the same block with the round-1 task graph (GPX_PANEL_STREAM=0: leaf, then row panel\n'
            '# and diagonal update as 32x32 product tasks)\n')
    f.write(run(os.path.join(T, 'panel_trace_summary.py'), os.path.join(src, 'panel_trace_r1graph.log')))
print('traffic per eval: %.3e B' % traffic)
