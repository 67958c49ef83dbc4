#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs of tools/collect_profile.sh (under
gpurun_out/<tag>/) into the committed summaries under profiles/:
  <tag>_rocprofv3_kernel_stats.csv, <tag>_trace_summary.txt, <tag>_pmc_summary.txt,
  traffic.json (HBM bytes per evaluation, gfx950 FETCH_SIZE correction applied).
usage: make_profile_summaries.py <tag> <evals in trace run>"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, nev = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
one = lambda pat: max(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)  # newest run
shutil.copy(one('trace/*/*kernel_stats.csv'), os.path.join(dst, tag + '_rocprofv3_kernel_stats.csv'))
if os.path.exists(os.path.join(src, 'bench_n1.json')):
    shutil.copy(os.path.join(src, 'bench_n1.json'), os.path.join(dst, tag + '_bench_n1.json'))
with open(os.path.join(dst, tag + '_trace_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 '
            '--no-cpu-baseline; figures per evaluation (%s evaluations in the trace: '
            '(1+5) steps x 3 thetas in flight + 6 sequential). Kernels of the 3 '
            'concurrent evaluations overlap, so their durations here are longer than '
            'in the sequential trace below, which is what bench.py\'s roofline uses.\n'
            % nev)
    f.write(subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_summary.py'),
                            one('trace/*/*kernel_trace.csv'), nev],
                           capture_output=True, text=True).stdout)
seq = glob.glob(os.path.join(src, 'trace_seq/*/*kernel_trace.csv'))
if seq:
    seq = max(seq, key=os.path.getmtime)
    shutil.copy(max(glob.glob(os.path.join(src, 'trace_seq/*/*kernel_stats.csv')),
                    key=os.path.getmtime),
                os.path.join(dst, tag + '_rocprofv3_kernel_stats_sequential.csv'))
    with open(os.path.join(dst, tag + '_trace_sequential_summary.txt'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats -- python3 tools/run_eval.py 16384 6: '
                '6 sequential loglik+grad evaluations at N=16384 D=8 on one stream (the '
                'mode bench.py measures its roofline section in); figures per evaluation\n')
        f.write(subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_summary.py'),
                                seq, '6'], capture_output=True, text=True).stdout)
tot = {}
with open(os.path.join(dst, tag + '_pmc_summary.txt'), 'w') as f:
    f.write('# rocprofv3 --pmc <group> -- python3 tools/run_eval.py 16384 1 (ONE evaluation, '
            'N=16384 D=8); one pass per counter group\n')
    for d in ('pmc_fetch', 'pmc_write', 'pmc_mfma'):
        path = one(d + '/*/*counter_collection.csv')
        f.write('## %s\n' % d)
        f.write(subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'),
                                path, '1000000'], capture_output=True, text=True).stdout)
        for r in csv.DictReader(open(path)):
            tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests
# as 64 B (MI355X_MICROARCH.md, HBM section) -> double it
traffic = (2.0 * tot.get('FETCH_SIZE', 0.0) + tot.get('WRITE_SIZE', 0.0)) * 1024.0
json.dump({'tag': tag, 'n': 16384, 'd': 8, 'hbm_bytes_per_eval': traffic,
           'fetch_size_kib_raw': tot.get('FETCH_SIZE'), 'write_size_kib': tot.get('WRITE_SIZE'),
           'mfma_busy_cycles': tot.get('SQ_VALU_MFMA_BUSY_CYCLES'),
           'note': 'one N=16384 D=8 loglik+grad evaluation; FETCH_SIZE doubled per the '
                   'gfx950 correction'},
          open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
print('traffic per eval: %.3e B' % traffic)
