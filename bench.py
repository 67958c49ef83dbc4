#!/usr/bin/env python3
"""
Headline benchmark: ExactGP log-likelihood + gradient evaluations per second
at N=16384, D=8, SE-ARD, fp64 (BASELINE.json `metric`), on N GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = every rank evaluates a block of --per-gpu (default 6) independent
hyperparameter vectors of the same dataset through the C-ABI batch entry
gpx_loglik_batch -- the batched-theta path of BASELINE.json's north_star (what
pygp's particle / sample loops do one at a time, meta/smc.py:113-126,
meta/mcmc.py:75-77). One evaluation = what optimize()'s objective does
(/root/reference/pygp/learning/optimization.py:54-59): set_hyper(theta) -> K
build + Cholesky + a, then loglikelihood(True) -> K^-1, alpha and the D+2 trace
terms. Every theta is new, nothing is cached; X and y are already resident in
HBM (uploaded once before the timed region, as GP.add_data does). The library
runs the block as ONE group of members in lock-step (pygp_amd/csrc/group.hip,
round 4): every kernel of the evaluation is one launch over all six members --
the tile engine's batch dimension, the panel kernel with the members' task
graphs interleaved -- so that a product launch is six times as long as a single
evaluation's while its ramp and tail stay the same (rounds 1-3: one context and
stream per member, three in flight; `config.batch_arrangement` says which ran).
With N > 1 GPUs the ranks take disjoint theta blocks (weak
scaling, no data-path collective) and the log-likelihood vector is assembled
with ONE all-gather over RCCL inside the timed region.

`sequential` in the JSON line is the same workload run strictly one evaluation
at a time on one stream (the optimize() objective pattern, where evaluation i+1
depends on i); its HIP-event stage times feed `roofline`.

Rank 0 prints ONE JSON line. Besides the driver contract it carries
  roofline      achieved fp64 TFLOP/s of the dense engine (algorithmic N^3 flop
                per evaluation / HIP-event time of the potrf+trtri+lauum stages
                on the library's stream) against the MI355X fp64 MFMA peak
  cpu_baseline  the NumPy/SciPy oracle (a port that keeps the reference's call
                sequence) timed on this box's host cores: ONE full evaluation at N
                itself in a child process with a time limit (--cpu-budget, 150 s;
                about 45 s on an MI355X box's 16 cores), N/2 or N/4 scaled per stage
                only if that runs out
  configs       C2..C5 of BASELINE.json with a roofline each (tools/bench_configs.py)

With --gpus 1 nothing imports torch (C ABI through ctypes only). `python bench.py
--gpus N` as ONE process drives N GPUs through gpx_loglik_batch_multi (host thread
per device + one ncclAllGather inside libgpx.so); under torch.distributed.run the
ranks are one process per GPU as the driver launches them.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import recipes                                     # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6    # MI355X fp64 matrix peak (vendor sheet); the
                                # probe in tools/probe_mfma.hip sustains 70-73


def host_cores():
    """CPU cores this process may actually use: the smallest of the machine's
    count, the affinity mask and the cgroup CPU quota (a GPU box hands each GPU
    a share of the host, BLAS must not oversubscribe it)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    env = os.environ.get('GPX_CPU_CORES')
    if env:
        n = max(1, int(env))
    return n


def _oracle_eval_timed(orc, sla, N, D):
    """ONE loglik+grad evaluation of the oracle at size N, timed stage by stage in
    the reference's call sequence (exact.py:50-55,118-143)."""
    X, y, _ = recipes.synthetic(N, D)
    theta = recipes.theta_eval(D, 1)
    spec = orc.spec_set_hyper(orc.se_spec(1.0, np.ones(D)), theta[1:-1])
    t = {}
    t0 = time.time()
    K = orc.kernel_get(spec, X) + np.exp(2 * theta[0]) * np.eye(N)
    t['build'] = time.time() - t0
    t0 = time.time()
    R = sla.cholesky(K)
    a = sla.solve_triangular(R, y - theta[-1], trans=True)
    t['cholesky'] = time.time() - t0
    del K
    t0 = time.time()
    alpha = sla.solve_triangular(R, a, trans=False)
    Q = sla.cho_solve((R, False), np.eye(N))
    Q -= np.outer(alpha, alpha)
    t['cho_solve'] = time.time() - t0
    t0 = time.time()
    dl = [-0.5 * np.sum(Q * dK) for dK in orc.kernel_grad(spec, X)]
    t['trace_loop'] = time.time() - t0
    del R, Q, dl
    return t


def _cpu_eval_child(N, D, again_s=60.0):
    """`bench.py --cpu-only N D [again_s]`: oracle evaluations at size N with BLAS limited
    to the cores this process may use -- one, and further ones (at most 3) while less than
    again_s seconds have gone by; prints one JSON line with the median evaluation (of two:
    the faster one, and the record says so)."""
    import platform
    import scipy
    import scipy.linalg as sla
    from oracle import gp_oracle as orc
    cores = host_cores()
    blas = 'unknown'
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
        threadpool_limits(limits=cores)
        blas = '; '.join('%s %s (%s threads)' % (i.get('internal_api'), i.get('version'),
                                                 i.get('num_threads'))
                         for i in threadpool_info() if i.get('user_api') == 'blas')
    except Exception:                                # pragma: no cover
        pass
    cpu_model = platform.processor() or 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu_model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    _oracle_eval_timed(orc, sla, 512, D)             # warm caches / threads
    # up to 3 timed evaluations, as many as start within `again_s`
    runs, t_start = [], time.time()
    while len(runs) < 3 and (not runs or time.time() - t_start < again_s):
        runs.append(_oracle_eval_timed(orc, sla, N, D))
    totals = [sum(r.values()) for r in runs]
    t = runs[int(np.argsort(totals)[(len(runs) - 1) // 2])]     # the median evaluation
    print(json.dumps({'n': N, 'stages': t, 'evals_timed': len(runs), 'eval_seconds': totals,
                      'cores': cores, 'blas': blas, 'cpu_model': cpu_model,
                      'versions': 'numpy %s, scipy %s' % (np.__version__, scipy.__version__)}),
          flush=True)


def cpu_baseline(N, D, budget_s):
    """Oracle (test infrastructure) timed on the host cores of this box: full
    loglik+grad evaluations (1 warm-up at N=512, then 2-3 timed ones: a further one is
    started while fewer than 3 are done and less than budget_s * 0.4 s have gone by; the
    median is reported with n), stage by stage, in a child process with a time limit.
    Size N itself is tried first (about 45 s at N = 16384 on the 16 host cores of an
    MI355X box); if it does not finish within `budget_s` the child is killed and N/2,
    then N/4 are timed instead, and every stage is scaled to N by its own complexity
    (x4 per doubling for the O(N^2 D) stages kernel build and trace loop, x8 for the
    LAPACK stages) -- the record says which it was."""
    import subprocess
    rec = None
    n_timed = N
    tried = []
    for cand in (N, N // 2, N // 4):
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-only',
                                  str(cand), str(D), str(0.4 * budget_s)],
                                 capture_output=True, text=True, timeout=budget_s)
            if out.returncode == 0 and out.stdout.strip():
                rec = json.loads(out.stdout.strip().splitlines()[-1])
                n_timed = cand
                break
            tried.append('N=%d failed (rc %d)' % (cand, out.returncode))
        except subprocess.TimeoutExpired:
            tried.append('N=%d not finished within %.0f s' % (cand, budget_s))
    if rec is None:
        return {'value': None, 'unit': 'evals/s', 'kind': 'port', 'sample': '; '.join(tried)}
    t = rec['stages']
    s_ = float(N) / n_timed
    t_full = s_ ** 2 * (t['build'] + t['trace_loop']) + s_ ** 3 * (t['cholesky'] + t['cho_solve'])
    how = ('measured at N=%d itself' % N if n_timed == N else
           'scaled to N=%d per stage (N^2 stages x%g, N^3 stages x%g; %s)' %
           (N, s_ ** 2, s_ ** 3, '; '.join(tried)))
    return {
        'value': 1.0 / t_full, 'unit': 'evals/s', 'cores': rec['cores'], 'kind': 'port',
        'n_timed': n_timed, 'evals_timed': rec.get('evals_timed', 1),
        'eval_seconds_at_n_timed': rec.get('eval_seconds'), 'seconds_per_eval': t_full,
        'stage_seconds_at_n_timed': t,
        'cpu_model': rec['cpu_model'], 'blas': rec['blas'], 'versions': rec['versions'],
        'sample': 'oracle/gp_oracle.py call sequence (cdist -> cholesky -> cho_solve(eye) '
                  '-> per-hyper sum(Q*dK)), %s loglik+grad evaluations timed at '
                  'N=%d D=%d on %d cores: ' % (
                      {1: 'one', 2: 'the faster of 2'}.get(rec.get('evals_timed', 1),
                                                            'median of n=%d' % rec.get('evals_timed', 1)),
                      n_timed, D, rec['cores']) +
                  ', '.join('%s %.2f s' % kv for kv in t.items()) +
                  '; %s -> %.1f s/eval' % (how, t_full),
    }


def main():
    if len(sys.argv) in (4, 5) and sys.argv[1] == '--cpu-only':
        return _cpu_eval_child(int(sys.argv[2]), int(sys.argv[3]),
                               *[float(a) for a in sys.argv[4:]])
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--size', type=int, default=16384, dest='n')
    ap.add_argument('--dims', type=int, default=8, dest='d')
    ap.add_argument('--per-gpu', type=int, default=6,
                    help='independent thetas per GPU per step (6 since round 3: one group of '
                         'six members in lock-step; rounds 1-2 measured 3 per step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=150.0,
                    help='time limit (s) of the CPU-baseline child (2-3 evaluations: a new '
                         'one starts while 40 %% of it is left); sizes N, N/2, N/4 are tried '
                         'in turn')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the C2..C5 records (tools/bench_configs.py)')
    ap.add_argument('--with-depth3', action='store_true',
                    help='C4s also in a child process with the groups switched off and round '
                         '3\'s pool of CU-masked streams (the comparison recorded in '
                         'profiles/r04_batch_small_depth3_baseline.json); off by default: the '
                         'driver\'s run creates no CU-masked stream')
    args = ap.parse_args()

    # Launch modes:
    #   torchrun (WORLD_SIZE > 1): one process per GPU, torch.distributed over RCCL
    #   one process, --gpus 1:     no torch at all, C ABI only
    #   one process, --gpus N > 1: the in-library multi-device entry
    #                              gpx_loglik_batch_multi (one host thread per GPU,
    #                              one ncclAllGather), still no torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = world > 1 or bool(os.environ.get('GPX_BENCH_FORCE_DIST'))
    in_lib = 1 if use_dist else max(1, args.gpus)      # devices driven by this process
    dist = torch = comm_dev = None
    device = 0
    if use_dist:
        import torch
        import torch.distributed as dist
        ndev = max(1, torch.cuda.device_count())
        device = local_rank % ndev
        torch.cuda.set_device(device)
        # nccl (= RCCL over xGMI) is the backend of the real run; GPX_BENCH_BACKEND=gloo
        # lets the N > 1 path be rehearsed with several ranks on a one-GPU box
        backend = os.environ.get('GPX_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(backend)
        comm_dev = torch.device('cuda', device) if backend == 'nccl' else torch.device('cpu')

    import pygp_amd
    from pygp_amd import _lib

    N, D = args.n, args.d
    X, y, _ = recipes.synthetic(N, D)
    dev = _lib.Handle(device)
    dev.set_data(X, y)                              # resident before timing
    kern = pygp_amd.kernels.SE(1.0, np.ones(D))
    spec = kern._kspec()
    per = max(1, args.per_gpu)
    n_gpus = world if use_dist else in_lib
    if in_lib > 1:                                  # replicate X, y on every device once
        _lib.loglik_batch_multi(spec, np.array([recipes.theta_eval(D, 10 ** 7)]), X, y,
                                grad=True, ndev=in_lib)

    def theta_block(step):
        if in_lib > 1:                              # this process feeds every device
            base = step * in_lib * per
            return np.array([recipes.theta_eval(D, base + j) for j in range(in_lib * per)])
        base = (step * world + rank) * per
        return np.array([recipes.theta_eval(D, base + j) for j in range(per)])

    def evaluate_block(step):
        if in_lib > 1:
            return _lib.loglik_batch_multi(spec, theta_block(step), grad=True, ndev=in_lib)
        return dev.loglik_batch(spec, theta_block(step), grad=True)

    def evaluate_one(i):
        th = recipes.theta_eval(D, i)
        k = kern.copy(th[1:-1])
        return dev.exact_eval(k._kspec(), th[0], th[-1], True)

    def sync():
        dev.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    for w in range(args.warmup):
        evaluate_block(10 ** 6 + w)

    lZ_local = np.empty((args.steps, per * in_lib))
    dlZ_first = None                            # gradient of member 0 of step 0 (parity_check)
    dev.enable_timing(True)                     # HIP events around the groups' dense stages
    if in_lib > 1:                              # ... and on the library's per-device handles
        _lib.multi_enable_timing(in_lib, True)
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        lZ_local[s], dlZ_s = evaluate_block(s)
        if s == 0:
            dlZ_first = np.array(dlZ_s[0])
    if use_dist:
        # the single collective of the batched-theta path: gather lZ
        send = torch.from_numpy(lZ_local.ravel()).to(comm_dev)
        slots = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(slots, send)
        lZ_all = torch.stack(slots).cpu().numpy()
    else:
        lZ_all = lZ_local                            # (in_lib > 1: gathered inside the call)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    batch_dense_ms, batch_members = dev.batch_timings()
    dev.enable_timing(False)
    per_device = None
    if in_lib > 1:
        # the handles of the in-library path live inside libgpx.so: their groups' event times,
        # how each device cut its block and whether its handle is in safe mode
        per_device = _lib.multi_batch_info(in_lib, per, grad=True)
        _lib.multi_enable_timing(in_lib, False)
        batch_dense_ms = max(r['dense_ms'] for r in per_device)     # the slowest device
        batch_members = min(r['members'] for r in per_device)       # (equal blocks: all alike)
    elif use_dist:
        # one process per GPU: every rank reports its own groups (gathered below)
        mine = torch.tensor([batch_dense_ms, float(batch_members),
                             float(dev.safe_mode)], dtype=torch.float64, device=comm_dev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                                # outside the timed region
        per_device = [{'rank': r, 'dense_ms': float(t[0]), 'members': int(t[1]),
                       'safe_mode': bool(t[2])} for r, t in enumerate(every)]
        batch_dense_ms = max(r['dense_ms'] for r in per_device)
        batch_members = min(r['members'] for r in per_device)

    # one-at-a-time evaluations on rank 0: HIP-event stage times for the roofline
    stage, seq_n, seq_s = {}, 0, 0.0
    if rank == 0:
        dev.enable_timing(True)
        evaluate_one(2 * 10 ** 6)
        seq_n = max(3, min(args.steps, 6))
        t1 = time.perf_counter()
        for i in range(seq_n):
            evaluate_one(3 * 10 ** 6 + i)
            for k_, v in dev.timings().items():
                stage[k_] = stage.get(k_, 0.0) + v
        seq_s = time.perf_counter() - t1
        dev.enable_timing(False)

    big_launch = None
    if rank == 0 and N == 16384:
        n_, k_ = N - 1024 - 2048, 2048
        ms_ = dev.la_gemm_bench_mnk(n_, n_, k_, ta=1, tb=0, flags=1, beta=1.0, reps=3)
        big_launch = {'kernel': 'gemm_f64_kernel<1,0,Geo<128,2,4,true>> (+ its 64-tile remainder '
                                'launch), upper tiles, n=%d k=%d, beta=1' % (n_, k_),
                      'flop': float(n_) * n_ * k_, 'ms': ms_,
                      'achieved': float(n_) * n_ * k_ / ms_ * 1e-9}
    if rank == 0:
        evals = args.steps * n_gpus * per
        dense_ms = (stage.get('potrf', 0.0) + stage.get('trtri', 0.0) +
                    stage.get('lauum', 0.0)) / seq_n
        traffic = traffic_src = None   # HBM bytes per evaluation from the committed PMC
        try:                           # passes (tools/collect_profile.sh -> profiles/)
            with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
                tj = json.load(f)
            if (tj.get('n', 16384), tj.get('d', 8)) == (N, D):   # measured at that size only
                traffic = tj['hbm_bytes_per_eval']
                traffic_src = ('profiles/traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / '
                               'WRITE_SIZE passes over one evaluation, FETCH_SIZE doubled '
                               '(gfx950); counters cannot be read from inside this run'
                               % tj.get('tag'))
        except (OSError, KeyError, ValueError):
            pass
        flops = float(N) ** 3                      # potrf N^3/3 + potri 2N^3/3
        seq_achieved = flops / (dense_ms * 1e-3) * 1e-12
        # The timed region itself: this rank's blocks ran as groups in lock-step on one
        # stream each, and HIP events on that stream bracket every group's factor + inverse
        # stage (all product launches and the panel launches between them; the build, the
        # vector and the trace kernels are outside). Without groups (rounds 1-3's contexts,
        # or the in-library multi-device path whose handles live inside the library) the
        # figure of the sequential evaluations stands in.
        if batch_members > 0 and batch_dense_ms > 0:
            achieved = flops * batch_members / (batch_dense_ms * 1e-3) * 1e-12
            measured_on = ('the timed (batched) region: HIP events on the stream of each group '
                           'of %d members in lock-step, around its factor + inverse stage '
                           '(gpx_batch_timings: %.1f ms for %d evaluations%s)'
                           % (per, batch_dense_ms, batch_members,
                              '' if n_gpus == 1 else '; per GPU, the slowest of %d: see '
                              'roofline.per_device' % n_gpus))
        else:
            achieved = seq_achieved
            measured_on = ('sequential evaluations, HIP events on the library stream '
                           'around the potrf (+trtri+lauum) stage: the look-ahead streams '
                           'are joined before the stage ends')
        mode = ('torch.distributed, one process per GPU' if use_dist else
                'one process, gpx_loglik_batch_multi over %d devices' % in_lib if in_lib > 1
                else 'one process, C ABI only (no torch)')
        out = {
            'metric': 'ExactGP log-lik+grad evals/sec at N=%d D=%d SE-ARD fp64' % (N, D),
            'value': evals / elapsed,
            'unit': 'evals/s',
            'n_gpus': n_gpus,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': 'ExactGP SE-ARD fp64 loglik+grad, N=%d D=%d (metric config); '
                            'batched-theta path: %d independent thetas per GPU per step '
                            'through gpx_loglik_batch, X/y resident in HBM' % (N, D, per),
                'evals_per_step': n_gpus * per,
                'thetas_per_gpu_per_step': per,
                # how the library cuts a block of `per` thetas on this device (gpx_batch_plan)
                # (its 'safe_mode' is the arithmetic the handle is on: False = the task-queue
                # launches of the default path; True would mean recursion, another order of
                # arithmetic -- never switched silently, see pygp_amd/_lib.py)
                'batch_arrangement': dev.batch_plan(per, grad=True),
                'safe_mode': dev.safe_mode,
                'safe_mode_switches': dev.safe_mode_switches,
                'parallelism': 'independent thetas sharded over %d GPU(s), no data-path '
                               'collective, one all-gather of lZ' % n_gpus,
                'launch': mode,
            },
            'roofline': {
                'bound': 'mfma',
                'kernel': 'gemm_f64_kernel (all launches of one evaluation: rank-NB trailing '
                          'updates, row panels, inverse columns, K^-1 accumulation) + the '
                          'diagonal-block kernels hidden under them',
                'measured_on': measured_on,
                'achieved': achieved,
                'peak': PEAK_FP64_MFMA_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': achieved / PEAK_FP64_MFMA_TFLOPS,
                # the same algorithmic flops against the wall clock of the TIMED (batched)
                # region: everything included -- builds, trace passes, host syncs
                'achieved_batched_wall': flops * evals / elapsed * 1e-12 / n_gpus,
                'frac_batched_wall': flops * evals / elapsed * 1e-12 / n_gpus / PEAK_FP64_MFMA_TFLOPS,
                'traffic': traffic,
                'traffic_source': traffic_src,
                'algorithmic_flop_per_eval': flops,
                'dense_ms_per_eval': (batch_dense_ms / batch_members
                                      if batch_members > 0 else dense_ms),
                # the figure rounds 1-3 reported as roofline.achieved / frac: one evaluation
                # at a time (the optimize() pattern), HIP events around its potrf(+trtri+
                # lauum) stage with the look-ahead streams joined
                'sequential': {'achieved': seq_achieved,
                               'frac': seq_achieved / PEAK_FP64_MFMA_TFLOPS,
                               'dense_ms_per_eval': dense_ms},
                # one launch of the dominant kernel in isolation (HIP events around 3
                # repetitions of the same launch, operands warm): the largest rank-2048
                # trailing update of the factorisation, C (13312 x 13312, upper tiles) -=
                # P^T P with P 2048 x 13312
                'largest_launch': big_launch,
            },
            'sequential': {
                'evals_per_s': seq_n / seq_s if seq_s > 0 else None,
                'ms_per_eval': seq_s / seq_n * 1e3 if seq_n else None,
                'tflops_N3_wall': flops / (seq_s / seq_n) * 1e-12 if seq_n else None,
                'stage_ms_per_eval': dict((k_, v / seq_n) for k_, v in stage.items()
                                          if v > 0),
            },
            'lZ_first': float(np.ravel(lZ_all)[0]),
        }
        if per_device is not None:
            # every device's / rank's own fraction of the fp64 MFMA peak over ITS groups
            for r in per_device:
                r['achieved'] = (flops * r['members'] / (r['dense_ms'] * 1e-3) * 1e-12
                                 if r['dense_ms'] > 0 else None)
                r['frac'] = r['achieved'] / PEAK_FP64_MFMA_TFLOPS if r['achieved'] else None
            out['roofline']['per_device'] = per_device
            out['config']['safe_mode'] = any(r['safe_mode'] for r in per_device)
        # Parity of the timed call itself: member 0 of step 0 is theta_eval(D, 0), the theta1
        # of the reference-generated fixture tests/golden/g_metric.npz (N = 16384, D = 8);
        # lZ and every gradient component of what the timed region returned against it
        # (tolerance of the tests: 1e-8 relative). The fixture is data, read from the repo.
        try:
            if (N, D) == (16384, 8) and rank == 0 and dlZ_first is not None:
                g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g_metric.npz'))
                if np.array_equal(g['theta1'], recipes.theta_eval(D, 0)):
                    lz0 = float(np.ravel(lZ_local)[0])
                    out['parity_check'] = {
                        'against': 'tests/golden/g_metric.npz lZ1 / dlZ1 (reference-generated)',
                        'member': 'step 0, member 0 of the timed region (theta_eval(8, 0))',
                        'lZ_rel_err': abs(lz0 - float(g['lZ1'])) / abs(float(g['lZ1'])),
                        'dlZ_max_rel_err': float(np.max(np.abs(dlZ_first - g['dlZ1']) /
                                                        np.abs(g['dlZ1']))),
                        'tolerance': 1e-8}
        except (OSError, KeyError, ValueError):
            pass
        # audit of the N > 1 paths: ranks the collective saw, and what every device /
        # rank contributed (sum of its lZ values over all steps)
        if use_dist:
            out['config']['collective_ranks'] = int(dist.get_world_size())
            out['config']['backend'] = os.environ.get('GPX_BENCH_BACKEND', 'nccl')
            out['config']['lZ_sum_per_rank'] = [float(v) for v in lZ_all.sum(axis=1)]
        elif in_lib > 1:
            out['config']['collective_ranks'] = int(_lib.multi_comm_size())
            sums = []
            for dv in range(in_lib):
                lo, hi = _lib.batch_partition(per * in_lib, in_lib, dv)
                sums.append(float(lZ_all[:, lo:hi].sum()))
            out['config']['lZ_sum_per_device'] = sums
        if n_gpus == 1 and not args.no_configs:
            # the other BASELINE configs, each with its own roofline (C2..C5)
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import bench_configs
            out['configs'] = bench_configs.run_all(dev, with_depth3=args.with_depth3)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(N, D, args.cpu_budget)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
