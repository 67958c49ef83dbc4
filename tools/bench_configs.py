#!/usr/bin/env python3
"""Timings of the BASELINE.json configs other than the headline metric on one
MI355X, each with the roofline that bounds it. bench.py attaches
these records to its JSON line under `configs`; run directly it prints one JSON
object per config.
usage: python tools/bench_configs.py [c2 c3 c4 c4s c5] [--with-depth3] [--out FILE]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import recipes                                     # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBPS = 8000.0


def timed(f, reps=3):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))


def mfma_roof(flop, seconds):
    ach = flop / seconds * 1e-12
    return {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP64_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': ach / PEAK_FP64_MFMA_TFLOPS, 'flop': flop,
            'ms': seconds * 1e3}


def hbm_roof(nbytes, seconds):
    ach = nbytes / seconds * 1e-9
    return {'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
            'frac': ach / PEAK_HBM_GBPS, 'bytes': nbytes, 'ms': seconds * 1e3}


def run_c2(dev):
    """ExactGP SE-ARD fp64, N=4096 D=8: build + blocked Cholesky + posterior."""
    import pygp_amd
    N, D, M = 4096, 8, 4096
    X, y, _ = recipes.synthetic(N, D)
    Xs = np.random.RandomState(1).rand(M, D)
    th = recipes.theta_eval(D, 0)
    k = pygp_amd.kernels.SE(1.0, np.ones(D)).copy(th[1:-1])
    dev.set_data(X, y)
    t_up = timed(lambda: dev.exact_update(k._kspec(), th[0], th[-1]), 5)
    t_ev = timed(lambda: dev.exact_eval(k._kspec(), th[0], th[-1], True), 5)
    dev.exact_update(k._kspec(), th[0], th[-1])
    t_po = timed(lambda: dev.exact_posterior(Xs))
    t_p1 = timed(lambda: dev.exact_posterior(Xs[:128]))
    return {'config': 'C2 ExactGP SE-ARD fp64 N=4096 D=8',
            'update_ms': t_up * 1e3, 'loglik_grad_eval_ms': t_ev * 1e3,
            'posterior_4096pts_ms': t_po * 1e3, 'posterior_128pts_ms': t_p1 * 1e3,
            'roofline': {'update (N^3/3 flop)': mfma_roof(float(N) ** 3 / 3, t_up),
                         'loglik+grad eval (N^3 flop)': mfma_roof(float(N) ** 3, t_ev),
                         'posterior 4096 pts (N^2 M flop)': mfma_roof(float(N) * N * M, t_po)}}


def run_c3(dev):
    """Matern-5/2 ARD fp64, N=16384 D=16: log-lik + 19-component gradient."""
    import pygp_amd
    N, D = 16384, 16
    X, y, _ = recipes.synthetic(N, D)
    th = recipes.theta0(D, 2.0)
    k = pygp_amd.kernels.Matern(1.0, np.ones(D), d=5).copy(th[1:-1])
    dev.set_data(X, y)
    dev.enable_timing(True)
    out = {}

    def run():
        out['r'] = dev.exact_eval(k._kspec(), th[0], th[-1], True)
    t = timed(run)
    stage = {a: b for a, b in dev.timings().items() if b > 0 and 'posterior' not in a}
    dev.enable_timing(False)
    return {'config': 'C3 ExactGP Matern-5/2 ARD fp64 N=16384 D=16 loglik+grad',
            'eval_ms': t * 1e3, 'evals_per_s': 1 / t, 'lZ': out['r'][0], 'stage_ms': stage,
            'roofline': {'loglik+grad eval (N^3 flop)': mfma_roof(float(N) ** 3, t)}}


def run_c4(dev):
    """64 thetas x N=8192 D=8 on ONE GPU (the per-rank share of the 8-GPU config is
    B / world)."""
    import pygp_amd
    N, D, B = 8192, 8, 64
    X, y, _ = recipes.synthetic(N, D)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev.set_data(X, y)
    # one untimed pass of the same shape first: the workspaces of the groups (a one-off
    # allocation of tens of GB for the life of the handle) and the tile lists
    dev.loglik_batch(k._kspec(), thetas, grad=True)
    dev.loglik_batch(k._kspec(), thetas, grad=False)
    t0 = time.perf_counter()
    lZ = dev.loglik_batch(k._kspec(), thetas, grad=False)
    t_val = time.perf_counter() - t0
    t0 = time.perf_counter()
    dev.loglik_batch(k._kspec(), thetas, grad=True)
    t_grad = time.perf_counter() - t0
    th = thetas[0]
    kk = k.copy(th[1:-1])
    t_one = timed(lambda: dev.exact_eval(kk._kspec(), th[0], th[-1], True), 5)
    return {'config': 'C4 batched sweep 64 thetas x ExactGP SE-ARD N=8192 D=8, 1 GPU',
            'value_only_s': t_val, 'value_only_evals_per_s': B / t_val,
            'with_grad_s': t_grad, 'with_grad_evals_per_s': B / t_grad,
            'one_eval_with_grad_ms': t_one * 1e3,
            'lZ0': float(lZ[0]), 'lZ1': float(lZ[1]),
            'roofline': {'64 value-only evals (B N^3/3 flop)':
                         mfma_roof(B * float(N) ** 3 / 3, t_val),
                         '64 loglik+grad evals (B N^3 flop)': mfma_roof(B * float(N) ** 3, t_grad),
                         'one loglik+grad eval (N^3 flop)': mfma_roof(float(N) ** 3, t_one)}}


def run_c5(dev):
    """fp32 SE+Periodic build N=32768 D=4: GB/s against the HBM roof (algorithmic
    bytes = N^2 x 4 written once)."""
    from pygp_amd import _lib
    N, D = 32768, 4
    X = np.random.RandomState(0).rand(N, D)
    dev.set_data(X, np.zeros(N))
    hse = _lib.KSpecHolder(_lib.KIND_SE, False, D, np.r_[0.0, np.log(np.linspace(.5, 1.5, D))])
    hper = _lib.KSpecHolder(_lib.KIND_PERIODIC, False, D, np.r_[0.0, 0.0, np.log(0.7)])
    hsum = _lib.KSpecHolder(_lib.KIND_SUM, False, D, parts=[hse, hper])
    # The GPU's clock ramp (round 5, tools/c5_context.py, profiles/r05_c5_context.txt): the
    # first ~10 ms after the device has been idle for a moment -- a host-side pause is enough
    # -- run at a lower shader clock. The SE+Periodic build (sqrt, sin, exp per pair: the one
    # kernel here that is close to ALU-bound at HBM speed) then takes 0.95-0.97 ms, and
    # 0.78-0.80 ms when the GPU was busy right before, whatever it was busy with and whatever
    # else the handle holds; the store-bound SE build is 0.79 ms either way. That was the
    # "regression" of round 4: round 3 timed C5 right behind C4's products, round 4 behind a
    # child process it had waited for. Both figures are recorded: `idle_start` = 10 builds
    # after 0.5 s of idle (what a lone call sees), the headline figure = 10 builds right
    # behind 100 untimed ones (steady state, what a loop over builds sees).
    time.sleep(0.5)
    ms_cold = dev.kernel_build_resident(hsum, np.float32, reps=10)
    dev.kernel_build_resident(hsum, np.float32, reps=100)
    ms = dev.kernel_build_resident(hsum, np.float32, reps=10)
    ms_se = dev.kernel_build_resident(hse, np.float32, reps=10)
    ms64 = dev.kernel_build_resident(hse, np.float64, reps=3)
    return {'config': 'C5 fp32 SE+Periodic kernel build N=32768 D=4 (full square, resident)',
            'se+periodic_fp32_ms': ms, 'se_fp32_ms': ms_se, 'se_fp64_ms': ms64,
            'se+periodic_fp32_idle_start_ms': ms_cold,
            'protocol': 'steady state: 10 builds timed right behind 100 untimed ones; '
                        'idle_start: 10 builds after 0.5 s of idle (clock ramp)',
            'roofline': {'SE+Periodic fp32 (N^2 x 4 B)': hbm_roof(N * N * 4.0, ms * 1e-3),
                         'SE fp32 (N^2 x 4 B)': hbm_roof(N * N * 4.0, ms_se * 1e-3),
                         'SE fp64 (N^2 x 8 B)': hbm_roof(N * N * 8.0, ms64 * 1e-3)}}


def run_c4s(dev, with_depth3=False):
    """The batched-theta path at the sizes its consumers use (the particle / sample loops
    of /root/reference/pygp/meta/smc.py:86-126 and learning/sampling.py:102-124): 256 thetas
    x N = 512, 1024, 2048, value-only and with gradients, each with its N^3/3 or N^3
    roofline. `depth3_same_run`: the same batches in a child process with the member-batched
    groups switched off (GPX_GROUP_MAX_NP=0: rounds 1-3's one stream and launch sequence per
    member, three in flight); profiles/r04_batch_small_depth3_baseline.json holds the same
    figures measured with the round-3 library."""
    import subprocess
    import batch_small
    recs = []
    for N in (512, 1024, 2048):
        r, _, _ = batch_small.run_size(dev, N, 8, 256, reps=7)
        recs.append(r)
    legacy = {}
    if not with_depth3:
        # (default since round 5: the comparison child -- the only thing in a bench run that
        # creates CU-masked streams -- is behind --with-depth3; its figures with the round-3
        # library are in profiles/r04_batch_small_depth3_baseline.json)
        return {'config': 'C4s batched sweep 256 thetas x ExactGP SE-ARD N in {512, 1024, 2048} '
                          'D=8, 1 GPU (the sizes of the reference\'s particle / sample loops)',
                'sizes': recs}
    # The child runs the contexts the way round 3 did (GPX_TWIN_MASKED=1: pool streams with
    # hardware queues of their own -- the library's default is plain streams since destroying
    # masked ones was found to hang now and then; GPX_PANEL_SERIAL=0: the
    # panel launches of the three contexts side by side, not one at a time per device). Its records are read
    # as they are printed and the child is never waited for beyond a kill: a teardown that
    # does not return must not hold up the bench.
    try:
        import threading
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'batch_small.py'),
                              '--b', '256', '--sizes', '512,1024,2048', '--reps', '2'],
                             env=dict(os.environ, GPX_GROUP_MAX_NP='0', GPX_TWIN_MASKED='1',
                                      GPX_PANEL_SERIAL='0'),
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

        def read():
            for line in p.stdout:
                if line.startswith('{'):
                    q = json.loads(line)
                    legacy[q['n']] = q
                    if len(legacy) == 3:
                        return
        t = threading.Thread(target=read, daemon=True)
        t.start()
        t.join(300)
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    except (subprocess.SubprocessError, OSError, ValueError):
        pass
    for r in recs:
        q = legacy.get(r['n'])
        if q:
            r['depth3_same_run'] = {'value_only_evals_per_s': q['value_only_evals_per_s'],
                                    'with_grad_evals_per_s': q['with_grad_evals_per_s']}
            r['speedup_vs_depth3'] = {
                'value_only': r['value_only_evals_per_s'] / q['value_only_evals_per_s'],
                'with_grad': r['with_grad_evals_per_s'] / q['with_grad_evals_per_s']}
    return {'config': 'C4s batched sweep 256 thetas x ExactGP SE-ARD N in {512, 1024, 2048} D=8, '
                      '1 GPU (the sizes of the reference\'s particle / sample loops)',
            'sizes': recs}


RUNNERS = {'c2': run_c2, 'c3': run_c3, 'c4': run_c4, 'c4s': run_c4s, 'c5': run_c5}


def run_all(dev, which=('c2', 'c3', 'c4', 'c4s', 'c5'), with_depth3=False):
    return [RUNNERS[c](dev, with_depth3) if c == 'c4s' else RUNNERS[c](dev) for c in which]


if __name__ == '__main__':
    from pygp_amd import _lib
    args = sys.argv[1:]
    out_file = None
    if '--out' in args:
        out_file = args[args.index('--out') + 1]
        args = [a for a in args if a not in ('--out', out_file)]
    depth3 = '--with-depth3' in args
    args = [a for a in args if a != '--with-depth3']
    res = run_all(_lib.Handle(0), args or ['c2', 'c3', 'c4', 'c4s', 'c5'], with_depth3=depth3)
    for r in res:
        print(json.dumps(r), flush=True)
    if out_file:
        with open(out_file, 'w') as f:
            json.dump(res, f, indent=1)
