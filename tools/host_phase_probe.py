"""GPU box: how long the host-side calls of a chain-pool iteration take, call by call -- bh_chains_propose,
bh_eval_submit, bh_eval_wait, bh_chains_accept -- for several pool sizes: median / 90th / 99th percentile / maximum
in microseconds and the sum.  Finds host stalls (helper threads asleep or descheduled, CPU quota) that a
per-run total hides.   usage: python tools/host_phase_probe.py [nchains ...]   (env: BH_CHAIN_THREADS, BH_CHAIN_SPIN_US)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'scenarios'))


def cgroup_stat():
    out = {}
    try:
        for l in open('/sys/fs/cgroup/cpu.stat'):
            k, v = l.split()
            out[k] = int(v)
    except (IOError, OSError, ValueError):
        pass
    return out


def thread_times():
    """{tid: (name, utime + stime in clock ticks)} of this process"""
    out = {}
    for tid in os.listdir('/proc/self/task'):
        try:
            f = open('/proc/self/task/%s/stat' % tid).read()
            name = f[f.index('(') + 1:f.rindex(')')]
            rest = f[f.rindex(')') + 2:].split()
            out[tid] = (name, int(rest[11]) + int(rest[12]))
        except (IOError, OSError, ValueError):
            pass
    return out


def main():
    from chain_scenario import CASES, joint_target
    from bayhunter_amd.chains import ChainPool
    case = CASES['tutorial']
    data = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]:
        def mk(it, **kw):
            return ChainPool(joint_target(data), initparams=dict(case['initparams'], iter_burnin=it, iter_main=it // 2, acceptance=(40, 100)),
                             modelpriors=case['priors'], seeds=np.arange(n) % 1000, **kw)
        with mk(6, nmodels=12) as warm:
            warm.run()
        time.sleep(float(os.environ.get('PROBE_SETTLE_S', 0.25)))     # the OpenBLAS pool's idle spin after the set-up
        with mk(120) as pool:
            ev = dict(propose=[], submit=[], wait=[], accept=[])
            for g in pool.groups:
                op, oa = g.propose, g.accept

                def propose(op=op):
                    t = time.perf_counter(); r = op(); ev['propose'].append(time.perf_counter() - t); return r

                def accept(a, b, oa=oa):
                    t = time.perf_counter(); oa(a, b); ev['accept'].append(time.perf_counter() - t)
                g.propose, g.accept = propose, accept
            osub, ocol = pool.evaluator.submit, pool.evaluator.collect

            def submit(*a):
                t = time.perf_counter(); r = osub(*a); ev['submit'].append(time.perf_counter() - t); return r

            def collect(tk):
                t = time.perf_counter(); r = ocol(tk); ev['wait'].append(time.perf_counter() - t); return r
            pool.evaluator.submit, pool.evaluator.collect = submit, collect
            c0, th0 = cgroup_stat(), thread_times()
            t0 = time.perf_counter()
            pool.run()
            dt = time.perf_counter() - t0
            c1, th1 = cgroup_stat(), thread_times()
        busy = sorted(((th1[t][1] - th0.get(t, (0, 0))[1]) / os.sysconf('SC_CLK_TCK'), th1[t][0]) for t in th1)
        rec = dict(cgroup={k: c1.get(k, 0) - c0.get(k, 0) for k in ('usage_usec', 'nr_periods', 'nr_throttled', 'throttled_usec')},
                   threads_total=len(th1), cpu_seconds_by_thread=[(n, round(s, 3)) for s, n in busy[-8:] if s > 0],
                   cpu_seconds_all_threads=round(sum(b[0] for b in busy), 3),
                   nchains=n, seconds=round(dt, 4), its_per_s=round(n * 180 / dt), threads=os.environ.get('BH_CHAIN_THREADS'),
                   spin_us=os.environ.get('BH_CHAIN_SPIN_US'))
        for k, v in ev.items():
            a = np.array(v) * 1e6
            rec[k] = dict(calls=len(a), sum_ms=round(a.sum() / 1e3, 2), p50=round(float(np.median(a)), 1), p90=round(float(np.percentile(a, 90)), 1),
                          p99=round(float(np.percentile(a, 99)), 1), max=round(float(a.max()), 1))
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
