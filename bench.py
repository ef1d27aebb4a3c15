#!/usr/bin/env python
"""bench.py -- forward evaluations per second of the BayHunter likelihood hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).

  * N = 1: one process, one GPU.
  * N > 1 started through `torch.distributed.run` (WORLD_SIZE/RANK/LOCAL_RANK in the environment):
    this process is one rank; WORLD_SIZE must equal --gpus.
  * N > 1 started as a plain `python bench.py --gpus N`: this process touches neither torch nor the
    GPU; it starts N fresh rank processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), waits for them
    and relays rank 0's line.  (Never an exec of a process that has initialised the GPU.)
  Every rank binds LOCAL_RANK to its own device and fails when fewer devices than ranks are visible
  (stacking ranks on one device is allowed only with BH_DIST_BACKEND=gloo, the one-GPU rehearsal).
  Rank 0 checks that the process group has --gpus ranks, collects every rank's device index / UUID
  and step time, and reports `n_gpus`, `ranks_seen`, `devices_seen`, `per_rank_ms_per_step`.

A "step" is one pass of the hot path over one batch of synthetic layered models that is already
resident in HBM: all dispersion targets of the workload (swd kernel) + the receiver function
(rf kernel) for every model of the rank's shard.  One "evaluation" = all targets of one model
(SURVEY.md section 8d).  Chains are independent (reference src/mcmcOptimizer.py:238-252), so ranks
shard the models with no data-path collective (weak scaling: the per-GPU batch is fixed); the only
collectives are the timing barrier, the MAX over ranks and the gather of the ranks' identities.

Workloads (BASELINE.json):
  joint10  (headline; the configuration the metric "forward evals/sec (SWD+RF, 10-layer)" is quoted
           on) Rayleigh phase velocity at 21 periods + P receiver function (201 samples @ 5 Hz,
           nsamp 512), 10-layer models
  cfg2     Rayleigh phase only, 5 layers, 20 periods, 1024 models
  cfg3     Rayleigh+Love x phase+group, 10 layers, 40 periods, 8192 models
  cfg4     Rayleigh phase (21) + P-RF, 15 layers, 64 models per GPU
  cfg5     ragged 2..31 layers, Rayleigh phase (21) + P-RF
The default run times the headline and then each of cfg2..cfg5 for a bounded number of steps
(`configs` in the line: value, ms_per_step, kernel form, roofline, cpu_baseline per config).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak (spec), SURVEY.md 8d
F_RAYLEIGH, F_LOVE = 190.0, 30.0   # flop per layer step of the period equation

WORKLOADS = {
    #           layers        swd refs                                     periods rf    batch/GPU  cfg id
    'joint10': dict(L=10, refs=['rdispph'], P=21, rf=True, B=524288, cfg=6),
    'cfg2':    dict(L=5, refs=['rdispph'], P=20, rf=False, B=1024, cfg=2),
    'cfg3':    dict(L=10, refs=['rdispph', 'rdispgr', 'ldispph', 'ldispgr'], P=40, rf=False, B=8192, cfg=3),
    'cfg4':    dict(L=15, refs=['rdispph'], P=21, rf=True, B=64, cfg=4),
    'cfg5':    dict(L=(2, 31), refs=['rdispph'], P=21, rf=True, B=8192, cfg=5),
}
CONFIG_ORDER = ['cfg2', 'cfg3', 'cfg4', 'cfg5']
REF_TAGS = {'rdispgr': (2, 1), 'ldispgr': (1, 1), 'rdispph': (2, 0), 'ldispph': (1, 0)}
RF_NSAMP, RF_NOUT = 512, 201


def make_models(wl, B, rank):
    from bayhunter_amd.synthetic import draw_models
    return draw_models(B, wl['L'], seed=1000 * wl['cfg'] + rank, sorted_vs=True)


def algorithmic_bytes(wl, Lmean):
    """32*L + 4 (model) + 8*sum(n_out) + 4*ntargets (err), per evaluation, split by
    kernel (each kernel reads the model once)."""
    model = 32.0 * Lmean + 4
    swd = model + sum(8.0 * wl['P'] + 4 for _ in wl['refs'])
    rf = (model + 8.0 * RF_NOUT) if wl['rf'] else 0.0
    return swd, rf


def rf_flops(Lmean, nfreq_computed):
    """SURVEY.md 8(d) receiver-function formula with the frequencies the kernel actually computes
    (`nact`, bh_rf_active_frequencies: bins whose Gauss-filter weight is below 3e-19 are zero-filled)
    in place of nfreq = 257."""
    return nfreq_computed * (Lmean - 1) * 500 + Lmean * 300 + nfreq_computed * 60 + 5 * RF_NSAMP * 9


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except (IOError, OSError):
        pass
    return 'unknown'


def _cpu_worker(args):
    wl_name, n, seed, use_ref = args
    from oracle import pyoracle as po
    from bayhunter_amd.synthetic import draw_models
    wl = WORKLOADS[wl_name]
    H, VP, VS, RHO, nl = draw_models(n, wl['L'], seed=seed, sorted_vs=True)
    per = np.linspace(1, 41, wl['P'])
    backend = 'ref' if use_ref else 'port'
    t0 = time.perf_counter()
    for ref in wl['refs']:
        iw, ig = REF_TAGS[ref]
        po.swd_batch(H, VP, VS, RHO, nl, per, iw, ig, backend=backend)
    if wl['rf']:
        po.rf_batch(H, VP, VS, RHO, nl, backend=backend)
    return n, time.perf_counter() - t0


def cpu_baseline(wl_name, seconds):
    """Reference native code (oracle/_ref, kind "reference") or the C restatement (kind "port")
    on the host cores of this box, one process per core, on a bounded sample of the same workload
    (`seconds` of work per core at the reference's measured single-core rates).
    Runs in a child process tree that never touches the GPU."""
    import multiprocessing as mp
    from oracle import pyoracle as po
    use_ref = po.have_ref()
    if not use_ref:
        po.port_lib()
    # a one-GPU box exposes 256 logical CPUs but its CPU share is 16: more workers only oversubscribe
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get('BH_CPU_WORKERS', 16)))
    wl = WORKLOADS[wl_name]
    cost = (len(wl['refs']) * wl['P'] / 21.0 * 0.5e-3 + (0.85e-3 if wl['rf'] else 0)) * \
           (np.mean(wl['L']) / 10.0)
    per_core = int(max(64, min(60000, seconds / cost)))
    ctx = mp.get_context('fork')
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(wl_name, per_core, 555000 + i, use_ref) for i in range(cores)])
    wall = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": n / busy, "unit": "evals/s", "cores": cores, "cpu": cpu_model_name(),
            "logical_cpus_visible": len(os.sched_getaffinity(0)),
            "kind": "reference" if use_ref else "port",
            "per_core": n / sum(r[1] for r in res),
            "sample": "%d models (%d per core, %d processes), workload %s, %.1f s of solver time per core, "
                      "%.1f s wall incl. model generation" % (n, per_core, cores, wl_name, busy, wall)}


def measured_traffic(kernel, workload, B, live_ms, lib_hash, tol=0.05):
    """PMC record of `kernel` (HBM bytes per launch, VALU busy, lane utilisation) from the committed
    passes (profiles/*_traffic.json: FETCH_SIZE doubled + WRITE_SIZE, separate --pmc runs) taken at
    this configuration -- bench.py itself cannot collect PMC counters.  The record is only used when
    it belongs to THIS code: its `lib_src_hash` must be the hash compiled into the loaded library and
    the kernel time measured live must agree with the record's rocprof time within `tol` (5 % for the lane kernel,
    10 % for a team kernel, 15 % for a call that runs its targets on several forms).  Returns
    (record or None, note)."""
    import glob
    best, name = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json'))):
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        cfg = d.get('config', '')
        wl_ok = ('--workload ' + workload) in cfg or (workload == 'joint10' and '--workload' not in cfg)
        b_ok = ('--batch %d' % B) in cfg or ('--batch' not in cfg and B == WORKLOADS[workload]['B'])
        for k, rec in d.get('kernels', {}).items():
            if wl_ok and b_ok and k.split('<')[0].split()[-1] == kernel:      # 'void rf_kernel<false>' -> rf_kernel
                best, name = dict(rec, lib_src_hash=d.get('lib_src_hash', ''), git_sha=d.get('git_sha', '')), f
    if best is None:
        return None, 'no PMC record for this configuration under profiles/'
    name = os.path.relpath(name, ROOT)
    if not best['lib_src_hash'] or best['lib_src_hash'] != lib_hash:
        return None, '%s was taken with library %s, this run uses %s: counters not carried over' % (
            name, best['lib_src_hash'] or '(unstamped)', lib_hash)
    ref = best.get('avg_ms_rocprof')
    if not ref or abs(live_ms - ref) > tol * ref:
        return None, '%s: rocprof %.3f ms vs %.3f ms measured now (> %.0f %%): counters not carried over' % (
            name, ref or float('nan'), live_ms, 100 * tol)
    return best, '%s (library %s, rocprof %.2f ms vs %.2f ms now)' % (name, lib_hash, ref, live_ms)


# Period-equation evaluations per model of the reference path and mean layer count, on 128
# benchmark-seed models of each workload: counted once with the evaluation counter of the CPU
# restatement (whose search is the reference's step for step; tests/scenarios/count_dltar.py) and
# committed here -- the timed path and this file's GPU leg never touch the oracle.  SURVEY 8(d) asks
# for the flop figure to be normalised by this reference-path count.
N_DLTAR = {
    'joint10': ({'rdispph': 695.03125}, 10.0),
    'cfg2': ({'rdispph': 608.875}, 5.0),
    'cfg3': ({'rdispph': 987.9765625, 'rdispgr': 1642.8984375, 'ldispph': 970.3828125, 'ldispgr': 1622.109375}, 10.0),
    'cfg4': ({'rdispph': 726.3359375}, 15.0),
    'cfg5': ({'rdispph': 711.9921875}, 16.75),
}


# --------------------------------------------------------------------------------- rank launcher
def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script
    from a parent that has imported neither torch nor the HIP library, wait, relay rank 0's line."""
    import socket
    n = args.gpus
    with socket.socket() as s:                       # a free rendezvous port on the loopback
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                MASTER_PORT=str(port), BH_BENCH_LAUNCHER='bench.py')
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      text=True))
    # Rank 0's stdout is drained by a thread while ALL ranks are polled: a rank that dies before the rendezvous (no
    # GPU visible, import error) must end the run at once, not leave rank 0 sitting in init_process_group until
    # torch's own timeout of several minutes.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get('BH_BENCH_LAUNCH_TIMEOUT', 1500))
    deadline = time.time() + limit
    failed = None
    while failed is None:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            failed = 'rank exit codes %s' % rcs
        elif all(rc == 0 for rc in rcs):
            break
        elif time.time() > deadline:
            failed = 'no result after %.0f s (rank exit codes so far %s)' % (limit, rcs)
        else:
            time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.kill()                              # the exact PIDs this launcher started
        for p in procs:
            p.wait()
    reader.join(timeout=10)
    out0 = ''.join(c for c in chunks if c)
    sys.stdout.write(out0)
    sys.stdout.flush()
    if failed:
        sys.stderr.write('bench.py launcher: %s\n' % failed)
        sys.exit(1)


_REAL_STDOUT = None


def claim_stdout():
    """stdout carries exactly ONE line, the JSON.  Libraries chat on file descriptor 1 -- gloo's "[Gloo] Rank 0
    is connected to ...", RCCL's version banner at the first collective -- so a rank points descriptor 1 at
    stderr for its whole life and keeps the real stdout for `emit`."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + '\n').encode())


# ------------------------------------------------------------------------------------ chain pool
# The set-up of an inversion factorises the receiver function's 201 x 201 correlation matrix with numpy; the workers
# of its OpenBLAS pool (one per core, 64 on a GPU box) then spin for about a tenth of a second, and a 0.3 s sample
# started inside that window loses 50-80 ms to the box's CPU quota (tools/host_phase_probe.py).  The factorisation is
# cached per process (targets.py), so only the warm-up pool pays it; the samples start when the spin is over.
BLAS_SETTLE_S = 0.25
CHAIN_IP = dict(propdist=(0.015, 0.015, 0.015, 0.005, 0.005), acceptance=(40, 100), thickmin=0.1, rcond=1e-5)
# The sampler's workloads (not part of `value`): the tutorial joint inversion (Rayleigh phase + P-RF, observed data in
# tests/golden/tutorial_observed; free vp/vs and noise).  `layers` is the reference's prior on the number of layers
# above the half-space (src/defaults/defaults.ini): (1, 14) = models of 2..15 layers (BASELINE cfg4: "15-layer, 64
# chains/GPU"), (1, 30) = 2..31 layers (cfg5: "transdimensional, 1-30 layers ragged, per-GPU chain pools").
CHAIN_WORKLOADS = {
    'tutorial': dict(layers=(1, 20), chains_per_gpu=4096, burnin=100, main=50),
    'cfg4': dict(layers=(1, 14), chains_per_gpu=64, burnin=100, main=50),
    'cfg5': dict(layers=(1, 30), chains_per_gpu=4096, burnin=100, main=50),
}


def chain_setup(layers):
    from bayhunter_amd import targets as T
    d = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
    sw, rf = np.loadtxt(os.path.join(d, 'st3_rdispph.dat')), np.loadtxt(os.path.join(d, 'st3_prf.dat'))
    joint = T.JointTarget([T.RayleighDispersionPhase(sw[:, 0], sw[:, 1]), T.PReceiverFunction(rf[:, 0], rf[:, 1])])
    priors = dict(vpvs=(1.4, 2.1), layers=layers, vs=(2, 5), z=(0, 60), mohoest=None, rfnoise_corr=0.9,
                  swdnoise_corr=0., rfnoise_sigma=(1e-5, 0.05), swdnoise_sigma=(1e-5, 0.05))
    return joint, priors


def chain_pool_sample(nchains=4096, burnin=100, main_it=50):
    """End-to-end sampler on top of the timed path, one GPU: a lock-step pool of chains on the tutorial
    inversion, chain iterations per second incl. host proposals/acceptance.  Every pool is closed before the
    next one is made (ChainPool.close: its evaluation plans retire their streams)."""
    import torch
    from bayhunter_amd.chains import ChainPool
    if os.environ.get('BH_BENCH_INJECT_FAILURE') == 'chain_pool':        # test hook: a leg that fails must be seen
        raise RuntimeError('injected failure of the chain-pool leg')
    joint, priors = chain_setup(CHAIN_WORKLOADS['tutorial']['layers'])
    ip = dict(CHAIN_IP, iter_burnin=burnin, iter_main=main_it)
    seeds = np.arange(nchains) % 1000
    # a short pool of the same size first: kernel forms loaded on their first launch, helper threads and pinned
    # buffers are not chain iterations (0.05-0.1 s of a 0.3 s sample when they fell into it)
    with ChainPool(joint, initparams=dict(ip, iter_burnin=6, iter_main=2), modelpriors=priors, seeds=seeds, nmodels=9) as warm:
        warm.run()
    time.sleep(BLAS_SETTLE_S)
    # three samples, the best one reported (all three in the record; since the set-up's BLAS spin is kept out of
    # them -- BLAS_SETTLE_S -- they agree within a few per cent)
    def sample(n, its_burnin, its_main, repeats, **kw):
        runs = []
        par = dict(ip, iter_burnin=its_burnin, iter_main=its_main)
        for _ in range(repeats):
            with ChainPool(joint, initparams=par, modelpriors=priors, seeds=seeds[:n], **kw) as pool:
                t0 = time.perf_counter()
                pool.run()
                torch.cuda.synchronize()
                runs.append((time.perf_counter() - t0, int(pool.evaluated), dict(pool.seconds), pool.lookahead, pool.advance(),
                             len(pool.groups)))
        dt, evaluated, seconds, lookahead, (calls, advanced, rows), groups = min(runs, key=lambda r: r[0])
        its = its_burnin + its_main
        return {"value": n * its / dt, "unit": "chain iterations/s", "nchains": n, "iterations": its,
                "models_evaluated": evaluated, "samples": [round(n * its / r[0]) for r in runs],
                "lookahead": lookahead, "device_calls": calls,
                "iterations_per_call": round(advanced / max(calls, 1) / (n / groups), 2),
                "host_seconds": {k: round(v, 4) for k, v in seconds.items()}}
    rec = sample(nchains, burnin, main_it, 3)
    rec.update({"ok": True, "workload": "tutorial joint inversion (rdispph + prf), free vp/vs and noise, 2-21 layers"})
    # the pool size the reference's users run (tutorial: 5 chains; one process per chain there): a call is bound by its
    # latency, and the pool looks ahead -- several iterations of a chain per call (bh_chains_set_lookahead; same chains)
    with ChainPool(joint, initparams=dict(ip, iter_burnin=6, iter_main=2), modelpriors=priors, seeds=seeds[:64], nmodels=9) as warm:
        warm.run()
    small = sample(64, 400, 200, 2)
    small["one_proposal_per_call"] = sample(64, 400, 200, 1, lookahead=1)["value"]
    rec["small_pool"] = small
    return rec


def chain_digest(blocks):
    """sha256 over the sample blocks in global chain order (models, likes, iter, noise, vpvs, naccepted): equal for
    the same seeds whatever the number of ranks."""
    import hashlib
    h = hashlib.sha256()
    for k in ('models', 'likes', 'iter', 'noise', 'vpvs', 'naccepted'):
        h.update(np.ascontiguousarray(blocks[k]).tobytes())
    return h.hexdigest()[:16]


def sharded_chain_pools(rank, world, ranks, backend, workloads=None, evaluator=None):
    """N > 1: the sampler as BASELINE.json words it -- cfg4 (64 chains per GPU, models of up to 15 layers) and cfg5
    (per-GPU pools, ragged 2-31 layers) with the chains block-sharded over the ranks and NO collective while
    sampling (reference: one process per chain, src/mcmcOptimizer.py:238-252) -- and then the one exchange the
    reference has: the per-chain sample blocks (its shared arrays, src/mcmcOptimizer.py:92-125, merged by
    src/Plotting.py:161-262) gathered over the process group (RCCL with backend nccl): to the root, raw and
    thinned, and as an all-gather for comparison.  Returns (chain_pool_sharded, gather) for rank 0's line.
    `workloads` / `evaluator` (test hooks): other pool sizes, and a likelihood function in place of the device's --
    the CPU tier runs these legs over gloo with a made-up likelihood (tests/test_distributed.py)."""
    import torch
    import torch.distributed as dist
    from bayhunter_amd.chains import ChainPool
    from bayhunter_amd.distributed import gather_rows, gather_rows_to_root
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')

    def sync():
        if dev.type == 'cuda':
            torch.cuda.synchronize()

    def timed(fn):
        ranks.barrier(); sync()
        t0 = time.perf_counter()
        out = fn()
        sync(); ranks.barrier()
        return out, ranks.max(time.perf_counter() - t0)

    pools_rec, gather_rec = {}, {}
    for name in ('cfg4', 'cfg5'):
        wl = (workloads or CHAIN_WORKLOADS)[name]
        joint, priors = chain_setup(wl['layers'])
        ip = dict(CHAIN_IP, iter_burnin=wl['burnin'], iter_main=wl['main'])
        total = wl['chains_per_gpu'] * world
        seeds = np.arange(total) % 1000
        shard = (rank, world)
        with ChainPool(joint, initparams=dict(ip, iter_burnin=6, iter_main=2), modelpriors=priors, seeds=seeds,
                       nmodels=9, shard=shard, evaluator=evaluator) as warm:
            warm.run()
        time.sleep(BLAS_SETTLE_S)
        best = None
        for _ in range(3 if name == 'cfg4' else 2):
            pool = ChainPool(joint, initparams=ip, modelpriors=priors, seeds=seeds, shard=shard, evaluator=evaluator)
            _, dt = timed(lambda: pool.run())
            pool.close()
            if best is None or dt < best[0]:
                best = (dt, pool)
        dt, pool = best
        its = wl['burnin'] + wl['main']
        evaluated = ranks.all_floats(float(pool.evaluated))
        pools_rec[name] = {
            "value": total * its / dt, "unit": "chain iterations/s", "chains_per_gpu": wl['chains_per_gpu'],
            "nchains": total, "iterations": its, "seconds": dt, "models_evaluated": int(sum(evaluated)),
            "lookahead": pool.lookahead, "device_calls_rank0": pool.advance()[0],
            "layers_prior": list(wl['layers']), "sharding": "chains block-partitioned over ranks, no collective while sampling",
            "workload": "tutorial joint inversion (rdispph + prf), free vp/vs and noise, models of 2-%d layers" % (wl['layers'][1] + 1)}
        # -- the exchange: this pool's sample blocks over the process group
        width = pool.models.shape[2] + pool.misfits.shape[2] + 1 + pool.noise.shape[2] + 1
        payload = np.concatenate([pool.models, pool.misfits, pool.likes[..., None], pool.noise, pool.vpvs[..., None]],
                                 axis=2).astype(np.float32)          # rows x width float32
        t_payload = torch.from_numpy(payload).to(dev)
        nbytes_rank = payload.nbytes
        gather_rows_to_root(t_payload, total)                         # first-use costs of the transport (connections,
        gather_rows(t_payload, total)                                 # staging buffers), both patterns, untimed
        full, t_root = timed(lambda: gather_rows_to_root(t_payload, total))
        _, t_all = timed(lambda: gather_rows(t_payload, total))
        blocks, t_blocks = timed(lambda: pool.gather())
        maxmodels = 50
        thinned, t_thin = timed(lambda: pool.gather_final(maxmodels))
        rec = {
            "payload": "%d chains x %d rows x %d float32 per rank (models | misfits | likes | noise | vpvs)" % (
                pool.nchains, pool.nmodels, width),
            "bytes_per_rank": nbytes_rank, "bytes_received_by_root": nbytes_rank * (world - 1), "backend": backend,
            "to_root": {"ms": t_root * 1e3, "GB/s": nbytes_rank * (world - 1) / max(t_root, 1e-9) / 1e9,
                        "how": "every rank sends its block once, point to point (distributed.gather_rows_to_root)"},
            "all_gather": {"ms": t_all * 1e3, "GB/s": nbytes_rank * (world - 1) * world / max(t_all, 1e-9) / 1e9,
                           "how": "padded fixed-shape all_gather, every rank ends with everything (distributed.gather_rows)"},
            "ChainPool.gather": {"ms": t_blocks * 1e3, "note": "the six arrays one by one from host memory, incl. the copies to and from the device"},
            "ChainPool.gather_final": {"ms": t_thin * 1e3, "maxmodels": maxmodels,
                                       "note": "residence-time weighting + thinning on every rank (host), then ragged blocks to the root"},
        }
        if rank == 0:
            assert full.shape[0] == total and len(thinned) == total
            rec["ChainPool.gather_final"]["rows_at_root"] = int(sum(t.shape[0] for t in thinned))
            rec["chains_sha256"] = chain_digest(blocks)
            rec["accepted_total"] = int(blocks['naccepted'].sum())
        gather_rec[name] = rec
        del t_payload, full
    return pools_rec, gather_rec


# -------------------------------------------------------------------------------- one workload
class Ranks(object):
    """The process group as the timed region sees it (a no-op for one rank)."""

    def __init__(self, world, backend, device_index, active=None):
        self.world, self.backend, self.dev = world, backend, device_index
        self.active = world > 1 if active is None else active      # a process group exists

    def barrier(self):
        if self.active:
            import torch.distributed as dist
            if self.backend == 'nccl':
                dist.barrier(device_ids=[self.dev])
            else:
                dist.barrier()

    def max(self, value):
        from bayhunter_amd.distributed import max_over_ranks
        return max_over_ranks(value, device='cuda' if self.backend == 'nccl' else 'cpu')

    def all_floats(self, value):
        if not self.active:
            return [float(value)]
        import torch
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64, device='cuda' if self.backend == 'nccl' else 'cpu')
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t)
        return [float(p.item()) for p in parts]


def run_workload(name, B, rank, ranks, steps, warmup, serial=False, budget_s=None):
    """Upload one batch of `name`, time `steps` passes (or, with budget_s, as many as fit that many
    seconds after a probe), then the per-kernel durations with events on the launch stream."""
    import torch
    from bayhunter_amd import _lib as bhlib
    from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec
    wl = WORKLOADS[name]
    per = np.linspace(1, 41, wl['P'])
    H, VP, VS, RHO, nl = make_models(wl, B, rank)
    swd_specs = [SwdSpec(r, per) for r in wl['refs']]
    rf_specs = [RfSpec('prf', np.linspace(-5, 35, RF_NOUT))] if wl['rf'] else []
    eng = ForwardEngine(swd=swd_specs, rf=rf_specs)               # the product path: one engine
    eng_only_swd = ForwardEngine(swd=swd_specs)                     # per-kernel timing only
    eng_only_rf = ForwardEngine(rf=rf_specs) if rf_specs else None
    if eng_only_rf:                                                 # same output row layout
        eng_only_rf._rfp[0].out_off = eng._rfp[0].out_off
        eng_only_rf.row = eng.row
    eng_only_swd.row = eng.row
    dmodels = eng.upload(H, VP, VS, RHO, nl)                       # inputs resident in HBM
    out, err = eng.alloc_out(B)
    if serial:
        eng.overlap = False       # profiling passes: rf_kernel behind swd_kernel on one stream

    def step():
        # a sampler brings new models every step: the processing order (a sort by depth and S travel
        # time, engine.reorder) is part of the step, not of the upload
        eng.run(eng.reorder(dmodels.packed, dmodels.nlay, depth=dmodels.depth, mean_depth=dmodels.mean_depth), out=out, err=err)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if budget_s is not None:          # bounded configs: the step count from a short probe, same on all ranks
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        est = ranks.max((time.perf_counter() - t0) / 3)
        steps = int(max(5, min(400, budget_s / max(est, 1e-5))))
    ranks.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    ranks.barrier()
    torch.cuda.synchronize()
    dt = ranks.max(time.perf_counter() - t0)
    form = bhlib.load().bh_swd_last_form()
    import ctypes
    fl = (ctypes.c_int * len(swd_specs))()
    bhlib.check(bhlib.load().bh_swd_last_forms(fl, len(swd_specs)))
    forms = {r: int(f) for r, f in zip(wl['refs'], fl)}      # per target: a call may use one form per target

    # per-kernel durations for the roofline: the same launches, serialised on one stream, bracketed
    # by events on that stream (in the timed steps above rf_kernel overlaps the tail of swd_kernel)
    nk = min(steps, 5)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(nk)]
    eng_only_swd.run(dmodels, out=out, err=err)         # (one untimed pass: first-use costs of these two
    if eng_only_rf:                                      # engines, e.g. lazily loaded fill kernels)
        eng_only_rf.run(dmodels, out=out, err=err)
    torch.cuda.synchronize()
    for i in range(nk):
        ev[i][0].record()
        eng_only_swd.run(dmodels, out=out, err=err)
        ev[i][1].record()
        if eng_only_rf:
            eng_only_rf.run(dmodels, out=out, err=err)
        ev[i][2].record()
    torch.cuda.synchronize()
    ms_swd = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_rf = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if eng_only_rf else 0.0
    nact = int(bhlib.load().bh_rf_active_frequencies(eng._rfp[0])) if eng_only_rf else 0
    return dict(name=name, B=B, steps=steps, dt=dt, dt_local=dt_local, ms_swd=ms_swd, ms_rf=ms_rf,
                nerr=int(err.sum().item()), form=form, forms=forms, nact=nact)


def kernel_symbol(form):
    """The dispersion kernel a form runs as (kernels.hip), as rocprofv3 names it."""
    return 'swd_kernel' if form == 0 else 'swd_team_kernel' if form == 64 else 'swd_team%d_kernel' % form


def form_name(form, forms=None):
    if forms and len(set(forms.values())) > 1:         # targets of one call on different forms (capi.hip: plan_forms)
        return 'swd kernels per target: ' + ', '.join('%s %s' % (r, 'lane' if f == 0 else 'team%d' % f) for r, f in forms.items())
    return 'swd_kernel (one search per lane)' if form == 0 else 'swd_team kernel, %d lanes per search' % form


def rooflines(r, lib_hash):
    """roofline of the dominant kernel of a workload + the same figures for both kernels."""
    wl = WORKLOADS[r['name']]
    counts, Lmean = N_DLTAR[r['name']]
    B = r['B']
    bytes_swd, bytes_rf = algorithmic_bytes(wl, Lmean)
    flop_swd = sum(counts[k] * (Lmean - 1) * (F_RAYLEIGH if REF_TAGS[k][0] == 2 else F_LOVE) for k in wl['refs'])
    flop_rf = rf_flops(Lmean, r['nact']) if wl['rf'] else 0.0

    def one(kernel, ms, flop, nbytes, extra, tol=0.05):
        tflops = flop * B / (ms * 1e-3) / 1e12
        gbs = nbytes * B / (ms * 1e-3) / 1e9
        pmc, note = measured_traffic(kernel, r['name'], B, ms, lib_hash, tol)
        pmc = pmc or {}
        d = {"kernel": kernel, "bound": "fp64_valu", "achieved": tflops, "peak": FP64_PEAK_TFLOPS,
             "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS, "traffic": pmc.get('hbm_bytes'),
             "kernel_ms": ms, "algorithmic_flop_per_launch": flop * B,
             "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": nbytes * B,
                     "traffic_bytes_per_launch_pmc": pmc.get('hbm_bytes')},
             "valu_pipes_busy_pmc": pmc.get('valu_busy'),
             "lane_utilisation_pmc": pmc.get('lane_utilisation'),
             # share of the VALU issue slots doing lane work: what is left to gain by scheduling
             "issue_frac_pmc": (pmc['valu_busy'] * pmc['lane_utilisation']
                                if pmc.get('valu_busy') and pmc.get('lane_utilisation') else None),
             "pmc_source": note}
        d.update(extra)
        return d

    # r['form'] is the form of the call's first launch: the one that takes its heaviest target.  A call that
    # spreads its targets over several forms runs them side by side; it lasts about as long as that launch, whose
    # counters are carried over with a wider time window.
    mixed = len(set(r['forms'].values())) > 1
    swd = one(kernel_symbol(r['form']), r['ms_swd'], flop_swd, bytes_swd,
              {"n_dltar_per_eval": counts, "form": form_name(r['form'], r['forms']),
               "flop_model": "N_dltar x (L-1) x 190 (Rayleigh) / 30 (Love), FMA = 2"},
              # (a team kernel's time varies by up to 7 % from launch to launch under rocprofv3 -- cfg5: 4.19-5.6 ms over
              # 13 launches, which waves share a SIMD -- the lane kernel's by 1 %)
              tol=0.15 if mixed else 0.10 if r['form'] else 0.05)
    rf = one('rf_kernel', r['ms_rf'], flop_rf, bytes_rf,
             {"frequencies_computed": r['nact'], "frequencies_total": RF_NSAMP // 2 + 1,
              "flop_model": "nact x (L-1) x 500 + L x 300 + nact x 60 + 5 nsamp log2 nsamp (SURVEY 8d with "
                            "the computed frequencies)"}) if wl['rf'] else None
    dom = swd if (rf is None or r['ms_swd'] >= r['ms_rf']) else rf
    return dom, swd, rf, flop_swd + flop_rf


def workload_label(name, B):
    wl = WORKLOADS[name]
    return "%s: %s%s, %s layers, %d periods, %d models/GPU/step" % (
        name, '+'.join(wl['refs']), '+prf(201 samples, nsamp 512)' if wl['rf'] else '', str(wl['L']), wl['P'], B)


# ----------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='joint10', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='models per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', default=None, metavar='WORKLOADS',
                    help='(internal) comma separated workload=seconds list; prints a JSON dict')
    ap.add_argument('--no-chain-pool', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the cfg2..cfg5 lines after the headline')
    ap.add_argument('--probe-ranks', action='store_true',
                    help='(test hook) rendezvous only, on the CPU over gloo: prints who showed up, no GPU work')
    ap.add_argument('--serial', action='store_true',
                    help='profiling: rf_kernel behind swd_kernel on one stream (no back-filling)')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')

    if args.cpu_baseline_only:
        res = {}
        for item in args.cpu_baseline_only.split(','):
            name, _, sec = item.partition('=')
            res[name] = cpu_baseline(name, float(sec or 3.0))
        print(json.dumps(res))
        return

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        launch_ranks(args)                      # this process never touches torch or the GPU
        return

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d: start one rank per GPU '
                 '(torch.distributed.run --nproc-per-node %d, or plain `python bench.py --gpus %d`)'
                 % (args.gpus, world, args.gpus, args.gpus))
    claim_stdout()                               # from here on this process is a rank: stdout = the one JSON line
    if args.probe_ranks:                         # launcher / rendezvous check without a GPU (tests)
        if os.environ.get('BH_BENCH_PROBE_DIE_RANK') == str(rank):      # test hook: a rank that never reaches the rendezvous
            sys.exit(3)
        import torch.distributed as dist
        seen = [None] * world
        if world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('gloo')
            dist.all_gather_object(seen, (rank, local_rank, os.getpid()))
            dist.destroy_process_group()
        else:
            seen = [(rank, local_rank, os.getpid())]
        if rank == 0:
            emit(json.dumps({"probe": True, "n_gpus": world, "ranks_seen": len(set(s[0] for s in seen)),
                              "local_ranks": sorted(s[1] for s in seen), "pids": len(set(s[2] for s in seen)),
                             "launcher": os.environ.get('BH_BENCH_LAUNCHER', 'external' if world > 1 else None)}))
        return
    headline = args.workload == 'joint10' and args.batch is None
    with_configs = headline and not args.no_configs

    # CPU baseline first, in a child that never initialises the GPU (rank 0, N=1 only)
    cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        want = ['%s=%g' % (args.workload, 22.0)]      # >= 10 s of solver time per core on the box's CPU
        if with_configs:
            want += ['%s=%g' % (c, 4.0) for c in CONFIG_ORDER]
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', ','.join(want)],
                           capture_output=True, text=True)
        if r.returncode == 0:
            cpu = json.loads(r.stdout.strip().splitlines()[-1])
        else:
            sys.stderr.write('cpu baseline failed: %s\n' % r.stderr[-2000:])

    import torch
    import torch.distributed as dist
    from bayhunter_amd import _lib as bhlib

    # One rank per GPU over RCCL ("nccl").  BH_DIST_BACKEND=gloo is only for rehearsing the N>1 code
    # path on a box with fewer GPUs than ranks: the ranks then share devices and say so in the line.
    backend = os.environ.get('BH_DIST_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()              # (does not initialise the GPU)
    if ndev < 1:
        sys.exit('bench.py: no GPU visible (there is no CPU fallback)')
    if ndev < world and backend != 'gloo':
        sys.exit('bench.py: %d ranks but only %d GPU(s) visible: one rank per GPU is required '
                 '(BH_DIST_BACKEND=gloo allows sharing, for rehearsal only)' % (world, ndev))
    dev = local_rank if ndev >= world else local_rank % ndev
    torch.cuda.set_device(dev)
    # BH_BENCH_FORCE_DIST=1 (test hook): a one-rank process group, so that the collective branches below run
    # over RCCL on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get('BH_BENCH_FORCE_DIST'))
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            sys.exit('bench.py: process group has %d ranks, --gpus %d' % (dist.get_world_size(), args.gpus))
    ranks = Ranks(world, backend, dev, active=use_dist)

    # who is here: every rank's device, gathered over the process group (RCCL with backend nccl)
    props = torch.cuda.get_device_properties(dev)
    me = torch.zeros(18, dtype=torch.int64)
    me[0], me[1] = rank, dev
    uuid = getattr(props, 'uuid', None)
    ub = uuid.bytes if uuid is not None and hasattr(uuid, 'bytes') else str(uuid).encode()[:16].ljust(16, b'\0')
    me[2:18] = torch.tensor(list(ub[:16]), dtype=torch.int64)
    if use_dist:
        me = me.to('cuda' if backend == 'nccl' else 'cpu')
        everyone = [torch.empty_like(me) for _ in range(world)]
        dist.all_gather(everyone, me)
        everyone = [e.cpu() for e in everyone]
    else:
        everyone = [me]
    ranks_seen = sorted(int(e[0]) for e in everyone)
    devices = ['%d:%s' % (int(e[1]), bytes(int(x) for x in e[2:18]).hex()) for e in everyone]
    devices_seen = len(set(devices))
    if ranks_seen != list(range(args.gpus)):
        sys.exit('bench.py: ranks seen %s, expected 0..%d' % (ranks_seen, args.gpus - 1))
    if backend == 'nccl' and devices_seen != world:
        sys.exit('bench.py: %d ranks on %d distinct devices' % (world, devices_seen))

    B = args.batch or WORKLOADS[args.workload]['B']
    head = run_workload(args.workload, B, rank, ranks, args.steps, args.warmup, serial=args.serial)
    per_rank_ms = [t / args.steps * 1e3 for t in ranks.all_floats(head['dt_local'])]
    cfg_runs = []
    if with_configs:
        for c in CONFIG_ORDER:
            cfg_runs.append(run_workload(c, WORKLOADS[c]['B'], rank, ranks, 0, 3, serial=args.serial, budget_s=0.6))

    # N > 1 (or the one-rank rehearsal of its code): the sharded sampler and the gather of its sample blocks --
    # collectives, so every rank takes part
    sharded = None
    if use_dist and headline and not args.no_chain_pool:
        sharded = sharded_chain_pools(rank, world, ranks, backend)

    exit_code = 0
    if rank == 0:
        lib_hash = bhlib.loaded_hash()
        dom, swd, rf, flop_eval = rooflines(head, lib_hash)
        value = world * B * args.steps / head['dt']
        dom = dict(dom, flop_per_eval_reference_path=flop_eval,
                   note="achieved = reference-path flops / kernel time, timed live with events on the launch "
                        "stream; traffic and the *_pmc fields come from the committed rocprofv3 --pmc passes and "
                        "are null unless that profile was taken with this very library and its kernel time "
                        "matches this run's")
        res = {
            "metric": "forward evals/sec (SWD+RF, 10-layer)" if args.workload == 'joint10'
                      else "forward evals/sec (%s)" % args.workload,
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head['dt'] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "ranks_seen": len(ranks_seen), "devices_seen": devices_seen, "devices": devices,
            "per_rank_ms_per_step": per_rank_ms, "dist_backend": backend if use_dist else None,
            "launcher": os.environ.get('BH_BENCH_LAUNCHER', 'external' if world > 1 else None),
            "config": {"workload": workload_label(args.workload, B),
                       "models_per_gpu": B, "sharding": "models block-partitioned over ranks, no data-path collective",
                       "err_models": head['nerr'], "library": bhlib.load().bh_version().decode(),
                       "overlap": not args.serial},
            # The dominant kernel is an fp64 scalar recurrence: FP64 vector issue is the roofline that
            # binds it, so that is the headline fraction; the HBM figure the contract asks
            # for is kept beside it.
            "roofline": dom,
            "roofline_rf": rf,
            "kernels_ms": {"swd_kernel": head['ms_swd'], "rf_kernel": head['ms_rf'],
                           "note": "serialised; in the timed steps rf_kernel runs on a second stream, co-resident with swd_kernel (2 x 192 + 128 VGPRs per SIMD)"},
            "cpu_baseline": cpu.get(args.workload),
        }
        if cfg_runs:
            cfgs = {}
            for r in cfg_runs:
                d, s, f, _ = rooflines(r, lib_hash)
                cfgs[r['name']] = {
                    "workload": workload_label(r['name'], r['B']),
                    "value": world * r['B'] * r['steps'] / r['dt'], "unit": "evals/s",
                    "ms_per_step": r['dt'] / r['steps'] * 1e3, "steps": r['steps'],
                    "kernel": form_name(r['form'], r['forms']), "kernels_ms": {"swd": r['ms_swd'], "rf": r['ms_rf']},
                    "roofline_frac": d['frac'],
                    "roofline": {k: d[k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'kernel_ms', 'traffic',
                                                   'valu_pipes_busy_pmc', 'lane_utilisation_pmc', 'issue_frac_pmc',
                                                   'pmc_source')},
                    "hbm_frac": d['hbm']['frac'], "err_models": r['nerr'],
                    "cpu_baseline": cpu.get(r['name'])}
            res["configs"] = cfgs
        if sharded is not None:
            res["chain_pool_sharded"], res["gather"] = sharded
        failed_legs = []
        if world == 1 and not use_dist and headline and not args.no_chain_pool:
            # A failing leg must be seen: the line is still emitted (it carries the error) and the process then exits
            # non-zero -- after a library error the state of the process is unknown, the run is not a healthy one.
            try:
                res["chain_pool"] = chain_pool_sample()
            except Exception as e:
                import traceback
                traceback.print_exc()
                res["chain_pool"] = {"ok": False, "value": None, "error": repr(e)}
                failed_legs.append('chain_pool')
        res["ok"] = not failed_legs
        emit(json.dumps(res))
        if failed_legs:
            sys.stderr.write('bench.py: failed legs: %s\n' % ', '.join(failed_legs))
            exit_code = 1
    if use_dist:
        dist.barrier() if backend != 'nccl' else dist.barrier(device_ids=[dev])
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == '__main__':
    main()
