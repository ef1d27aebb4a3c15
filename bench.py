#!/usr/bin/env python
"""bench.py -- forward evaluations per second of the BayHunter likelihood hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N>1 launched through
torch.distributed.run, one rank per GPU) prints ONE JSON line on rank 0.

A "step" is one pass of the hot path over one batch of synthetic layered models that is already
resident in HBM: all dispersion targets of the workload (swd kernel) + the receiver function
(rf kernel) for every model of the rank's shard.  One "evaluation" = all targets of one model
(SURVEY.md section 8d).  Chains are independent, so ranks shard the models with no data-path
collective (weak scaling: the per-GPU batch is fixed); the only collectives are the timing barrier
and the MAX over ranks.

Workloads (BASELINE.json):
  joint10  (default; the configuration the metric "forward evals/sec (SWD+RF, 10-layer)" is quoted
           on) Rayleigh phase velocity at 21 periods + P receiver function (201 samples @ 5 Hz,
           nsamp 512), 10-layer models
  cfg2     Rayleigh phase only, 5 layers, 20 periods, 1024 models
  cfg3     Rayleigh+Love x phase+group, 10 layers, 40 periods, 8192 models
  cfg4     Rayleigh phase (21) + P-RF, 15 layers, 64 models per GPU
  cfg5     ragged 2..31 layers, Rayleigh phase (21) + P-RF
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
REF_TAGS = {'rdispgr': (2, 1), 'ldispgr': (1, 1), 'rdispph': (2, 0), 'ldispph': (1, 0)}


def make_models(wl, B, rank):
    from bayhunter_amd.synthetic import draw_models
    return draw_models(B, wl['L'], seed=1000 * wl['cfg'] + rank, sorted_vs=True)


def algorithmic_bytes(wl, Lmean):
    """32*L + 4 (model) + 8*sum(n_out) + 4*ntargets (err), per evaluation, split by
    kernel (each kernel reads the model once)."""
    model = 32.0 * Lmean + 4
    swd = model + sum(8.0 * wl['P'] + 4 for _ in wl['refs'])
    rf = (model + 8.0 * 201) if wl['rf'] else 0.0
    return swd, rf


# ------------------------------------------------------------------------------------ CPU baseline
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


def cpu_baseline(wl_name, per_core=None):
    """Reference native code (oracle/_ref, kind "reference") or the C restatement (kind "port")
    on all host cores of this box, one process per core, on a bounded sample of the same workload.
    Runs in a child process tree that never touches the GPU."""
    import multiprocessing as mp
    from oracle import pyoracle as po
    use_ref = po.have_ref()
    if not use_ref:
        po.port_lib()
    # a one-GPU box exposes 256 logical CPUs but its CPU share is 16: more workers only oversubscribe
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get('BH_CPU_WORKERS', 16)))
    wl = WORKLOADS[wl_name]
    if per_core is None:
        # ~10 s per core at the reference's measured single-core rates
        cost = (len(wl['refs']) * wl['P'] / 21.0 * 0.5e-3 + (0.85e-3 if wl['rf'] else 0)) * \
               (np.mean(wl['L']) / 10.0)
        per_core = int(max(64, min(20000, 8.0 / cost)))
    ctx = mp.get_context('fork')
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(wl_name, per_core, 555000 + i, use_ref) for i in range(cores)])
    wall = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": n / busy, "unit": "evals/s", "cores": cores,
            "kind": "reference" if use_ref else "port",
            "per_core": n / sum(r[1] for r in res),
            "sample": "%d models (%d per core, %d processes), workload %s, %.1f s wall"
                      % (n, per_core, cores, wl_name, wall)}


def measured_traffic(kernel, workload, B, live_ms, lib_hash):
    """PMC record of `kernel` (HBM bytes per launch, VALU busy, lane utilisation) from the committed
    passes (profiles/*_traffic.json: FETCH_SIZE doubled + WRITE_SIZE, separate --pmc runs) taken at
    this configuration -- bench.py itself cannot collect PMC counters.  The record is only used when
    it belongs to THIS code: its `lib_src_hash` must be the hash compiled into the loaded library and
    the kernel time measured live must agree with the record's rocprof time within 5 %.  Returns
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
        if wl_ok and b_ok and kernel in d.get('kernels', {}):
            best, name = dict(d['kernels'][kernel], lib_src_hash=d.get('lib_src_hash', ''), git_sha=d.get('git_sha', '')), f
    if best is None:
        return None, 'no PMC record for this configuration under profiles/'
    name = os.path.relpath(name, ROOT)
    if not best['lib_src_hash'] or best['lib_src_hash'] != lib_hash:
        return None, '%s was taken with library %s, this run uses %s: counters not carried over' % (
            name, best['lib_src_hash'] or '(unstamped)', lib_hash)
    ref = best.get('avg_ms_rocprof')
    if not ref or abs(live_ms - ref) > 0.05 * ref:
        return None, '%s: rocprof %.2f ms vs %.2f ms measured now (> 5 %%): counters not carried over' % (
            name, ref or float('nan'), live_ms)
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


# ----------------------------------------------------------------------------------------- main
def chain_pool_sample(nchains=4096, burnin=100, main_it=50):
    """End-to-end sampler on top of the timed path (not part of `value`): a lock-step pool of
    chains on the tutorial inversion (Rayleigh phase + P-RF, observed data in
    tests/golden/tutorial_observed), chain iterations per second incl. host proposals/acceptance."""
    try:
        import torch
        from bayhunter_amd import targets as T
        from bayhunter_amd.chains import ChainPool
        d = os.path.join(ROOT, 'tests', 'golden', 'tutorial_observed')
        sw, rf = np.loadtxt(os.path.join(d, 'st3_rdispph.dat')), np.loadtxt(os.path.join(d, 'st3_prf.dat'))
        joint = T.JointTarget([T.RayleighDispersionPhase(sw[:, 0], sw[:, 1]), T.PReceiverFunction(rf[:, 0], rf[:, 1])])
        priors = dict(vpvs=(1.4, 2.1), layers=(1, 20), vs=(2, 5), z=(0, 60), mohoest=None, rfnoise_corr=0.9,
                      swdnoise_corr=0., rfnoise_sigma=(1e-5, 0.05), swdnoise_sigma=(1e-5, 0.05))
        ip = dict(iter_burnin=burnin, iter_main=main_it, propdist=(0.015, 0.015, 0.015, 0.005, 0.005),
                  acceptance=(40, 100), thickmin=0.1, rcond=1e-5)
        pool = ChainPool(joint, initparams=ip, modelpriors=priors, seeds=np.arange(nchains) % 1000)
        t0 = time.perf_counter()
        pool.run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": nchains * (burnin + main_it) / dt, "unit": "chain iterations/s", "nchains": nchains,
                "iterations": burnin + main_it, "models_evaluated": int(pool.evaluated),
                "workload": "tutorial joint inversion (rdispph + prf), free vp/vs and noise, 2-21 layers",
                "host_seconds": {k: round(v, 4) for k, v in pool.seconds.items()}}
    except Exception as e:            # the sample must never take the benchmark line down
        return {"value": None, "error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='joint10', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='models per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true')
    ap.add_argument('--no-chain-pool', action='store_true')
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.workload)))
        return

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))

    # CPU baseline first, in a child that never initialises the GPU (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only',
                            '--workload', args.workload], capture_output=True, text=True)
        if r.returncode == 0:
            cpu = json.loads(r.stdout.strip().splitlines()[-1])
        else:
            sys.stderr.write('cpu baseline failed: %s\n' % r.stderr[-2000:])

    import torch
    import torch.distributed as dist
    from bayhunter_amd.distributed import max_over_ranks
    from bayhunter_amd.engine import ForwardEngine, RfSpec, SwdSpec

    # BH_DIST_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the N>1 code path on a
    # one-GPU box (ranks then share device 0); the driver runs one rank per GPU over RCCL ("nccl")
    backend = os.environ.get('BH_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend)

    B = args.batch or wl['B']
    per = np.linspace(1, 41, wl['P'])
    H, VP, VS, RHO, nl = make_models(wl, B, rank)
    swd_specs = [SwdSpec(r, per) for r in wl['refs']]
    rf_specs = [RfSpec('prf', np.linspace(-5, 35, 201))] if wl['rf'] else []
    eng_swd = ForwardEngine(swd=swd_specs, rf=rf_specs)            # the product path: one engine
    eng_only_swd = ForwardEngine(swd=swd_specs)                     # per-kernel timing only
    eng_only_rf = ForwardEngine(rf=rf_specs) if rf_specs else None
    if eng_only_rf:                                                 # same output row layout
        eng_only_rf._rfp[0].out_off = eng_swd._rfp[0].out_off
        eng_only_rf.row = eng_swd.row
    eng_only_swd.row = eng_swd.row
    dmodels = eng_swd.upload(H, VP, VS, RHO, nl)                       # inputs resident in HBM
    out, err = eng_swd.alloc_out(B)

    def step():
        # a sampler brings new models every step: the processing order (a sort by depth and S travel
        # time, engine.reorder) is part of the step, not of the upload
        eng_swd.run(eng_swd.reorder(dmodels.packed, dmodels.nlay), out=out, err=err)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, device='cuda' if backend == 'nccl' else 'cpu')

    # per-kernel durations for the roofline: the same launches, serialised on one stream, bracketed
    # by events on that stream (in the timed steps above rf_kernel overlaps the tail of swd_kernel)
    nk = min(args.steps, 5)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(nk)]
    eng_swd.overlap = False
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
    eng_swd.overlap = True
    ms_swd = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_rf = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if eng_only_rf else 0.0
    nerr = int(err.sum().item())

    if rank == 0:
        counts, Lmean = N_DLTAR[args.workload]
        bytes_swd, bytes_rf = algorithmic_bytes(wl, Lmean)
        flop_swd = sum(counts[r] * (Lmean - 1) * (F_RAYLEIGH if REF_TAGS[r][0] == 2 else F_LOVE)
                       for r in wl['refs'])
        flop_rf = (257 * (Lmean - 1) * 500 + Lmean * 300 + 257 * 60 + 5 * 512 * 9) if wl['rf'] else 0.0
        dom_is_swd = ms_swd >= ms_rf
        dom_ms = ms_swd if dom_is_swd else ms_rf
        dom_bytes = (bytes_swd if dom_is_swd else bytes_rf) * B
        dom_flop = (flop_swd if dom_is_swd else flop_rf) * B
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        from bayhunter_amd import _lib as bhlib
        dom_kernel = 'swd_kernel' if dom_is_swd else 'rf_kernel'
        pmc, pmc_note = measured_traffic(dom_kernel, args.workload, B, dom_ms, bhlib.loaded_hash())
        pmc = pmc or {}
        traffic = pmc.get('hbm_bytes')
        tflops = dom_flop / (dom_ms * 1e-3) / 1e12
        value = world * B * args.steps / dt
        res = {
            "metric": "forward evals/sec (SWD+RF, 10-layer)" if args.workload == 'joint10'
                      else "forward evals/sec (%s)" % args.workload,
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s%s, %s layers, %d periods, %d models/GPU/step"
                                   % (args.workload, '+'.join(wl['refs']),
                                      '+prf(201 samples, nsamp 512)' if wl['rf'] else '',
                                      str(wl['L']), wl['P'], B),
                       "models_per_gpu": B, "sharding": "models block-partitioned over ranks, no data-path collective",
                       "err_models": nerr, "library": bhlib.load().bh_version().decode()},
            # The dominant kernel is an fp64 scalar recurrence: FP64 vector issue is the roofline that
            # binds it, so that is the headline fraction; the HBM figure the contract asks
            # for is kept beside it.
            "roofline": {"kernel": dom_kernel, "bound": "fp64_valu",
                         "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "algorithmic_flop_per_launch": dom_flop,
                         "flop_per_eval_reference_path": flop_swd + flop_rf, "n_dltar_per_eval": counts,
                         "kernel_ms": dom_ms,
                         "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": dom_bytes,
                                 "traffic_bytes_per_launch_pmc": traffic},
                         "valu_pipes_busy_pmc": pmc.get('valu_busy'),
                         "lane_utilisation_pmc": pmc.get('lane_utilisation'),
                         "pmc_source": pmc_note,
                         "note": "achieved = reference-path flops (N_dltar x (L-1) x 190, FMA = 2) / kernel time, "
                                 "timed live with events on the launch stream; traffic and the *_pmc fields come "
                                 "from the committed rocprofv3 --pmc passes and are null unless that profile was "
                                 "taken with this very library and its kernel time matches this run's"},
            "kernels_ms": {"swd_kernel": ms_swd, "rf_kernel": ms_rf,
                           "note": "serialised; in the timed steps rf_kernel runs on a second stream and back-fills the tail of swd_kernel"},
            "cpu_baseline": cpu,
        }
        if world == 1 and args.workload == 'joint10' and not args.no_chain_pool:
            res["chain_pool"] = chain_pool_sample()
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
