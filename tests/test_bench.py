"""bench.py's launch contract: `--gpus N` must mean N ranks.

CPU tier: the launcher (a parent that never touches torch or the GPU starts N fresh rank processes)
and the refusal of a process group whose size is not --gpus, through the rendezvous-only probe.
GPU tier: two ranks sharing the one GPU of the box over gloo (rehearsal of the N>1 path: barrier, MAX
over ranks, gather of the ranks' devices), the refusal to stack RCCL ranks on one device, and the
shape of the N=1 line (roofline for both kernels, the BASELINE configs)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout          # ONE line on stdout, the JSON, whatever the number of ranks
    return json.loads(lines[0])


def test_launcher_starts_one_process_per_rank():
    d = _line(_run(['--gpus', '3', '--probe-ranks']))
    assert d['n_gpus'] == 3 and d['ranks_seen'] == 3 and d['pids'] == 3
    assert d['local_ranks'] == [0, 1, 2] and d['launcher'] == 'bench.py'


def test_single_rank_needs_no_launcher():
    d = _line(_run(['--gpus', '1', '--probe-ranks']))
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['launcher'] is None


def test_world_size_must_equal_gpus():
    """Started by an outer launcher with the wrong number of ranks: refuse instead of reporting
    n_gpus = WORLD_SIZE under a --gpus N label."""
    r = _run(['--gpus', '4', '--probe-ranks'], env=dict(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_PORT='1'))
    assert r.returncode != 0 and '--gpus 4 but WORLD_SIZE=2' in r.stderr


def test_parent_of_the_ranks_never_imports_torch():
    """The launcher branch must run before anything that could initialise the GPU."""
    src = open(BENCH).read()
    main = src[src.index('def main():'):]
    assert main.index('launch_ranks(args)') < main.index('import torch')
    head = src[:src.index('def make_models')]
    assert 'import torch' not in head and 'bayhunter_amd' not in head.split('"""', 2)[2]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_over_gloo():
    d = _line(_run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '8192'],
                   env=dict(BH_DIST_BACKEND='gloo')))
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['launcher'] == 'bench.py'
    assert d['devices_seen'] == 1 and len(d['devices']) == 2          # rehearsal: both on the one GPU
    assert len(d['per_rank_ms_per_step']) == 2 and all(t > 0 for t in d['per_rank_ms_per_step'])
    assert d['config']['models_per_gpu'] == 8192 and d['scaling'] == 'weak'
    # whole-job value: both ranks' models over the slowest rank's time
    assert abs(d['value'] - 2 * 8192 * 2 / (d['ms_per_step'] * 2e-3)) <= 1e-6 * d['value']
    assert d['cpu_baseline'] is None and 'configs' not in d


@pytest.mark.gpu
def test_collective_branches_run_over_rccl_with_one_rank():
    """What an 8-GPU run does between the ranks -- RCCL rendezvous, gather of the ranks' devices, barrier with
    device_ids, MAX all-reduce and gather of the step times on the device -- executed for real by a one-rank
    `nccl` process group (the only RCCL configuration a one-GPU box can run)."""
    d = _line(_run(['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '8192', '--no-cpu-baseline', '--no-chain-pool'],
                   env=dict(BH_BENCH_FORCE_DIST='1')))
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['devices_seen'] == 1 and d['dist_backend'] == 'nccl'
    assert len(d['per_rank_ms_per_step']) == 1 and d['value'] > 0


@pytest.mark.gpu
def test_rccl_ranks_are_not_stacked_on_one_gpu():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('more than one GPU here')
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '4096'])
    assert r.returncode != 0 and 'one rank per GPU is required' in r.stderr


@pytest.mark.gpu
def test_single_gpu_line_has_both_rooflines_and_the_configs():
    d = _line(_run(['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-chain-pool']))
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['metric'] == 'forward evals/sec (SWD+RF, 10-layer)'
    for k in ('roofline', 'roofline_rf'):
        rl = d[k]
        assert rl['peak'] == 78.6 and 0 < rl['frac'] < 1 and rl['kernel_ms'] > 0 and 0 < rl['hbm']['frac'] < 1
    assert d['roofline_rf']['frequencies_computed'] == 213 and d['roofline_rf']['frequencies_total'] == 257
    assert sorted(d['configs']) == ['cfg2', 'cfg3', 'cfg4', 'cfg5']
    for c in d['configs'].values():
        assert c['value'] > 0 and c['ms_per_step'] > 0 and 0 < c['roofline_frac'] < 1 and c['err_models'] == 0
        assert c['kernel'].startswith('swd')
