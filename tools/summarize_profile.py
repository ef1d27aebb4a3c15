"""Turn the raw rocprofv3 CSVs of tools/profile.sh into the committed summaries:
profiles/<tag>_rocprofv3_summary.md, profiles/<tag>_kernel_stats.csv, profiles/<tag>_traffic.json.

HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a coalesced
streaming read -> doubled."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main(tag, src=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = src or os.path.join(root, 'gpurun_out', 'prof_' + tag)
    dst = os.path.join(root, 'profiles')
    os.makedirs(dst, exist_ok=True)
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
    stats = newest(src + '/trace/*/*_kernel_stats.csv')
    shutil.copy(stats, os.path.join(dst, tag + '_kernel_stats.csv'))
    cfgf = os.path.join(src, 'config.txt')
    cfg = open(cfgf).read().strip() if os.path.exists(cfgf) else ''
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ('pmc_sq', 'pmc_lds', 'pmc_fetch', 'pmc_write'):
        f = glob.glob(src + '/' + p + '/*/*_counter_collection.csv')
        if not f:
            continue
        for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
            k = r['Kernel_Name'].split('(')[0]
            if 'bh::' in k:
                pmc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
    dur = {}
    for r in csv.DictReader(open(stats)):
        k = r['Name'].split('(')[0]
        if 'bh::' in k:
            dur[k] = float(r['AverageNs']) * 1e-6
    traffic = {}
    for k, d in avg.items():
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            # VALU pipes busy: SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; a dispatch
            # lasts GRBM_GUI_ACTIVE/8 cycles (the counter is summed over the 8 XCDs) on 1024 SIMDs
            busy = None
            if 'SQ_ACTIVE_INST_VALU' in d and 'GRBM_GUI_ACTIVE' in d:
                busy = d['SQ_ACTIVE_INST_VALU'] * 4.0 / (d['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
            traffic[k.replace('bh::', '')] = {
                'valu_busy': busy,
                'lane_utilisation': (d['SQ_THREAD_CYCLES_VALU'] / (64 * d['SQ_ACTIVE_INST_VALU'])
                                     if 'SQ_THREAD_CYCLES_VALU' in d else None),
                'read_bytes': 2.0 * d['FETCH_SIZE'] * 1024, 'write_bytes': d['WRITE_SIZE'] * 1024,
                'hbm_bytes': 2.0 * d['FETCH_SIZE'] * 1024 + d['WRITE_SIZE'] * 1024,
                'avg_ms_rocprof': dur.get(k)}
    # stamp: the code the counters belong to (bench.py drops them when its own library or its live
    # kernel time disagree)
    import subprocess
    hf = os.path.join(src, 'src_hash.txt')
    src_hash = open(hf).read().strip() if os.path.exists(hf) else ''
    try:
        sha = subprocess.run(['git', '-C', root, 'rev-parse', 'HEAD'], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(['git', '-C', root, 'status', '--porcelain', '--', 'bayhunter_amd', 'include'],
                                    capture_output=True, text=True).stdout.strip())
    except OSError:
        sha, dirty = '', False
    json.dump({'tag': tag, 'config': cfg, 'lib_src_hash': src_hash, 'git_sha': sha + ('+dirty' if dirty else ''),
               'kernels': traffic},
              open(os.path.join(dst, tag + '_traffic.json'), 'w'), indent=1)
    with open(os.path.join(dst, tag + '_rocprofv3_summary.md'), 'w') as out:
        out.write('# %s rocprofv3 summary (MI355X)\n\nbench config: `%s`\n\n' % (tag, cfg))
        out.write('Commands (tools/profile.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- '
                  'python3 bench.py ...` and separate `--pmc` passes (no trace flags) of the same command.\n\n')
        out.write('## kernel stats (rocprofv3 --stats)\n\n```\n')
        out.write(''.join(open(stats).readlines()[:3]))
        out.write('```\n\n## PMC, average per dispatch\n\n| kernel | counter | value |\n|---|---|---|\n')
        for k in sorted(avg):
            for c in sorted(avg[k]):
                out.write('| %s | %s | %.6g |\n' % (k, c, avg[k][c]))
        out.write('\n## derived\n\n')
        for k in sorted(avg):
            d = avg[k]
            if 'SQ_INSTS_VALU' in d:
                out.write('* %s: lane utilisation SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU) = %.3f; '
                          'issue share SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES = %.3f\n'
                          % (k, d['SQ_THREAD_CYCLES_VALU'] / (64 * d['SQ_ACTIVE_INST_VALU']),
                             d.get('SQ_ACTIVE_INST_ANY', float('nan')) / d['SQ_WAVE_CYCLES']))
        for k, t in traffic.items():
            if t.get('valu_busy') is not None:
                out.write('* %s: VALU pipes busy %.0f %% of the dispatch (SQ_ACTIVE_INST_VALU*4 / '
                          '(GRBM_GUI_ACTIVE/8 * 1024 SIMDs))\n' % (k, 100 * t['valu_busy']))
            out.write('* %s: HBM read %.1f MB (2 x FETCH_SIZE), written %.1f MB per launch\n'
                      % (k, t['read_bytes'] / 1e6, t['write_bytes'] / 1e6))
    print(open(os.path.join(dst, tag + '_rocprofv3_summary.md')).read())


if __name__ == '__main__':
    main(*sys.argv[1:])
