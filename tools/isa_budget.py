"""Static instruction budget of a kernel: cross-compiles kernels.hip to gfx950 assembly (no GPU needed) and
prints, per basic block of the chosen kernel, the instruction counts by class -- fp64 VALU, other VALU, moves,
selects, LDS, SALU, waits.  Used for the per-phase budgets in DESIGN.md / profiles/r03_rf_instruction_budget.md
(dynamic totals come from the PMC pass, SQ_INSTS_VALU).

    python tools/isa_budget.py rf_kernelILb0 [min block size]      # mangled-name fragment
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayhunter_amd import _lib  # noqa: E402


def cls(op):
    if op.startswith('v_') and 'f64' in op:
        return 'v_f64'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'):
        return 'v_mov'
    if op.startswith('v_cndmask'):
        return 'v_sel'
    if op.startswith('v_'):
        return 'v_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'vmem'
    if op.startswith(('s_waitcnt', 's_nop')):
        return 's_wait'
    return 'salu' if op.startswith('s_') else 'other'


def main(frag, minsize=1):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'k.s')
        flags = [f for f in _lib.HIPCC_FLAGS if f not in ('-shared', '-fPIC')]
        subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['--cuda-device-only', '-S', 'kernels.hip', '-o', asm],
                       cwd=_lib.CSRC, check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*%s\w*:' % re.escape(frag), l))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    blocks, cur = [], dict(name='entry', note='', ins=[])
    blocks.append(cur)
    for l in lines[start + 1:end]:
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', l)
        if m:
            cur = dict(name=m.group(1), note=(m.group(2) or '').strip('; '), ins=[])
            blocks.append(cur)
        elif l.startswith('\t') and not l.strip().startswith((';', '.')):
            cur['ins'].append(l.split()[0])
    keys = ['v_f64', 'v_other', 'v_sel', 'v_mov', 'lds', 'vmem', 'salu', 's_wait']
    print('%-12s %5s  %s  %s' % ('block', 'instr', '  '.join('%7s' % k for k in keys), 'loop'))
    tot = collections.Counter()
    for b in blocks:
        c = collections.Counter(cls(o) for o in b['ins'])
        tot.update(c)
        if len(b['ins']) >= minsize:
            print('%-12s %5d  %s  %s' % (b['name'], len(b['ins']), '  '.join('%7d' % c[k] for k in keys), b['note'][:44]))
    print('%-12s %5d  %s' % ('total', sum(tot.values()), '  '.join('%7d' % tot[k] for k in keys)))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
