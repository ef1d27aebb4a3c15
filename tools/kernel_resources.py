"""Registers and scratch of every kernel of kernels.hip as the product flags compile them (cross-compiled, no GPU).

    python tools/kernel_resources.py        -> one line per kernel
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayhunter_amd import _lib  # noqa: E402


def kernel_resources(source='kernels.hip'):
    """{demangled-ish kernel name: dict(vgpr, sgpr, scratch)} from the .amdhsa_kernel descriptors."""
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'k.s')
        flags = [f for f in _lib.HIPCC_FLAGS if f not in ('-shared', '-fPIC')]
        subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['--cuda-device-only', '-S', source, '-o', asm],
                       cwd=_lib.CSRC, check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S):
        field = lambda k: int(re.search(r'\.amdhsa_%s (\d+)' % k, m.group(2)).group(1))
        name = re.sub(r'^_ZN2bh\d+', '', m.group(1))
        name = re.sub(r'ENS_\d+\w+E$|EvNS_\d+\w+E$', '', name).replace('ILb0EE', '<false>').replace('ILb1EE', '<true>')
        out[name] = dict(vgpr=field('next_free_vgpr'), sgpr=field('next_free_sgpr'),
                         scratch=field('private_segment_fixed_size'))
    return out


if __name__ == '__main__':
    for k, v in kernel_resources().items():
        print('%-24s %3d VGPRs (allocated %3d)  %3d SGPRs  scratch %d B' % (k, v['vgpr'], -(-v['vgpr'] // 8) * 8, v['sgpr'], v['scratch']))
