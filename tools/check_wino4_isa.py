"""Build check of csrc/winograd_conv4.hip: the kernels name their accumulator AGPRs in inline asm, so the compiler must not put
anything of its own there. Compiles the file to ISA and fails if a compiler-generated v_accvgpr_write (VGPR source) targets an
accumulator register, or if a kernel uses scratch. Usage: python tools/check_wino4_isa.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'crb-active-3ddet_amd', 'csrc', 'winograd_conv4.hip')


def main(measure=False):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'w.s')
        cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '--cuda-device-only', '-S',
               SRC, '-o', out] + (['-DCRB_MEASURE'] if measure else [])
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad = []
    for m in re.finditer(r'^(_ZN\S*winograd4([bc]?)_kernel\S*):.*?s_endpgm', text, flags=re.S | re.M):
        name, body, second = m.group(1), m.group(0), m.group(2) == 'b'
        if 'ILi9E' in name:           # the stamp build (mode 9) may clobber accumulators: timing only
            continue
        first_acc = 16 if second else 0
        for w in re.finditer(r'v_accvgpr_write_b32 a(\d+), v\d+', body):
            if int(w.group(1)) >= first_acc:
                bad.append('%s: %s' % (name, w.group(0)))
        if re.search(r'\bscratch_(load|store)', body):
            bad.append('%s: scratch accesses' % name)
    if bad:
        print('\n'.join(bad[:20]))
        raise SystemExit('winograd_conv4.hip: the compiler touched accumulator registers (%d findings)' % len(bad))
    print('winograd_conv4.hip ISA check ok (%s)' % ('measure' if measure else 'product'))


if __name__ == '__main__':
    main('--measure' in sys.argv)
