#!/usr/bin/env python
"""Device-ISA identity check for kernel experiments: compile a .hip file of two source trees (or two git revisions)
to gfx950 assembly and compare every kernel's instruction stream (labels and comments normalised).  Used to prove
that an experimental template variant behind a default-off switch leaves the default kernels' code untouched, so it
can be committed without a GPU run.

    python scripts/isa_diff.py <rev_a> <rev_b> [path/to/file.hip]      # e.g. HEAD~1 HEAD
Template arguments appended with a default (kernel<256, false> -> kernel<256, false, false>) are matched by trying the
old mangled name with `ELb0` inserted before the argument-list terminator."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = 'tensorflow_end2end_speech_recognition_amd/csrc/lstm_cluster.hip'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '--cuda-device-only', '-S']


def asm_of(rev, rel, tmp):
    tree = os.path.join(tmp, rev.replace('/', '_'))
    os.makedirs(tree)
    subprocess.run('git -C %s archive %s include %s | tar -x -C %s' % (ROOT, rev, os.path.dirname(rel), tree), shell=True,
                   check=True)
    out = os.path.join(tree, 'out.s')
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-I', os.path.join(tree, 'include'), '-o', out,
                                                     os.path.join(tree, rel)], check=True, capture_output=True)
    return open(out).read()


def functions(txt):
    parts = re.split(r'\n\t\.globl\t(\S+)\s*; -- Begin function \S+\n', txt)
    return {parts[i]: parts[i + 1][:parts[i + 1].find('; -- End function')] for i in range(1, len(parts), 2)}


def normalise(body, name):
    b = body.replace(name, 'FN')
    b = re.sub(r'\.Lfunc_end\d+', '.Lfunc_end', b)
    b = re.sub(r'\.LBB\d+_', '.LBB_', b)
    b = re.sub(r';.*', '', b)
    return [line.rstrip() for line in b.splitlines() if line.strip()]


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    rel = sys.argv[3] if len(sys.argv) > 3 else DEFAULT
    with tempfile.TemporaryDirectory() as tmp:
        a, b = functions(asm_of(sys.argv[1], rel, tmp)), functions(asm_of(sys.argv[2], rel, tmp))
    same = True
    for name in a:
        other = name if name in b else re.sub(r'(ELb[01])EEEv', r'\1ELb0EEEv', name)
        if other not in b:
            print('MISSING   %s' % name[:100])
            same = False
            continue
        ok = normalise(a[name], name) == normalise(b[other], other)
        same &= ok
        print('%s %s' % ('IDENTICAL' if ok else 'DIFFERENT', name[:100]))
    for name in b:
        if name not in a and re.sub(r'ELb0EEEv', 'EEEv', name) not in a:
            print('NEW       %s' % name[:100])
    print('default kernels unchanged' if same else 'DEFAULT KERNELS CHANGED')
    return 0 if same else 1


if __name__ == '__main__':
    sys.exit(main())
