"""Build libasr_hip.so (gfx950) in-tree with hipcc.

    python -m tensorflow_end2end_speech_recognition_amd.build [--force]

Each csrc/*.hip is compiled to an object in parallel and linked into
tensorflow_end2end_speech_recognition_amd/libasr_hip.so.  hipcc cross-compiles
without a GPU, so this runs in the build container; the .so travels to the GPU box.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(PKG, 'libasr_hip.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=%s' % ARCH, '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value']
# ctc.hip: the SLP vectoriser pairs the per-state multiplies of the single-wave recursion into v_pk_mul_f32 and then
# re-packs freshly loaded emission registers right behind their global_load (s_waitcnt vmcnt(0) every frame: the whole
# memory latency on the serial chain); without it the loads stay 8 frames ahead of their use
# lstm_cluster.hip: MFMA results in VGPRs (the default picks AGPR accumulators for the 4-wave kernels, whose gate math then
# starts with 12 v_accvgpr_read per step on the serial chain)
FILE_FLAGS = {'ctc.hip': ['-fno-slp-vectorize'], 'lstm_cluster.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}
# ASR_BUILD_ABLATE=1: the ablated instantiations of the headline recurrence kernels (scripts/probe_lstm_ablate.py); never
# part of a shipped library -- the default kernels' ISA is identical with and without it
if os.environ.get('ASR_BUILD_ABLATE') == '1':
    FILE_FLAGS['lstm_cluster.hip'] = FILE_FLAGS['lstm_cluster.hip'] + ['-DASR_LSTM_ABLATE']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, headers, force):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
    if force or _stale(obj, [src] + headers):
        cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, r.stderr[-4000:]))
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    headers = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + \
        sorted(glob.glob(os.path.join(PKG, '..', 'include', '*.h')))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, headers, force), srcs))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), '--offload-arch=%s' % ARCH, '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr[-4000:])
    if verbose:
        print('built %s (%d objects)' % (LIB, len(objs)))
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
