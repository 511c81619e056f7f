"""Build libppyolo_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python pytorch-ppyolo_amd/ppyolo_hip/build.py [--force]

The .so is kept IN-TREE (ppyolo_hip/lib/) so it travels to the GPU box with the repo
snapshot; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libppyolo_hip.so')
SOURCES = ['capi.hip', 'conv_igemm.hip', 'conv_x3.hip', 'stem_pool.hip', 'dcn.hip', 'decode_nms.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-unused-function']
FLAGS += os.environ.get('PPY_EXTRA_HIPCC_FLAGS', '').split()      # experiments (-D...): part of the build stamp


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files.append(os.path.join(HERE, '..', '..', 'include', 'ppyolo_hip.h'))
    for f in files:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = LIB + '.sha256'
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    # one translation unit per process, all at once (the two conv files dominate the build time)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        jobs.append((cmd, obj, subprocess.Popen(cmd)))
    for cmd, obj, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [j[1] for j in jobs] + ['-o', LIB]
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    with open(stamp, 'w') as fh:
        fh.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
