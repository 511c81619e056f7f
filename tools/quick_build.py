#!/usr/bin/env python
"""Incremental build of libppyolo_hip.so for development: recompiles only the translation units whose own text (or any
csrc/*.h / include/ppyolo_hip.h) changed since the last quick build, relinks, and writes the same stamp ppyolo_hip.build
writes -- `__graft_entry__.build()` then sees an up-to-date library.  (build.build() itself always compiles all 14 units:
~5 minutes, the two big convolution files dominate.)   usage: python tools/quick_build.py [repo root]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
from ppyolo_hip import build as B          # noqa: E402


def h(paths, extra=''):
    d = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, 'rb') as fh:
            d.update(fh.read())
    return d.hexdigest()


def main():
    objdir = os.path.join(B.LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    cache_path = os.path.join(objdir, 'quick_build.json')
    cache = json.load(open(cache_path)) if os.path.exists(cache_path) else {}
    headers = sorted(os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith('.h'))
    headers.append(os.path.join(ROOT, 'include', 'ppyolo_hip.h'))
    hh = h(headers, ' '.join(B.FLAGS))
    jobs = []
    for src in B.SOURCES:
        path, obj = os.path.join(B.CSRC, src), os.path.join(objdir, src.replace('.hip', '.o'))
        key = h([path], hh)
        if cache.get(src) == key and os.path.exists(obj):
            continue
        cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + B.FLAGS + ['-c', path, '-o', obj]
        print('compile', src, flush=True)
        jobs.append((src, key, cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, universal_newlines=True)))
    for src, key, cmd, proc in jobs:
        err = ''.join(ln for ln in proc.communicate()[1].splitlines(True) if B._HOST_PASS_NOISE not in ln)
        if err.strip():
            sys.stderr.write(err)
        if proc.returncode != 0:
            raise SystemExit('%s failed' % src)
        cache[src] = key
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in B.SOURCES]
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', B.LIB])
    with open(B.LIB + '.sha256', 'w') as fh:
        fh.write(B._digest())
    json.dump(cache, open(cache_path, 'w'))
    print('linked', B.LIB, '(%d unit(s) recompiled)' % len(jobs))


if __name__ == '__main__':
    main()
