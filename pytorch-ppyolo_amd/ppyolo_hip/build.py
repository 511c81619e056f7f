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
SOURCES = ['capi.hip', 'conv_igemm.hip', 'conv_x3.hip', 'conv_stream.hip', 'conv_patch.hip', 'conv_ws.hip', 'conv_small.hip', 'conv_b2b.hip', 'conv_bwd.hip', 'train.hip', 'yolo_loss.hip', 'stem_pool.hip', 'dcn.hip', 'dcn_fused.hip', 'decode_nms.hip', 'preprocess.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-unused-function']
# No packed-fp32 VALU ops (v_pk_add/mul/fma_f32) in any kernel of this library.  Measured on MI355X (ROCm 7.2): while
# waves of the 16-bit-MFMA convolution kernels are resident on a CU, v_pk_*_f32 instructions of ANOTHER kernel's waves on
# that CU (two batches in flight on two streams) intermittently return a wrong result for one 16-lane pass -- the
# decode kernel's y0 came out as the box centre in 4-40 % of the steps, bit-exactly reproducible with
# tools/pk_hazard_probe.py, never with one stream, never with the scalar forms (0 mismatches in tools/lane_soak.py).
# The trigger (tools/probes/pk_hazard_asm.hip, profiles/r02_pk_hazard_trigger.txt): a packed op that takes the HIGH half of
# src1 for its LOW result (op_sel:[.,1]) reads zero there in lanes 48..63 while a 16-bit MFMA of another wave is in flight.
# The scalar forms are also not slower here (R50-608 bs8, two lanes: 1826 vs 1790 img/s).
# tests/test_capi_symbols.py::test_no_packed_fp32_ops_in_device_code disassembles the code objects of the shipped .so.
NO_PACKED_FP32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
if os.environ.get('PPY_ALLOW_PACKED_FP32', '0') != '1':          # (the probe's reproducer builds with packed ops)
    FLAGS += NO_PACKED_FP32
FLAGS += os.environ.get('PPY_EXTRA_HIPCC_FLAGS', '').split()      # experiments (-D...): part of the build stamp
_HOST_PASS_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"      # the x86 pass of hipcc


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files.append(os.path.join(HERE, '..', '..', 'include', 'ppyolo_hip.h'))
    for f in files:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True, out=None):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = LIB + '.sha256'
    dig = _digest()
    if out is None and not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
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
        jobs.append((cmd, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, universal_newlines=True)))
    for cmd, obj, proc in jobs:
        err = proc.communicate()[1]
        err = ''.join(ln for ln in err.splitlines(True) if _HOST_PASS_NOISE not in ln)
        if err.strip():
            sys.stderr.write(err)
        if proc.returncode != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [j[1] for j in jobs] + ['-o', out or LIB]
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    if out is not None:          # a variant for an experiment: the stamp keeps describing the product library
        return out
    with open(stamp, 'w') as fh:
        fh.write(dig)
    return LIB


def device_code_objects(path):
    """The gfx950 code objects embedded in an ELF -- a .o of this build (one offload bundle), the linked .so (one bundle per
    translation unit, concatenated in .hip_fatbin), or a vendor library such as librccl.so (ONE compressed bundle).  Bundle
    layout (clang-offload-bundler, uncompressed): 24-byte magic, u64 entry count, then per entry u64 offset, u64 size, u64
    triple length, triple.  Compressed bundles ('CCOB', u16 version, u16 method, v3: u64 total size, ...) are handed to
    clang-offload-bundler --unbundle, which knows the compression."""
    import struct
    import tempfile
    llvm = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    out = []
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, 'fb.bin')
        subprocess.check_call([os.path.join(llvm, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', path, fb])
        blob = open(fb, 'rb').read()
        pos = 0
        while blob[pos:pos + 4] == b'CCOB':      # compressed bundles, back to back from the start of the section
            ver, = struct.unpack_from('<H', blob, pos + 4)
            if ver >= 3:
                total, = struct.unpack_from('<Q', blob, pos + 8)
            else:
                total, = struct.unpack_from('<I', blob, pos + 8)
            one = os.path.join(td, 'ccob%d.bin' % len(out))
            with open(one, 'wb') as fh:
                fh.write(blob[pos:pos + total])
            bundler = os.path.join(llvm, 'clang-offload-bundler')
            listed = subprocess.check_output([bundler, '--list', '--type=o', '--input=' + one], universal_newlines=True).split()
            for tgt in listed:
                if 'gfx950' not in tgt:
                    continue
                co = os.path.join(td, 'co%d.bin' % len(out))
                subprocess.check_call([bundler, '--unbundle', '--type=o', '--input=' + one, '--targets=' + tgt, '--output=' + co])
                out.append(open(co, 'rb').read())
            os.remove(one)
            pos += total
            pos = (pos + 7) // 8 * 8
    pos = blob.find(magic)
    while pos >= 0:
        n, = struct.unpack_from('<Q', blob, pos + 24)
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'gfx950' in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(magic, pos + 1)
    return out


def packed_fp32_ops(path, detail=False):
    """Number of v_pk_{add,mul,fma}_f32 instructions in the gfx950 device code of `path` (a .o of this build, the .so, or a vendor
    library).  detail=True: dict(total, src1_high = those that route the HIGH half of src1 into the LOW result (`op_sel:[x,1..]`,
    the one form that misreads beside a 16-bit MFMA: profiles/r02_pk_hazard_trigger.txt), by_op, functions {name: count})."""
    import re
    import tempfile
    llvm = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
    info = dict(total=0, src1_high=0, by_op={}, functions={}, code_objects=0)
    pk = re.compile(r'\bv_pk_(add|mul|fma)_f32\b')
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(device_code_objects(path)):
            f = os.path.join(td, 'dev%d.co' % i)
            with open(f, 'wb') as fh:
                fh.write(co)
            info['code_objects'] += 1
            proc = subprocess.Popen([os.path.join(llvm, 'llvm-objdump'), '-d', f], stdout=subprocess.PIPE, universal_newlines=True)
            fn = '?'
            for ln in proc.stdout:          # (a vendor library disassembles to tens of millions of lines: stream it)
                if ln.endswith('>:\n'):
                    fn = ln.split('<', 1)[1][:-3]
                    continue
                m = pk.search(ln)
                if m is None:
                    continue
                code = ln.split('//')[0]
                info['total'] += 1
                info['by_op'][m.group(1)] = info['by_op'].get(m.group(1), 0) + 1
                info['functions'][fn] = info['functions'].get(fn, 0) + 1
                if re.search(r'op_sel:\[[01],1', code):
                    info['src1_high'] += 1
            proc.wait()
    return info if detail else info['total']


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
