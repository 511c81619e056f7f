"""Does RCCL's gfx950 device code contain the packed-fp32 instruction form that misreads beside a 16-bit MFMA?

The training step can run RCCL's all-reduce kernels (fp32 sums) on RCCL's stream WHILE the f16x2 convolution kernels of the
backward run -- the "foreign kernel beside 16-bit MFMA waves" condition.  The trigger named in round 2
(profiles/r02_pk_hazard_trigger.txt): a v_pk_{add,mul,fma}_f32 that routes the HIGH half of src1 into the LOW result
(`op_sel:[.,1]`) reads zero in lanes 48..63; the plain forms never failed.  librccl.so ships ONE compressed offload bundle
(ppyolo_hip.build.device_code_objects unbundles it); this script disassembles its gfx950 code object and counts both forms.

    python tools/rccl_pk_scan.py [/opt/rocm/lib/librccl.so] > profiles/r03_rccl_pk_scan.txt      (takes ~3 minutes, CPU only)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))


def main():
    from ppyolo_hip import build
    path = os.path.realpath(sys.argv[1] if len(sys.argv) > 1 else '/opt/rocm/lib/librccl.so')
    info = build.packed_fp32_ops(path, detail=True)
    print('%s (%d bytes): %d gfx950 code object(s)' % (path, os.path.getsize(path), info['code_objects']))
    print('v_pk_{add,mul,fma}_f32 instructions: %d  (%s)' % (info['total'], ', '.join('%s %d' % kv for kv in sorted(info['by_op'].items()))))
    print('... of which route src1.HIGH into the low result (op_sel:[.,1], the form that misreads beside a 16-bit MFMA): %d' % info['src1_high'])
    fam = {}
    for fn, n in info['functions'].items():
        key = 'other'
        for tag in ('FuncSum', 'FuncProd', 'FuncPreMulSum', 'FuncMinMax', 'FuncSumPostDiv', 'AllGather', 'Broadcast', 'SendRecv'):
            if tag in fn:
                key = tag
                break
        fam[key] = fam.get(key, 0) + n
    print('by reduction functor of the enclosing device function (%d functions): %s' % (len(info['functions']), ', '.join('%s %d' % kv for kv in sorted(fam.items()))))
    for fn, n in sorted(info['functions'].items(), key=lambda kv: -kv[1])[:6]:
        print('   %4d  %s' % (n, fn))
    print('this library (%s): %d packed fp32 instructions' % (os.path.relpath(build.LIB, ROOT), build.packed_fp32_ops(build.LIB)))


if __name__ == '__main__':
    main()
