#!/usr/bin/env python
"""Benchmark of the PP-YOLO inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (backbone + head + decode + Matrix-NMS, i.e.
`PPYOLO.forward` minus its final host list-building) over one batch of 8 synthetic images
per GPU that are already resident in HBM.  Metric: images/s of PPYOLO ResNet50-vd 608x608
bs=8 (BASELINE.json `metric`, configs[2]); weak scaling (8 images per GPU, one RCCL
all-gather of the detection records per step when N > 1).  Prints ONE JSON line on rank 0.

`roofline`: the dominant kernels are the fp32-MFMA implicit-GEMM convolutions.  After the
timed region the same plan is replayed eagerly with HIP events bracketing every conv / DCN
launch on the launch stream; achieved = algorithmic conv FLOPs per step (2*MAC, BN/act
excluded: 102.0 GFLOP per image at R50-608) / summed conv launch time, against
the 157.3 TFLOP/s fp32 MFMA peak.  `cpu_baseline`: the oracle (same PyTorch-CPU ops as the
reference) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 16 * FP32_MFMA_PEAK_TFLOPS     # same guide: fp32 MFMA = 1/16 of the dense bf16 MFMA rate (~2.5 PF)
# bf16x3 kernels (csrc/conv_x3.hip) execute 6 bf16 MFMA products per fp32 multiply-add: their MFMA roofline in
# units of ALGORITHMIC fp32 FLOPs is the bf16 peak / 6
X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
# f16x2 kernels: 3 fp16 MFMA products per multiply-add (fp16 MFMA rate = bf16 rate)
F16X2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
NUM_FP32_CFGS, NUM_X3_CFGS = 31, 9

WORKLOADS = {
    'r50vd_608': dict(cfg='PPYOLO_2x_Config', size=608, model='PPYOLO ResNet50-vd (DCNv2, CoordConv, SPP)'),
    'r18vd_416': dict(cfg='PPYOLO_r18vd_Config', size=416, model='PPYOLO_r18vd'),
    'r18vd_320': dict(cfg='PPYOLO_r18vd_Config', size=320, model='PPYOLO_r18vd'),
    # the other configurations the reference publishes demo FPS for (README.md:13-17), batch 1
    'r50vd_320': dict(cfg='PPYOLO_2x_Config', size=320, model='PPYOLO ResNet50-vd (DCNv2, CoordConv, SPP)'),
    'r18vd_608': dict(cfg='PPYOLO_r18vd_Config', size=608, model='PPYOLO_r18vd'),
}


def build_model(cfg_name, device):
    import config as C
    from model.ppyolo import PPYOLO
    from ppyolo_hip import synth
    cfg = getattr(C, cfg_name)()
    bb = C.select_backbone(cfg.backbone_type)(**cfg.backbone)
    hd = C.select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head)
    m = PPYOLO(bb, hd)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict(sd)
    m.eval()
    hd.set_dropblock(is_test=True)
    return m.to(device), sd, cfg


def model_from_sd(cfg_name, sd, device):
    import config as C
    from model.ppyolo import PPYOLO
    cfg = getattr(C, cfg_name)()
    bb = C.select_backbone(cfg.backbone_type)(**cfg.backbone)
    hd = C.select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head)
    m = PPYOLO(bb, hd)
    m.load_state_dict(sd)
    m.eval()
    hd.set_dropblock(is_test=True)
    return m.to(device)


def worst_case_leg(wl, dev, x, ims, sd, cfg, depth, seconds, cpu_threads):
    """Not `value`: the WHOLE step in SURVEY 8(d)'s second score regime -- the head as a default initialisation leaves it (output
    logits ~ N(0, 0.1)), so every one of the 1 819 440 (box, class) pairs of an image passes the 0.01 threshold; the reference
    sorts 1.8 M keys per image there (model/matrix_nms.py:120; BASELINE.md section 2: 1.32 img/s on 8 CPU threads).  Same
    network, same images; only the three output convolutions are rescaled (bias 0, weights x a factor measured on one forward so
    that the logits' std is 0.1).  The CPU oracle runs ONE batch with the same weights beside it."""
    from collections import OrderedDict
    sdw = OrderedDict((k, v.clone()) for k, v in sd.items())
    for k in sdw:
        if k.startswith('head.yolo_output_convs.') and k.endswith('.conv.bias'):
            sdw[k].zero_()
    model = model_from_sd(wl['cfg'], sdw, dev)
    ex = model._plans.executor(x)
    ex.set_inputs(x, ims)
    ex.run()
    torch.cuda.synchronize()
    for i, a in enumerate(ex.plan.head_outs):
        sdw['head.yolo_output_convs.%d.conv.weight' % i] *= 0.1 / float(ex.view(a).dense().std())
    del ex, model
    model = model_from_sd(wl['cfg'], sdw, dev)
    lanes = model.in_flight(depth).lanes(x)
    for e, _ in lanes:
        e.set_inputs(x, ims)

    def go(n, d):
        for i in range(n):
            e, st = lanes[i % d]
            with torch.cuda.stream(st):
                e.run()
        torch.cuda.synchronize()

    def rate(d):
        go(3 * d, d)
        n = 8
        while True:
            t0 = time.perf_counter()
            go(n, d)
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 1 << 14:
                return x.shape[0] * n / dt, dt / n * 1e3
            n *= 4
    v2, ms2 = rate(depth)
    v1, ms1 = rate(1) if depth > 1 else (v2, ms2)
    cands = int(lanes[0][0].cand_count.float().mean().item())
    out = dict(value=round(v2, 1), unit='images/s', ms_per_step=round(ms2, 3), one_batch_at_a_time=round(v1, 1),
               candidates_per_image=cands,
               regime='every (box, class) pair above the score threshold: output convolutions rescaled to logits ~ N(0, 0.1), bias 0 '
                      '(the reference\'s default initialisation); same images, same network otherwise')
    if cpu_threads:
        from oracle import ppyolo_oracle as orc
        torch.set_num_threads(cpu_threads)
        xc, ic = x.cpu(), ims.cpu()
        orc.ppyolo_forward(sdw, cfg, xc[:1], ic[:1])
        t0 = time.perf_counter()
        orc.ppyolo_forward(sdw, cfg, xc, ic)
        dt = time.perf_counter() - t0
        out['cpu_baseline'] = dict(value=round(x.shape[0] / dt, 3), unit='images/s', cores=cpu_threads, kind='port',
                                   sample='1 batch of %d images in the same regime (the oracle sorts all 1.8 M candidates per image like the reference)' % x.shape[0])
    return out


def pmc_traffic_leg(argv_tail, nconv, timeout=120, extra_kernels=()):
    """HBM-side bytes of the convolution launches, MEASURED BY THIS RUN: bench.py re-executes itself twice under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (counters need their own passes and their own process: the guide's
    HBM / rocprofv3 section) as a short eager one-lane child, and sums the counters of the convolution kernels per pass of the
    plan (= per stem launch).  gfx950: FETCH_SIZE reports half of a 16 B/lane streaming read (same section), hence 2 x FETCH +
    WRITE; both are KB.  Returns (bytes per conv launch, source dict) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    main_k = ('conv_igemm', 'conv1x1_stream', 'conv3x3_patch', 'conv_b2b', 'dcn_fused') + tuple(extra_kernels)
    tot = {}
    launches = 0.0
    t0 = time.perf_counter()
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='ppy_pmc_')
        cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
               sys.executable, os.path.abspath(__file__), '--pmc-child'] + argv_tail
        env = dict(os.environ, TMPDIR='/tmp')
        try:
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            files = glob.glob(os.path.join(d, '**', '*counter_collection*.csv'), recursive=True)
            val, passes, nl = 0.0, 0, 0
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get('Counter_Name') != counter:
                            continue
                        k = r['Kernel_Name']
                        if 'stem_conv' in k:
                            passes += 1
                        if any(n in k for n in main_k) or 'splitk_reduce' in k:
                            val += float(r['Counter_Value'])
                            nl += 0 if 'splitk_reduce' in k else 1
            if not passes:
                return None, 'no counter rows for %s' % counter
            tot[counter] = val / passes
            launches = nl / passes
        except Exception as exc:       # a profiler that is missing / hangs / fails must not take the benchmark line with it
            return None, '%s pass failed: %s' % (counter, type(exc).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # third pass: the matrix pipe's own busy count (BASELINE.json's metric names "conv MFMA util %"): SQ_VALU_MFMA_BUSY_CYCLES is 32
    # per 32x32x16 wave-instruction over all SIMDs, GRBM_GUI_ACTIVE the active cycles summed over the 8 XCDs -- at the clock the
    # board actually ran.  Optional: a failure here leaves the traffic figures standing.
    mfma = None
    d = tempfile.mkdtemp(prefix='ppy_pmc_')
    try:
        cmd = [exe, '--kernel-trace', '--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
               sys.executable, os.path.abspath(__file__), '--pmc-child'] + argv_tail
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=timeout, check=True)
        busy = active = 0.0
        for f in glob.glob(os.path.join(d, '**', '*counter_collection*.csv'), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r['Kernel_Name']
                    if any(n in k for n in main_k):
                        if r.get('Counter_Name') == 'SQ_VALU_MFMA_BUSY_CYCLES':
                            busy += float(r['Counter_Value'])
                        elif r.get('Counter_Name') == 'GRBM_GUI_ACTIVE':
                            active += float(r['Counter_Value'])
        if busy > 0 and active > 0:
            mfma = dict(frac=round(busy / (active / 8.0 * 1024.0), 4), unit='share of the matrix pipe\'s cycles with an MFMA in flight, all conv / DCN launches, time-weighted',
                        note='SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), rocprofv3 --pmc pass of this run; 3 MFMA products per fp32 '
                             'multiply-add in the f16x2 scheme, i.e. a third of the busy cycles is useful fp32 work')
    except Exception:
        mfma = None
    finally:
        shutil.rmtree(d, ignore_errors=True)
    byt = (2.0 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0
    if not nconv:
        nconv = max(1.0, launches)
    return round(byt / nconv), dict(launches_per_step=round(launches, 1), mfma_busy_pmc=mfma, measured_by='this run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of an eager one-lane child process',
                                    hbm_bytes_per_step=round(byt), fetch_size_kb_per_step=round(tot['FETCH_SIZE'], 1),
                                    write_size_kb_per_step=round(tot['WRITE_SIZE'], 1), gfx950_fetch_correction=2.0,
                                    seconds=round(time.perf_counter() - t0, 1), stale=False)


def traced_pass(ex, steps=3, warm=2):
    """The --trace-child loop: eager one-lane passes of the plan with a marker launch (ppy_debug_marker: grid = op index + 1
    workgroups) in front of every op, one more in front of decode + Matrix-NMS and one behind them -- what kernel_trace_leg cuts a
    `rocprofv3 --kernel-trace` into."""
    import ctypes
    from ppyolo_hip import _lib
    from ppyolo_hip.ops import _stream
    mark = _lib.lib().ppy_debug_marker
    mark.restype, mark.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]
    n = len(ex.plan.ops)
    for _ in range(warm + steps):
        ex.amax.zero_()
        for i, op in enumerate(ex.plan.ops):
            mark(i, _stream())
            ex._run_op(op)
        mark(n, _stream())
        if ex.plan.decode is not None:
            ex._run_decode()
        mark(n + 1, _stream())
        torch.cuda.synchronize()


def kernel_trace_leg(argv_tail, ex, per_op_flops, steps=3, timeout=180, table_path=None):
    """Solo kernel time of the plan's launches from a ONE-LANE, EAGER `rocprofv3 --kernel-trace` of this tree, taken by this run:
    bench.py re-executes itself as a short child (--trace-child: traced_pass above) under the profiler and cuts the dispatch list
    into plan ops at the marker launches.  A launch's time is End - Start of its kernels (split-K combine, a pooling launch that
    follows its producer: all kernels between two markers belong to the op in front) -- no host gaps, no event packets.  Returns
    (dict for the bench line, per-op rows) or (None, reason).  round-5 review, item 3."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from ppyolo_hip.engine import tune_key
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    d = tempfile.mkdtemp(prefix='ppy_ktrace_')
    t0 = time.perf_counter()
    try:
        cmd = [exe, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'kt', '--',
               sys.executable, os.path.abspath(__file__), '--trace-child'] + argv_tail
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=timeout, check=True)
        rows = []
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace*.csv'), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    g = r.get('Grid_Size_X') or r.get('Grid_Size') or '0'
                    w = r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or '64'
                    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(float(g)), int(float(w))))
        rows.sort()
    except Exception as exc:
        return None, 'kernel-trace pass failed: %s' % type(exc).__name__
    finally:
        shutil.rmtree(d, ignore_errors=True)
    nops = len(ex.plan.ops)
    # cut at the markers: a step starts at marker 0 and ends at marker nops + 1
    passes, cur, cur_id = [], None, None
    for s, e, name, grid, wg in rows:
        if 'ppy_marker_kernel' in name:
            mid = grid // max(1, wg) - 1 if grid >= wg else grid - 1      # (grid in work-items; some versions report workgroups)
            if mid == 0:
                cur = dict(start=s, ops={}, end=None)
            if cur is not None:
                cur_id = mid
                if mid == nops + 1:
                    cur['end'] = s
                    passes.append(cur)
                    cur, cur_id = None, None
            continue
        if cur is not None and cur_id is not None:
            ent = cur['ops'].setdefault(cur_id, [0.0, []])
            ent[0] += (e - s) * 1e-3          # us
            ent[1].append(name)
    if len(passes) < steps:
        return None, 'only %d complete passes in the trace' % len(passes)
    passes = passes[-steps:]
    convs = [i for i, op in enumerate(ex.plan.ops) if op['op'] in ('conv', 'dcn') and op.get('b2b_of') is None]
    us = {i: sum(p['ops'].get(i, [0.0])[0] for p in passes) / len(passes) for i in list(range(nops + 1))}
    conv_ms = sum(us[i] for i in convs) * 1e-3
    other_ms = sum(us[i] for i in range(nops) if i not in convs) * 1e-3
    span_ms = sum(p['end'] - p['start'] for p in passes) / len(passes) * 1e-6
    table = []
    for i, op in enumerate(ex.plan.ops):
        key = tune_key(op) if op['op'] in ('conv', 'dcn') else op['op']
        if op.get('b2b') is not None:
            key += ' + ' + tune_key(op['b2b']) + ' (one launch)'
        elif op.get('b2b_of') is not None:
            key += ' (fused into the launch in front of it)'
        names = passes[-1]['ops'].get(i, [0.0, []])[1]
        table.append(dict(i=i, key=key, cfg=op.get('cfg'), splitk=op.get('splitk'), us=round(us[i], 2), kernels=len(names),
                          gflop=round(per_op_flops[i] / 1e9, 3),
                          tflops=round(per_op_flops[i] / (us[i] * 1e-6) / 1e12, 1) if us[i] > 0 and per_op_flops[i] else 0.0))
    table.append(dict(i=nops, key='decode + matrix_nms', cfg=None, splitk=None, us=round(us[nops], 2),
                      kernels=len(passes[-1]['ops'].get(nops, [0.0, []])[1]), gflop=0.0, tflops=0.0))
    if table_path:
        with open(table_path, 'w') as fh:
            fh.write('# one lane, eager, rocprofv3 --kernel-trace of `bench.py --trace-child %s`: per plan op, mean over %d passes of the summed\n'
                     '# End - Start of its kernels (us); markers (ppy_marker_kernel) cut the dispatch list.  conv / DCN launches: %.3f ms,\n'
                     '# other plan ops %.3f ms, decode + Matrix-NMS %.3f ms; first marker -> last marker %.3f ms per pass (includes host gaps\n'
                     '# of the eager loop).\n' % (' '.join(argv_tail), len(passes), conv_ms, other_ms, us[nops] * 1e-3, span_ms))
            fh.write('%4s %-100s %5s %3s %9s %3s %8s %7s\n' % ('op', 'key', 'cfg', 'sk', 'us', 'n', 'GFLOP', 'TF/s'))
            for r in table:
                fh.write('%4d %-100s %5s %3s %9.2f %3d %8.3f %7.1f\n' % (r['i'], r['key'][:100], r['cfg'], r['splitk'], r['us'], r['kernels'],
                                                                         r['gflop'], r['tflops']))
    return dict(kernel_ms_per_step=round(conv_ms, 4), other_plan_ops_ms=round(other_ms, 4), decode_nms_ms=round(us[nops] * 1e-3, 4),
                passes=len(passes), seconds=round(time.perf_counter() - t0, 1),
                measured_by='this run: rocprofv3 --kernel-trace of an eager one-lane child (bench.py --trace-child), dispatches cut into '
                            'plan ops at marker launches; sum of End - Start over the conv / DCN ops\' kernels, mean of %d passes' % len(passes)), table


class _SmiSampler(object):
    """Socket power and shader clock from rocm-smi every ~0.25 s on a thread (diagnostic: --power-trace)."""

    def __init__(self):
        import threading
        self.samples, self._stop = [], False
        self._th = threading.Thread(target=self._poll, daemon=True)
        self._th.start()

    def _poll(self):
        import re
        import subprocess
        while not self._stop:
            try:
                o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r'Power \(W\): ([0-9.]+)', o)
                ck = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', o)
                if pw and ck:
                    self.samples.append((float(pw.group(1)), int(ck.group(1))))
            except Exception:
                pass
            time.sleep(0.25)

    def stop(self):
        self._stop = True
        self._th.join(timeout=10)
        if not self.samples:
            return dict(samples=0)
        sm = self.samples[1:] or self.samples        # (the first sample may precede the loop)
        return dict(samples=len(sm), mean_w=round(sum(p for p, _ in sm) / len(sm), 1), max_w=max(p for p, _ in sm),
                    mean_sclk_mhz=round(sum(c for _, c in sm) / len(sm)), min_sclk_mhz=min(c for _, c in sm))


def _tuned_path(mode):
    from ppyolo_hip import engine
    return engine._TUNED_PATHS[mode]


def conv_flops(plan):
    """Algorithmic FLOPs (2*MAC) of every conv-shaped launch of one step."""
    total = 0
    per_op = []
    for op in plan.ops:
        f = 0
        if op['op'] == 'conv':
            K, R, S, C = op['w'].shape
            y = op['y']
            Ho, Wo = (y.H // 2, y.W // 2) if op['ups'] else (y.H, y.W)
            f = 2 * y.N * Ho * Wo * K * R * S * C
        elif op['op'] == 'dcn':
            K, R, S, C = op['w'].shape
            y = op['y']
            f = 2 * y.N * y.H * y.W * K * R * S * C
        elif op['op'] == 'stem':
            y = op['y']
            f = 2 * y.N * y.H * y.W * op['w'].shape[0] * 27
        per_op.append(f)
        total += f
    for i, op in enumerate(plan.ops):          # conv2 -> conv3 as one launch (csrc/conv_b2b.hip): the work belongs to the launch that does it
        a = op.get('b2b_of') if op['op'] == 'conv' else None
        if a is not None:
            j = next(k for k, o in enumerate(plan.ops) if o is a)
            per_op[j] += per_op[i]
            per_op[i] = 0
    return total, per_op


def timed_conv_pass(ex, per_op_flops, reps=3):
    """Eager replay with HIP events around every MFMA conv / DCN launch (current stream =
    the launch stream).  Returns (sum of conv launch ms per step, flops covered)."""
    convs = [i for i, op in enumerate(ex.plan.ops) if op['op'] in ('conv', 'dcn') and op.get('b2b_of') is None]      # (a fused-away conv3 launches nothing)
    best = None
    for _ in range(reps):
        evs = []
        for i, op in enumerate(ex.plan.ops):
            if i in convs:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ex._run_op(op)
                e.record()
                evs.append((s, e))
            else:
                ex._run_op(op)
        torch.cuda.synchronize()
        per = [s.elapsed_time(e) for s, e in evs]
        ms = sum(per)
        if best is None or ms < best:
            best = ms
            timed_conv_pass.last_per_op = dict(zip(convs, per))      # (ms per conv / DCN op of the best pass: per-class tables)
    # MFMA-pipe time the same launches would need at peak: per launch flops / (peak of the kernel family it ran on)
    ideal_s = 0.0
    fam_flops = {'fp32': 0, 'bf16x3': 0, 'f16x2': 0}
    for i in convs:
        op = ex.plan.ops[i]
        cfg = op['cfg']
        if op['op'] == 'dcn':      # ids of the fused DCNv2 kernel: scheme * tiles + tile; < 0 = the library picks by operands
            from ppyolo_hip import ops as K
            fam = (K.dcnv2_scheme(cfg) if cfg >= 0 else
                   ('f16x2' if op.get('wf16') is not None else ('bf16x3' if op.get('w3') is not None else 'fp32')))
        else:
            fam = 'fp32' if cfg < NUM_FP32_CFGS else ('bf16x3' if cfg < NUM_FP32_CFGS + NUM_X3_CFGS else 'f16x2')
        peak = {'fp32': FP32_MFMA_PEAK_TFLOPS, 'bf16x3': X3_PEAK_TFLOPS, 'f16x2': F16X2_PEAK_TFLOPS}[fam]
        ideal_s += per_op_flops[i] / (peak * 1e12)
        fam_flops[fam] += per_op_flops[i]
    return best, sum(per_op_flops[i] for i in convs), len(convs), ideal_s, fam_flops


def batch_scaling_leg(model, wl, dev, ex8, seconds=0.6, batches=(8, 16, 32)):
    """Diagnostic, not `value`: ONE lane at batch 8 / 16 / 32 on the SAME kernels (a larger batch takes the tile configuration and
    split-K its layer has in the batch-8 table: only the grid grows), so that what a step loses to grid quantisation, launch tails
    and per-launch latency -- all of which shrink as 1 / batch -- is separated from the main loops' own efficiency, which does
    not.  Per batch: images/s of one lane (hipGraph replay), `frac` = conv FLOPs / solo conv kernel time / the mix's MFMA peak
    (as roofline.frac), `frac_one_lane` = conv FLOPs / whole-step time / that peak."""
    from ppyolo_hip import synth
    rows = []
    for bs in batches:
        try:
            x = synth.synth_images(bs, wl['size'], seed=4321 + bs).to(dev)
            ims = synth.synth_im_size(bs).to(dev)
            ex = model._plans.executor(x)
            borrowed = 0
            for op, src in zip(ex.plan.ops, ex8.plan.ops):
                if op['op'] in ('conv', 'dcn') and (bs != ex8.plan.N):
                    op['cfg'], op['splitk'] = src['cfg'], src['splitk']
                    borrowed += 1
            if borrowed:
                ex._size_workspace()
                ex._link_splits()
                ex.invalidate_graph()
            ex.set_inputs(x, ims)
            ex.use_graph = True
            for _ in range(3):
                ex.run()
            torch.cuda.synchronize()
            n = 8
            while True:
                t0 = time.perf_counter()
                for _ in range(n):
                    ex.run()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if dt >= seconds or n >= 4096:
                    break
                n *= 2
            ms_step = dt / n * 1e3
            ex.use_graph = False
            total_flops, per_op = conv_flops(ex.plan)
            conv_ms, covered, nconv, ideal_s, _ = timed_conv_pass(ex, per_op)
            ex.use_graph = True
            peak = covered / ideal_s / 1e12
            cls = {}
            for i, ms in timed_conv_pass.last_per_op.items():
                op = ex.plan.ops[i]
                c = 'dcn' if op['op'] == 'dcn' else ('3x3+1x1 fused' if op.get('b2b') is not None else ('3x3' if op['w'].shape[1] == 3 else '1x1'))
                d = cls.setdefault(c, [0.0, 0.0])
                d[0] += ms
                d[1] += per_op[i]
            rows.append(dict(batch=bs, tflops_by_class={c: round(d[1] / (d[0] * 1e-3) / 1e12, 1) for c, d in sorted(cls.items())}, one_lane_images_per_s=round(bs * n / dt, 1), ms_per_step=round(ms_step, 3),
                             conv_kernel_ms=round(conv_ms, 3), conv_tflops=round(covered / (conv_ms * 1e-3) / 1e12, 1),
                             frac=round(covered / (conv_ms * 1e-3) / 1e12 / peak, 4),
                             frac_one_lane=round(total_flops / (ms_step * 1e-3) / 1e12 / peak, 4),
                             tiles='batch-8 table' if bs == ex8.plan.N else 'the batch-8 entry of every layer (cfg, split-K) on a %dx grid' % (bs // ex8.plan.N)))
        except Exception as exc:          # a diagnostic must not take the benchmark line with it
            rows.append(dict(batch=bs, error='%s: %s' % (type(exc).__name__, str(exc)[:200])))
        finally:
            ex = x = None
            model._plans._ex = {k: v for k, v in model._plans._ex.items() if k[0] == ex8.plan.N}
            torch.cuda.empty_cache()
    return rows


def layer_report(ex, per_op, path):
    """Per-launch time (best of 5, HIP events) and achieved TFLOP/s, for kernel work."""
    from ppyolo_hip.engine import tune_key
    rows = []
    for i, op in enumerate(ex.plan.ops):
        best = None
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ex._run_op(op)
            e.record()
            e.synchronize()
            ms = s.elapsed_time(e)
            best = ms if best is None else min(best, ms)
        key = tune_key(op) if op['op'] in ('conv', 'dcn') else op['op']
        if op.get('b2b') is not None:
            key += ' + ' + tune_key(op['b2b']) + ' (one launch)'
        elif op.get('b2b_of') is not None:
            key += ' (fused into the launch in front of it)'
        rows.append(dict(i=i, key=key, cfg=op.get('cfg'), splitk=op.get('splitk'), ms=round(best, 4),
                         gflop=round(per_op[i] / 1e9, 3), tflops=round(per_op[i] / (best * 1e-3) / 1e12, 2)))
    with open(path, 'w') as fh:
        json.dump(rows, fh, indent=0)


def decode_nms_leg(ex, reps=5):
    """Secondary rooflines: the decode kernel against HBM bandwidth -- algorithmic bytes = the head outputs
    read once + boxes written once -- and the Matrix-NMS launches (latency-bound, reported as time), in BOTH score regimes
    of SURVEY 8d: "realistic" (the step's own head outputs: ~0.3 % of the (box, class) pairs pass the 0.01 threshold) and
    "worst case" (default-initialised head: EVERY pair passes, 1.8 M candidates per image -- where the reference spends
    1.43 s in aten::sort, model/matrix_nms.py:120).  A decode launch is ~20 us, less than the ~10 us a pair of HIP events
    around ONE launch adds: the realistic regime times `burst` back-to-back launches between two events (= the average
    launch duration rocprofv3 reports, profiles/*_kernel_trace_stats.txt)."""
    from ppyolo_hip import ops as K
    d = ex.plan.decode
    heads = [ex.view(a) for a in ex.plan.head_outs]
    byt = sum(h.N * h.H * h.W * h.C * 4 for h in heads) + ex.boxes.numel() * 4
    n = d['nms']

    def decode():
        K.yolo_decode_levels(heads, [lvl['anchors'] for lvl in d['levels']], [lvl['downsample'] for lvl in d['levels']],
                             d['num_classes'], d['scale_x_y'], d['iou_aware'], d['iou_aware_factor'], d['clip_bbox'],
                             ex.im_size, ex.boxes, n['score_threshold'], ex.cand_key, ex.cand_idx, ex.cand_count)

    def nms():
        K.matrix_nms(ex.boxes, d['num_classes'], ex.cand_key, ex.cand_idx, ex.cand_count, n['post_threshold'],
                     n['nms_top_k'], n['keep_top_k'], n['use_gaussian'], n['gaussian_sigma'], ex.out_dets, ex.out_count,
                     ex.out_keep, ex.nms_ws)

    def timed(fn, burst):
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(burst):
                fn()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / burst
            best = t if best is None else min(best, t)
        return best

    def regime(burst):
        ex.cand_count.zero_()
        decode()
        torch.cuda.synchronize()
        cands = int(ex.cand_count.float().mean().item())
        single = timed(lambda: (ex.cand_count.zero_(), decode()), 1)
        ex.cand_count.zero_()
        # (back-to-back launches append to the same lists: the candidate buffers hold M * C entries per image, `burst` realistic
        # lists fit; in the all-pass regime every launch fills them, so that one is timed launch by launch, memset included)
        dec = timed(decode, burst) if burst > 1 else single
        ex.cand_count.zero_()
        decode()
        t_nms = timed(nms, 1)
        return dec, single, t_nms, cands
    dec, single, t_nms, cands = regime(20)
    gbs = byt / (dec * 1e-3) / 1e9
    out = dict(decode=dict(bound='hbm', achieved=round(gbs, 1), peak=8000.0, unit='GB/s', frac=round(gbs / 8000.0, 4),
                           bytes_per_launch=byt, us_per_launch=round(dec * 1e3, 1), us_between_events_single_launch=round(single * 1e3, 1),
                           timing='mean of 20 back-to-back launches between two HIP events (best of %d); one launch between its own '
                                  'pair of events carries ~10 us of event overhead' % reps,
                           regime='realistic: %d candidates per image' % cands,
                           kernel='yolo_decode_stream_kernel<3, 80, iou_aware> (all head levels, one launch; registers + quad permutes, '
                                  'no LDS staging, no workgroup barrier)'),
               matrix_nms=dict(bound='latency', us_per_step=round(t_nms * 1e3, 1), candidates_per_image=cands,
                               kernels='nms_select / nms_colmax / nms_decay / nms_finish'))
    # ---- worst case: every (box, class) pair a candidate ----
    saved = [h.t.clone() for h in heads]
    g = torch.Generator(device=heads[0].t.device).manual_seed(7)
    for h in heads:
        h.t.copy_(torch.randn(h.t.shape, generator=g, device=h.t.device) * 0.1)      # logits ~ 0: score ~ 0.25 everywhere, all distinct
    try:
        dec_w, _, nms_w, cands_w = regime(1)
    finally:
        for h, sv in zip(heads, saved):
            h.t.copy_(sv)
        ex.cand_count.zero_()
        decode()
        nms()
        torch.cuda.synchronize()
    byt_w = byt + 8 * cands_w * ex.boxes.shape[0]                # + the candidate lists written (key + index per candidate)
    gbs_w = byt_w / (dec_w * 1e-3) / 1e9
    out['decode_worst_case'] = dict(bound='hbm', achieved=round(gbs_w, 1), peak=8000.0, unit='GB/s', frac=round(gbs_w / 8000.0, 4),
                                    bytes_per_launch=byt_w, us_per_launch=round(dec_w * 1e3, 1),
                                    regime='all %d (box, class) pairs per image above the score threshold (head logits N(0, 0.1)); bytes = head '
                                           'outputs + boxes + the candidate lists (8 B per candidate); a memset of the counters inside the timed span' % cands_w)
    out['matrix_nms_worst_case'] = dict(bound='latency', us_per_step=round(nms_w * 1e3, 1), candidates_per_image=cands_w,
                                        note='top-500 of 1.8 M candidates per image by radix select from global memory (the reference sorts all of them: '
                                             '1.43 s per batch on the CPU, model/matrix_nms.py:120)')
    return out


def hbm_conv_leg(ex, reps=5):
    """Secondary roofline of the HBM-bound convolution launches: the streaming 1x1 kernel's slowest launch of the
    plan against HBM bandwidth -- algorithmic bytes = activations + shortcut read once, output (+ its 2x2 average) written once."""
    from ppyolo_hip import ops as K
    first = K.stream_first_cfg()
    cands = [op for op in ex.plan.ops if op['op'] == 'conv' and first <= op['cfg'] < first + 2 and op.get('b2b_of') is None]
    best = None
    for op in cands:
        x, y = op['x'], op['y']
        M = y.N * y.H * y.W
        byt = 4 * (M * x.C + M * y.C * (2 if op['res'] is not None else 1) + (M // 4 * y.C if op.get('pool') is not None else 0))
        t = None
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ex._run_op(op)
            e.record()
            e.synchronize()
            ms = s.elapsed_time(e)
            t = ms if t is None else min(t, ms)
        if best is None or byt > best[0]:
            best = (byt, t, x, y, op)
    if best is None:
        return None
    byt, t, x, y, op = best
    gbs = byt / (t * 1e-3) / 1e9
    return dict(bound='hbm', achieved=round(gbs, 1), peak=8000.0, unit='GB/s', frac=round(gbs / 8000.0, 4), bytes_per_launch=byt,
                us_per_launch=round(t * 1e3, 1),
                kernel='conv1x1_stream_kernel: C%d -> K%d at %dx%d%s%s (largest launch of %d on this kernel)' % (
                    x.C, y.C, y.H, y.W, ' + shortcut' if op['res'] is not None else '', ' + pooled output' if op.get('pool') is not None else '',
                    len(cands)))


def b2b_leg(ex, reps=5):
    """conv2 -> conv3 of an identity bottleneck as one launch (csrc/conv_b2b.hip) against HBM bandwidth: algorithmic bytes = the
    pre-split input and the shortcut read once, the output (+ its 2x2 average) written once; the 64-channel intermediate moves nothing."""
    best = None
    for op in ex.plan.ops:
        b = op.get('b2b') if op['op'] == 'conv' else None
        if b is None:
            continue
        x, y = op['x'], b['y']
        M = y.N * y.H * y.W
        byt = 4 * (M * x.C + 2 * M * y.C + (M // 4 * y.C if b.get('pool') is not None else 0))
        t = None
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ex._run_op(op)
            e.record()
            e.synchronize()
            ms = s.elapsed_time(e)
            t = ms if t is None else min(t, ms)
        if best is None or byt > best[0]:
            best = (byt, t, x, y, b)
    if best is None:
        return None
    byt, t, x, y, b = best
    gbs = byt / (t * 1e-3) / 1e9
    flops = 2.0 * y.N * y.H * y.W * (x.C * 9 * x.C + x.C * y.C)
    return dict(bound='hbm', achieved=round(gbs, 1), peak=8000.0, unit='GB/s', frac=round(gbs / 8000.0, 4), bytes_per_launch=byt,
                us_per_launch=round(t * 1e3, 1), tflops=round(flops / (t * 1e-3) / 1e12, 1),
                kernel='conv_b2b_kernel: 3x3 C%d -> %d, then 1x1 -> K%d + shortcut at %dx%d%s (one launch between its own pair of events: ~4 us of '
                       'event overhead included)' % (x.C, x.C, y.C, y.H, y.W, ' + pooled output' if b.get('pool') is not None else ''))


def alt_math_leg(wl, dev, x, ims, steps, current, depth=1):
    """Not `value`: the same step with the other convolution math modes, measured in the same process right after the
    headline run (hipGraph replay, inputs resident): 'fp32' = every convolution on the exact-fp32 MFMA."""
    out = {}
    keep = os.environ.get('PPYOLO_HIP_MATH')
    for mode in ('f16x2', 'bf16x3', 'fp32'):
        if mode == current:
            continue
        os.environ['PPYOLO_HIP_MATH'] = mode
        try:
            model, _, _ = build_model(wl['cfg'], dev)
            lanes = model.in_flight(depth).lanes(x)
            for e, _ in lanes:
                e.set_inputs(x, ims)

            def go(n):
                for i in range(n):
                    e, st = lanes[i % depth]
                    with torch.cuda.stream(st):
                        e.run()
                torch.cuda.synchronize()
            go(3 * depth)
            n, dt = steps, 0.0
            while True:                           # short steps (r18): repeat until the timed region is >= 0.2 s
                t0 = time.perf_counter()
                go(n)
                dt = time.perf_counter() - t0
                if dt >= 0.2 or n >= 64 * steps:
                    break
                n *= 4
            out[mode] = round(x.shape[0] * n / dt, 1)
            del lanes, model
        finally:
            if keep is None:
                os.environ.pop('PPYOLO_HIP_MATH', None)
            else:
                os.environ['PPYOLO_HIP_MATH'] = keep
    out['unit'] = 'images/s'
    return out


def host_input_leg(ex, x, ims, steps, lanes=None):
    """Not `value`: the same step when the batch is handed over as a HOST buffer (pinned), i.e. with the PCIe copy
    inside the loop -- serialised on the launch stream, and overlapped (copy of batch i+1 on a second stream while
    batch i runs; the step then only pays a device-to-device move of the staged batch)."""
    dev = x.device
    pin = x.cpu().pin_memory()
    out = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ex.x_in.copy_(pin, non_blocking=True)
        ex.run()
    torch.cuda.synchronize()
    out['serial_images_per_s'] = round(x.shape[0] * steps / (time.perf_counter() - t0), 1)
    stage = [torch.empty_like(x), torch.empty_like(x)]
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    done = [torch.cuda.Event(), torch.cuda.Event()]
    cs = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    with torch.cuda.stream(cs):
        stage[0].copy_(pin, non_blocking=True)
        evs[0].record(cs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        nxt = (i + 1) & 1
        with torch.cuda.stream(cs):
            if i >= 1:
                cs.wait_event(done[nxt])          # the staged batch that used this buffer has been consumed
            stage[nxt].copy_(pin, non_blocking=True)
            evs[nxt].record(cs)
        main.wait_event(evs[i & 1])
        ex.x_in.copy_(stage[i & 1])
        done[i & 1].record(main)
        ex.run()
    torch.cuda.synchronize()
    out['overlapped_images_per_s'] = round(x.shape[0] * steps / (time.perf_counter() - t0), 1)
    if lanes is not None and len(lanes) > 1:
        # the lanes of `value`: each lane copies its next batch on its own stream, which overlaps the other lane's kernels
        hd = [torch.empty(e.out_dets.shape, dtype=torch.float32).pin_memory() for e, _ in lanes]
        hc = [torch.empty(e.out_count.shape, dtype=torch.int32).pin_memory() for e, _ in lanes]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % len(lanes)
            e, st = lanes[k]
            with torch.cuda.stream(st):
                e.x_in.copy_(pin, non_blocking=True)
                e.run()
                hd[k].copy_(e.out_dets, non_blocking=True)          # D2H of the padded detections (19.2 KB) + counts
                hc[k].copy_(e.out_count, non_blocking=True)
        torch.cuda.synchronize()
        out['in_flight_images_per_s'] = round(x.shape[0] * steps / (time.perf_counter() - t0), 1)
    out['note'] = ('input batch of %.1f MB per step from pinned host memory; not the headline value (inputs resident in '
                   'HBM); serial / overlapped = one lane, H2D only; in_flight = the lanes of `value`, H2D of the batch and D2H of the detections' % (x.numel() * 4 / 1e6))
    return out


def preprocess_leg(cfg, size, batch, lanes, steps, with_cpu):
    """Not `value` (the step in front of the path): the pre-processing kernel against HBM bandwidth
    -- algorithmic bytes = the raw uint8 images read once + the float NCHW batch written once -- and the whole job from
    RAW images (480x640x3 uint8 in pinned host memory -> H2D -> resize/normalise kernel -> forward) on the lanes of
    `value`; the numpy oracle of Decode.process_image on one host core beside it."""
    import numpy as np
    from ppyolo_hip import ops as K
    from ppyolo_hip.preprocess import Preprocessor
    dev = lanes[0][0].device
    pre = Preprocessor(cfg, size, dev)
    rng = np.random.RandomState(7)
    raw = [torch.from_numpy(rng.randint(0, 256, size=(480, 640, 3)).astype(np.uint8)).pin_memory() for _ in range(batch)]
    on_dev = [r.to(dev) for r in raw]
    out = torch.empty((batch, 3, size, size), dtype=torch.float32, device=dev)
    # a launch is ~15 us of kernel behind ~40 us of host work (ctypes marshalling of four per-image arrays): one launch between two
    # events measures the host (rounds 1-4 quoted 56 us that way).  As for the decode launch: 20 launches captured into a graph,
    # replayed between two events.
    K.preprocess_images(on_dev, size, pre.lut, out, swap_rb=pre.to_rgb)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            K.preprocess_images(on_dev, size, pre.lut, out, swap_rb=pre.to_rgb)
    g.replay()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 20.0
        best = ms if best is None else min(best, ms)
    g.reset()
    byt = sum(r.numel() for r in raw) + out.numel() * 4
    gbs = byt / (best * 1e-3) / 1e9
    res = dict(bound='hbm', achieved=round(gbs, 1), peak=8000.0, unit='GB/s', frac=round(gbs / 8000.0, 4),
               bytes_per_launch=byt, us_per_launch=round(best * 1e3, 1),
               kernel='preprocess_tile_kernel (BGR->RGB, 8-bit bicubic resize 480x640 -> %dx%d, normalise, HWC->CHW; %d images per '
                      'launch; mean of 20 launches replayed from a graph)' % (size, size, batch))
    ims = torch.tensor([[480., 640.]] * batch, device=dev)
    stage = [[torch.empty_like(o) for o in on_dev] for _ in lanes]

    def go(n):
        for i in range(n):
            e, st = lanes[i % len(lanes)]
            with torch.cuda.stream(st):
                for d, r in zip(stage[i % len(lanes)], raw):
                    d.copy_(r, non_blocking=True)
                K.preprocess_images(stage[i % len(lanes)], size, pre.lut, e.x_in, swap_rb=pre.to_rgb)
                e.im_size.copy_(ims)
                e.run()
        torch.cuda.synchronize()
    go(2 * len(lanes))
    t0 = time.perf_counter()
    go(steps)
    res['from_raw_host_images_per_s'] = round(batch * steps / (time.perf_counter() - t0), 1)
    if with_cpu:
        from oracle import preprocess_oracle as orc
        img = raw[0].numpy()
        orc.process_image(img, size)
        t0 = time.perf_counter()
        for _ in range(3):
            orc.process_image(img, size)
        res['cpu_oracle_images_per_s'] = round(3 / (time.perf_counter() - t0), 1)
        res['cpu_oracle_note'] = 'numpy restatement of Decode.process_image (cv2 is not installed), 1 core'
    return res


def parity_note_leg(model, wl, batch, x):
    """The headline parity figure in the bench line (round-5 review, item 7): this workload's batch against the rows the REFERENCE
    ITSELF produced for it (tests/golden/g18_*.npz, made by tools/make_goldens.py importing the reference: im_size (480, 640), the
    same images and weights as `value`) -- how many of the kept boxes lie beyond the north star's literal 1e-3 px, for this HIP
    path and for the reference against itself (its own runs with another thread count / conv backend / one image at a time).
    Data only: no oracle and no reference code runs here."""
    import numpy as np
    tag = {('PPYOLO_2x_Config', 608): 'r50vd_608', ('PPYOLO_r18vd_Config', 416): 'r18vd_416'}.get((wl['cfg'], wl['size']))
    path = os.path.join(ROOT, 'tests', 'golden', 'g18_%s.npz' % tag) if tag else None
    if not path or not os.path.exists(path) or batch != 8:
        return None
    g = np.load(path)
    if int(g['meta'][0]) != wl['size'] or int(g['meta'][1]) != batch:
        return None
    runs = [str(r) for r in g['runs']]
    alts = [r for r in runs if r not in ('t8', 'f64')]
    ims = torch.from_numpy(g['im_size_a']).float()
    dets, cnt, keep = model.forward_padded(x, ims.to(x.device))
    torch.cuda.synchronize()
    dets, cnt, keep = dets.cpu().double(), cnt.cpu(), keep.cpu().numpy()

    def dist(rows_a, keep_a, rows_b, keep_b):       # rows matched by keep index (= candidate id: the same detection)
        pos = {int(k): j for j, k in enumerate(keep_b)}
        ia = [i for i, k in enumerate(keep_a) if int(k) in pos]
        ib = [pos[int(keep_a[i])] for i in ia]
        A, B = rows_a[ia], rows_b[ib]
        return (A[:, 2:] - B[:, 2:]).abs().max(dim=1).values, (A[:, 1] - B[:, 1]).abs().max(), len(keep_b) - len(ia), ia == ib

    hip_beyond = ref_beyond = boxes = unmatched = hip_moved = 0
    ref_moved = {r: 0 for r in alts}
    hip_max = ref_max = score_max = 0.0
    same_order = True
    for i in range(batch):
        ref, rkeep = torch.from_numpy(g['t8_a_pred%d' % i]).double(), g['t8_a_keep%d' % i]
        kk = int(cnt[i])
        d, es, un, so = dist(dets[i, :kk], keep[i, :kk], ref, rkeep)
        hip_beyond += int((d > 1e-3).sum()); boxes += int(ref.shape[0]); unmatched += un + abs(kk - ref.shape[0])
        hip_max, score_max, same_order = max(hip_max, float(d.max())), max(score_max, float(es)), same_order and so
        if kk == ref.shape[0]:
            hip_moved += int((keep[i, :kk] != rkeep).sum())
        worst = 0
        for r in alts:
            ak = g['%s_a_keep%d' % (r, i)]
            dr = dist(torch.from_numpy(g['%s_a_pred%d' % (r, i)]).double(), ak, ref, rkeep)[0]
            worst = max(worst, int((dr > 1e-3).sum()))
            ref_max = max(ref_max, float(dr.max()))
            if ak.shape == rkeep.shape:
                ref_moved[r] += int((ak != rkeep).sum())
        ref_beyond += worst
    return dict(text='boxes beyond 1e-3 px: HIP %d / reference-vs-itself %d of %d (max %.2e / %.2e px); scores max %.1e; keep indices %s'
                     % (hip_beyond, ref_beyond, boxes, hip_max, ref_max, score_max,
                        'identical, same order' if (unmatched == 0 and same_order) else (
                            'identical set, %d of %d rows trade places between near-tied scores (the reference against itself: up to %d)'
                            % (hip_moved, boxes, max(ref_moved.values())) if unmatched == 0 else '%d differ' % unmatched)),
                hip_boxes_beyond_1e_3_px=hip_beyond, reference_vs_itself_boxes_beyond_1e_3_px=ref_beyond, boxes=boxes,
                hip_max_box_err_px=round(hip_max, 6), reference_vs_itself_max_px=round(ref_max, 6), max_score_err=float('%.2e' % score_max),
                keep_indices_identical=unmatched == 0, same_order=bool(same_order), rows_out_of_place=hip_moved,
                reference_vs_itself_rows_out_of_place=ref_moved,
                against='tests/golden/g18_%s.npz: rows + keep indices made by the reference itself (8 threads), im_size (480, 640); '
                        'reference-vs-itself = per image the worst of its other fp32 runs (%s), summed' % (tag, ', '.join(alts)),
                north_star='box coords <= 1e-3, scores <= 1e-4, keep indices bit-exact')


def cpu_baseline(sd, cfg, size, batch):
    """Oracle (PyTorch-CPU restatement of the reference forward) on the host cores."""
    from oracle import ppyolo_oracle as orc
    from ppyolo_hip import synth
    cores = os.cpu_count() or 1
    x = synth.synth_images(batch, size)
    ims = synth.synth_im_size(batch)
    # torch's intra-op pool does not scale to every core of a big host (256 threads ran 20x
    # SLOWER than 8 here); pick the best thread count on a 2-image probe, then time the batch.
    best_t, best_dt = None, None
    for t in sorted(set(min(cores, c) for c in (8, 16, 32, 64))):
        torch.set_num_threads(t)
        orc.ppyolo_forward(sd, cfg, x[:1], ims[:1])        # warm-up (thread pool, MKLDNN primitives)
        t0 = time.perf_counter()
        orc.ppyolo_forward(sd, cfg, x[:2], ims[:2])
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = t, dt
    torch.set_num_threads(best_t)
    n_batches = 1 if best_dt * batch / 2 > 10.0 else 2
    t0 = time.perf_counter()
    for _ in range(n_batches):
        orc.ppyolo_forward(sd, cfg, x, ims)
    dt = time.perf_counter() - t0
    return dict(value=round(n_batches * batch / dt, 3), unit='images/s', cores=best_t, kind='port',
                sample='%d batch(es) of %d images, %dx%d; oracle = same ATen ops as the reference forward; '
                       'thread count = best of {8,16,32,64} on a 2-image probe (host has %d cores)'
                       % (n_batches, batch, size, size, cores))


def train_cpu_baseline(sd, cfg, x, gt, targets):
    """The training oracle (oracle/train_oracle.py: the reference's step in the same ATen ops + torch autograd, bit-equal to
    the reference on the build box) on the host cores: ONE step of the same workload (forward of the whole network in
    training mode, YOLOv3Loss, backward through the trainable part)."""
    from oracle import train_oracle as trn
    cores = os.cpu_count() or 1
    threads = min(cores, 16)                 # (the inference baseline's probe: 16 threads beat 8, 32 and 64 on the 2x64-core host)
    torch.set_num_threads(threads)
    xs, gs, ts = x.cpu(), gt.cpu(), [t.cpu() for t in targets]
    trn.train_step(sd, cfg, xs[:1], gs[:1], [t[:1] for t in ts], rng_seed=1)      # warm-up
    t0 = time.perf_counter()
    trn.train_step(sd, cfg, xs, gs, ts, rng_seed=1)
    dt = time.perf_counter() - t0
    return dict(value=round(xs.shape[0] / dt, 3), unit='images/s', cores=threads, kind='port',
                sample='1 training step of %d images %dx%d (training-mode forward, loss, backward; SGD not included); oracle = same '
                       'ATen ops as the reference step; %d threads (host has %d cores)' % (xs.shape[0], xs.shape[2], xs.shape[3], threads, cores))


def _free_port():
    import socket
    so = socket.socket()
    so.bind(('127.0.0.1', 0))
    port = so.getsockname()[1]
    so.close()
    return port


def launcher_argv(gpus, argv, port=None):
    """`python bench.py --gpus N` from a bare shell (no torchrun environment): the command this process replaces itself
    with -- one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr',
            '127.0.0.1', '--master-port', str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def check_world(gpus, env):
    """--gpus must equal the number of ranks actually running; returns WORLD_SIZE or raises SystemExit."""
    world = int(env.get('WORLD_SIZE', '1'))
    if world != gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with `python bench.py --gpus %d`, which starts the '
                         'ranks itself, or torch.distributed.run --nproc-per-node %d)' % (gpus, world, gpus, gpus))
    return world


def train_bench(a, wl, dev, rank, world):
    """BASELINE config 5: the TRAINING step (reference train.py:416-443) of the same network, 8 images per GPU, data parallel:
    training-mode forward, YOLOv3Loss, backward through the head, ONE RCCL all-reduce of all gradients, SGD
    (ppyolo_hip/train.py).  Prints its own JSON line (rank 0)."""
    import numpy as np
    from ppyolo_hip import synth
    from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
    from ppyolo_hip.train import TrainStep, lr_at
    model, sd, cfg = build_model(wl['cfg'], dev)
    if a.freeze_at is not None:
        cfg.backbone['freeze_at'] = a.freeze_at
    S, hc = wl['size'], None
    hc = cfg.head
    x = synth.synth_images(a.batch, S, seed=1234 + rank + a.seed_offset).to(dev)
    bb, cc, ss = synth_ground_truth(a.batch, 50 + rank + a.seed_offset)
    targets = [torch.from_numpy(t).to(dev) for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)]
    gt = torch.from_numpy(bb).to(dev)
    ts = TrainStep(model, cfg, world, seed_rank=rank + a.seed_offset)      # (DropBlock draws: rank r == a single run with --seed-offset r)
    lr = lr_at(4000, cfg)
    losses = []
    if a.autotune:          # measure the convolution geometries the tables do not know (-> ppyolo_hip/tuned_gfx950_train.json)
        got = ts.autotune(x, gt, targets, a.save_tuning if rank == 0 else None)
        if rank == 0:
            print('train autotune: %d geometries measured' % len(got), file=sys.stderr)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    if a.pmc_child:          # profiled by pmc_traffic_leg: a few eager steps, nothing else
        for _ in range(3):
            ts.step(x, gt, targets, lr)          # (plain steps: the counters are per kernel, not per overlap)
        torch.cuda.synchronize()
        return
    # Round 5: the loop hands step() the NEXT batch as well (a loader has it): with the backbone frozen (freeze_at = 5) its
    # training-mode forward runs on a third stream beside this batch's head / loss / backward (TrainStep.prefetch_backbone) -- every
    # step still computes everything, results are bit-identical to the plain loop (tests/test_gpu_train_step.py).  --no-prefetch:
    # the plain loop, which is also timed afterwards as `one_step_at_a_time`.
    pre = not a.no_prefetch

    def loop(n, prefetch):
        for _ in range(n):
            losses.append(ts.step(x, gt, targets, lr, next_x=x if prefetch else None))
    loop(max(1, a.warmup), pre)
    barrier()
    t0 = time.perf_counter()
    loop(a.steps, pre)
    barrier()
    dt = time.perf_counter() - t0
    if a.dump_dets and rank == 0:                       # tests: the averaged gradients of the last timed step
        torch.cuda.synchronize()
        np.save(a.dump_dets, ts.gflat.cpu().numpy())
    dt_plain = None
    if pre and world == 1 and not a.dump_dets:
        ts._pref = None
        loop(2, False)
        barrier()
        t1 = time.perf_counter()
        loop(a.steps, False)
        barrier()
        dt_plain = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        tot = [float(l.sum()) for l in losses]
        flops = ts.flops
        ach = flops / (dt / a.steps) / 1e12
        out = dict(metric='images/sec %s %dx%d TRAIN step, %d images per GPU' % (wl['model'], S, S, a.batch),
                   value=round(world * a.batch * a.steps / dt, 2), unit='images/s', n_gpus=world, steps=a.steps, warmup=max(1, a.warmup),
                   ms_per_step=round(dt / a.steps * 1e3, 3), higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32',
                   data='synthetic (randn images, synthetic boxes through Gt2YoloTarget, deterministic random weights)',
                   config=dict(workload='%s %dx%d training step (reference train.py:416-443, freeze_at=%d), %d images per '
                                        'GPU' % (wl['model'], S, S, ts.freeze_at, a.batch), global_batch=world * a.batch,
                               parallelism=('data parallel x%d, %.1f MB of gradients per step averaged in %d collectives started during the backward '
                                            '(one per finished bucket range)' % (world, ts.gflat.numel() * 4 / 1e6,
                                                                                  sum(len(v['ranges']) for v in ts._buckets.values()) if ts._buckets else 1))
                               if world > 1 else 'single GPU',
                               math=('f16x2 (two fp16 terms after power-of-two scaling by tracked maxima, 3 MFMA products, fp32 accumulate) for the '
                                     'forward convolutions, the data and the weight gradients' if ts.f16 else
                                     'bf16x3 (exact 3-term bf16 split, 6 products) for every convolution and gradient'), eager=True),
                   loss_first=round(tot[0], 4), loss_last=round(tot[min(len(tot), max(1, a.warmup) + a.steps) - 1], 4),
                   pipelining=('the next batch\'s frozen-backbone forward runs beside this batch\'s head / loss / backward (step(next_x=)); '
                               'bit-identical to the plain loop' if (pre and ts.freeze_at == 5) else 'none'),
                   roofline=dict(bound='mfma', achieved=round(ach, 2), peak=round(F16X2_PEAK_TFLOPS if ts.f16 else X3_PEAK_TFLOPS, 1), unit='TFLOP/s',
                                 frac=round(ach / (F16X2_PEAK_TFLOPS if ts.f16 else X3_PEAK_TFLOPS), 4), traffic=None, flops_per_step=flops,
                                 kernel='conv_igemm_x3_kernel<*> forward + dgrad, conv_wgrad_x3_kernel<F16> (%s)' % ('f16x2' if ts.f16 else 'bf16x3'),
                                 peak_note='achieved = algorithmic convolution FLOPs of forward + backward / WHOLE step time (BatchNorm, loss, '
                                           'SGD, EMA and launch gaps included: the step is not graph-captured); peak = dense 16-bit MFMA / %d '
                                           'products per multiply-add' % (3 if ts.f16 else 6)))
        if dt_plain is not None:
            out['one_step_at_a_time'] = dict(value=round(world * a.batch * a.steps / dt_plain, 2), unit='images/s', ms_per_step=round(dt_plain / a.steps * 1e3, 3),
                                             note='the plain loop: step(x) without the next batch (what the reference\'s loop does, train.py:416-443)')
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = train_cpu_baseline(sd, cfg, x, gt, targets)
        # `value_fp32_exact` (round 5): the same step with every convolution, data gradient and weight gradient on the exact-fp32 MFMA
        # -- its own process, because the kernel choices are read once per process
        if world == 1 and not a.no_alt_math and os.environ.get('PPYOLO_HIP_TRAIN_MATH', 'f16x2') != 'fp32':
            import subprocess
            env = dict(os.environ, PPYOLO_HIP_TRAIN_MATH='fp32', PPY_WGRAD_FP32='1', PPY_DGRAD_FP32='1')
            cmd = [sys.executable, os.path.abspath(__file__), '--train', '--steps', str(min(a.steps, 10)), '--warmup', '2', '--no-cpu-baseline',
                   '--no-alt-math', '--no-pmc', '--workload', a.workload, '--batch', str(a.batch)] + (['--freeze-at', str(a.freeze_at)] if a.freeze_at is not None else [])
            try:
                r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True, universal_newlines=True)
                child = json.loads(r.stdout.strip().splitlines()[-1])
                out['value_fp32_exact'] = child['value']
                out['value_fp32_exact_note'] = ('same step, PPYOLO_HIP_TRAIN_MATH=fp32 PPY_WGRAD_FP32=1 PPY_DGRAD_FP32=1: forward, data and weight gradients on '
                                                'v_mfma_f32_32x32x2_f32 (%.2f ms per step, loss %s -> %s); `value` builds every product from 2-term fp16 splits'
                                                % (child['ms_per_step'], child.get('loss_first'), child.get('loss_last')))
            except Exception as exc:
                out['value_fp32_exact'] = None
                out['value_fp32_exact_note'] = 'the exact-fp32 child run failed: %s' % type(exc).__name__
        # HBM-side bytes of the convolution launches (forward, data gradient, weight gradient), MEASURED BY THIS RUN like the inference
        # line's: two rocprofv3 --pmc child passes of three eager steps; the committed summary only as a fallback, marked stale
        measured = None
        if world == 1 and not a.no_pmc and a.workload == 'r50vd_608':
            measured, src = pmc_traffic_leg(['--train', '--workload', a.workload, '--batch', str(a.batch)], None, timeout=300,
                                            extra_kernels=('conv_wgrad',))
            if measured is not None:
                src.pop('mfma_busy_pmc', None)
                out['roofline']['traffic'] = measured
                out['roofline']['traffic_unit'] = 'bytes per launch (mean over the convolution launches of a step: forward, dgrad, wgrad; PMC 2*FETCH_SIZE+WRITE_SIZE)'
                out['roofline']['traffic_source'] = src
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_train_pmc_traffic.json')))
        if measured is None and files and a.workload == 'r50vd_608' and a.batch == 8 and ts.freeze_at == 5:
            with open(files[-1]) as fh:
                rec = json.load(fh)
            out['roofline']['traffic'] = round(rec['hbm_bytes_per_step'] / max(1, rec['conv_launches_per_step']))
            out['roofline']['traffic_unit'] = 'bytes per launch (mean over the convolution launches of a step: forward, dgrad, wgrad; PMC 2*FETCH_SIZE+WRITE_SIZE)'
            out['roofline']['traffic_source'] = dict(file='profiles/' + os.path.basename(files[-1]), measured=rec.get('measured'), stale=True,
                                                     note='NOT measured by this run (--no-pmc, N > 1 or the profiler failed): committed summary of tools/prof_train.sh')
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='r50vd_608', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=8, help='images per GPU')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--in-flight', type=int, default=2, help='batches kept on the device at once (runtime.InFlight: one '
                    'executor + hipGraph + stream per lane, steps go round-robin over the lanes); 1 = one batch at a time')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-host-input', action='store_true', help='skip the PCIe-inclusive side measurement')
    ap.add_argument('--no-alt-math', action='store_true', help='skip the side measurement of the other math modes')
    ap.add_argument('--autotune', action='store_true', help='re-measure tile configs instead of using the '
                    'committed tuned_gfx950.json table')
    ap.add_argument('--freeze-at', type=int, default=None, help='--train: backbone stages 1..N frozen (default: the configuration\'s 5 = the '
                    'head trains; 1..4 add stages, incl. the DCNv2 backward of stage 5)')
    ap.add_argument('--tune-kinds', default='conv,dcn', help='with --autotune: plan op kinds to re-measure (conv,dcn)')
    ap.add_argument('--tune-cfgs', default=None, help='with --autotune: measure only these comma-separated tile ids against every layer\'s current entry')
    ap.add_argument('--tune-match', default=None, help='with --autotune: only the layers whose table key contains all of these '
                    'comma-separated pieces (e.g. ":C64:,:R1:")')
    ap.add_argument('--insitu-tune', action='store_true', help='with --autotune: re-rank the front-runners of every layer by their duration inside a '
                    'whole eager pass of the plan (HipExecutor.insitu_tune)')
    ap.add_argument('--insitu-topk', type=int, default=6)
    ap.add_argument('--co-tune', action='store_true', help='with --autotune: choose among the front-runners of a layer '
                    'the best NEIGHBOUR of a second lane (HipExecutor.co_tune)')
    ap.add_argument('--verbose-tune', action='store_true')
    ap.add_argument('--save-tuning', default=None, help='write the measured table to this JSON file')
    ap.add_argument('--layer-report', default=None, help='write per-launch timings to this JSON file')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL); "gloo" '
                    'together with --share-gpu lets the N>1 code path be smoke-tested on a 1-GPU box')
    ap.add_argument('--share-gpu', action='store_true', help='all ranks use cuda:0 (smoke test only)')
    ap.add_argument('--seed-offset', type=int, default=0, help='added to the image seed (a single-rank run with offset r '
                    'computes the batch rank r computes in a multi-rank run; tests)')
    ap.add_argument('--dump-dets', default=None, help='rank 0 writes the gathered detection records of one extra step '
                    '([world*batch, keep_top_k+1, 6]) to this .npy file (tests)')
    ap.add_argument('--train', action='store_true', help='BASELINE config 5: time the training step (ppyolo_hip/train.py) instead of '
                    'inference; prints its own JSON line')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 --pmc child passes that measure roofline.traffic '
                    '(then the committed summary under profiles/ is quoted, marked stale)')
    ap.add_argument('--pmc-child', action='store_true', help='(internal) the short eager one-lane run the PMC passes profile')
    ap.add_argument('--trace-child', action='store_true', help='(internal) the eager one-lane run with op markers that kernel_trace_leg profiles')
    ap.add_argument('--no-kernel-trace', action='store_true', help='skip the rocprofv3 --kernel-trace child pass (roofline.trace)')
    ap.add_argument('--trace-layers', default=None, help='write the trace-derived per-op table (text) to this file')
    ap.add_argument('--tune-cu-mask', default=None, help='with --autotune: measure the layers on a stream restricted to these CUs (one term of '
                    'runtime.lane_cu_masks, e.g. m256:0-127 = half of every XCD) -- the table a CU-masked lane would want')
    ap.add_argument('--no-prefetch', action='store_true', help='--train: the plain loop (no backbone prefetch of the next batch)')
    ap.add_argument('--lane-phase', type=float, default=0.0, help='diagnostic: start lane 1 this fraction of a one-lane step behind lane 0 (a spin '
                    'kernel on its stream, once, before the steady-state loop): do the lanes run in lockstep, and would an offset help?')
    ap.add_argument('--power-trace', action='store_true', help='sample rocm-smi (socket power, shader clock) during the `sustained` loop')
    ap.add_argument('--no-batch-scaling', action='store_true', help='skip roofline.batch_scaling (one lane at batch 8 / 16 / 32 on the batch-8 kernels)')
    ap.add_argument('--no-worst-case', action='store_true', help='skip the whole-step measurement in the all-pass score regime')
    ap.add_argument('--min-seconds', type=float, default=2.0, help='steady-state running before the timed K steps and '
                    'length of the `sustained` measurement (the board is power-managed: DESIGN.md 4.1)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU)
        cmd = launcher_argv(a.gpus, sys.argv[1:])
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    check_world(a.gpus, os.environ)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (the HIP path has no CPU fallback)')
    from ppyolo_hip import dist as pd
    rank, world, local = pd.init_from_env(a.backend if not a.share_gpu else (a.backend or 'gloo'))
    if a.share_gpu:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank %d needs cuda:%d but this node shows %d device(s) (--share-gpu puts all ranks on '
                         'cuda:0 for a smoke test)' % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    wl = WORKLOADS[a.workload]
    os.environ['PPYOLO_HIP_GRAPH'] = '0' if a.no_graph else '1'

    import __graft_entry__ as ge
    ge.build()
    if a.train:
        train_bench(a, wl, dev, rank, world)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    model, sd, cfg = build_model(wl['cfg'], dev)
    from ppyolo_hip import synth
    x = synth.synth_images(a.batch, wl['size'], seed=1234 + rank + a.seed_offset).to(dev)
    ims = synth.synth_im_size(a.batch).to(dev)
    depth = 1 if (a.pmc_child or a.trace_child) else max(1, a.in_flight)
    lanes = model.in_flight(depth).lanes(x)            # [(executor, stream)]; depth 1 = the plain forward's executor
    ex = lanes[0][0]
    for k, (e, _) in enumerate(lanes):
        # inputs resident in HBM before the timed region (every lane holds its own batch)
        e.set_inputs(x if k == 0 else synth.synth_images(a.batch, wl['size'], seed=1234 + rank + a.seed_offset + 100 * k).to(dev), ims)
        e.use_graph = False
        e.run()
    torch.cuda.synchronize()
    if a.autotune:
        import contextlib
        tune_ctx = contextlib.nullcontext()
        if a.tune_cu_mask:
            from ppyolo_hip import runtime as _rt
            _tune_stream = _rt._MaskedStream(dev, _rt.lane_cu_masks(a.tune_cu_mask, 1, torch.cuda.get_device_properties(dev).multi_processor_count)[0])
            tune_ctx = torch.cuda.stream(_tune_stream.stream)
        with tune_ctx:
            for alt in (a.tune_match.split('|') if a.tune_match else [None]):      # "a,b|c,d": layers matching (a and b) or (c and d)
                ex.autotune(iters=5, verbose=a.verbose_tune, kinds=tuple(a.tune_kinds.split(',')), match=alt.split(',') if alt else None,
                            only_cfgs=[int(v) for v in a.tune_cfgs.split(',')] if a.tune_cfgs else None)
            torch.cuda.synchronize()
        if a.insitu_tune:
            changed = ex.insitu_tune(topk=a.insitu_topk, verbose=a.verbose_tune)
            if rank == 0:
                print('insitu_tune: %d layers changed their config' % changed, file=sys.stderr)
        if a.co_tune and depth > 1:
            lanes[1][0].use_graph = True
            changed = ex.co_tune(lanes[1][0], verbose=a.verbose_tune)
            lanes[1][0].invalidate_graph()
            if rank == 0:
                print('co_tune: %d layers changed their config' % changed, file=sys.stderr)
        if a.save_tuning and rank == 0:
            ex.save_tuning(a.save_tuning)
        for e, _ in lanes[1:]:
            for op, src in zip(e.plan.ops, ex.plan.ops):
                if 'cfg' in src:
                    op['cfg'], op['splitk'] = src['cfg'], src['splitk']
            e._size_workspace()
            e._link_splits()
    if a.pmc_child:          # profiled by pmc_traffic_leg: a few more eager passes of the plan, nothing else
        for _ in range(3):
            ex.run()
        torch.cuda.synchronize()
        return
    if a.trace_child:        # profiled by kernel_trace_leg: eager passes with a marker launch in front of every plan op
        traced_pass(ex)
        return
    gats = [pd.DetectionGatherer(a.batch, ex.out_dets.shape[1], dev, world) for _ in lanes]
    for e, _ in lanes:
        e.use_graph = not a.no_graph
    it = [0]

    def step():
        k = it[0] % depth
        it[0] += 1
        e, st = lanes[k]
        with torch.cuda.stream(st):
            e.run()
            if world > 1:
                gats[k].gather(e.out_dets, e.out_count)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(a.warmup, depth)):
        step()
    if a.lane_phase > 0 and depth > 1:
        # one-lane step time, then a calibrated spin on lane 1's stream: lane 1 runs that far behind lane 0 from here on
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(lanes[0][1]):
            for _ in range(5):
                lanes[0][0].run()
        torch.cuda.synchronize()
        t_one = (time.perf_counter() - t0) / 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(10000000)
        e1.record()
        e1.synchronize()
        per_cycle = e0.elapsed_time(e1) * 1e-3 / 1e7
        with torch.cuda.stream(lanes[1][1]):
            torch.cuda._sleep(int(a.lane_phase * t_one / per_cycle))
        it[0] = 0
    # Untimed steady-state running before the clock starts: under dense 16-bit MFMA the board sits at its power cap and
    # the governor needs a moment to settle the shader clock -- K timed steps of 4 ms right after a cold
    # start would be measured at a clock the board does not sustain.  Every rank runs the same number of steps.
    barrier()
    settle_steps, t_settle = 0, time.perf_counter()
    while a.min_seconds > 0:
        for _ in range(2 * depth):
            step()
        settle_steps += 2 * depth
        torch.cuda.synchronize()
        go_on = torch.tensor([1.0 if time.perf_counter() - t_settle < a.min_seconds else 0.0], device=dev)
        if world > 1:
            torch.distributed.all_reduce(go_on, op=torch.distributed.ReduceOp.MAX)
        if go_on.item() == 0.0:
            break

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank's own K steps (before the closing barrier)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # `sustained`: the same loop over at least --min-seconds (a multiple of K steps), timed the same way
    sustained, power = None, None
    if a.min_seconds > 0:
        n_sus = a.steps * max(1, int(a.min_seconds / max(dt, 1e-6)) + 1)
        barrier()
        smi = _SmiSampler() if (a.power_trace and rank == 0) else None
        t1 = time.perf_counter()
        for _ in range(n_sus):
            step()
        barrier()
        dts = time.perf_counter() - t1
        power = smi.stop() if smi is not None else None
        if world > 1:
            t = torch.tensor([dts], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dts = float(t.item())
        sustained = dict(value=round(world * a.batch * n_sus / dts, 2), unit='images/s', steps=n_sus, seconds=round(dts, 3),
                         note='same loop, same timing, over >= %.1f s; `value` above is the K = %d steps of the contract, '
                              'timed after %d untimed steady-state steps' % (a.min_seconds, a.steps, settle_steps))

    if a.dump_dets:
        e0, st0 = lanes[0]
        with torch.cuda.stream(st0):
            e0.run()
            packed = gats[0].gather(e0.out_dets, e0.out_count)
        st0.synchronize()
        if rank == 0:
            import numpy as np
            np.save(a.dump_dets, packed.cpu().numpy())

    ms_per_step = dt / a.steps * 1e3
    value = world * a.batch * a.steps / dt
    # what the collective library saw, and every rank's own rate over the same timed region (the driver computes scaling
    # efficiency itself from the per-N `value`s; these fields let it check that N ranks really ran on N devices)
    ranks = dict(world_size=world, backend='none (single process)', per_rank_images_per_s=[round(a.batch * a.steps / dt_own, 1)],
                 devices=[torch.cuda.get_device_name(dev)],
                 multi_gpu_note='no multi-GPU box was available to the builder: N > 1 lines are measured by the driver only')
    if world > 1:
        own = torch.tensor([a.batch * a.steps / dt_own, float(local)], dtype=torch.float64, device=dev)
        allr = torch.zeros(world * 2, dtype=torch.float64, device=dev)
        torch.distributed.all_gather_into_tensor(allr, own)
        allr = allr.view(world, 2).cpu()
        if torch.distributed.get_backend() == 'nccl':      # what the collective library says it is (RCCL reports an NCCL-style version)
            try:
                ranks['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception as exc:
                ranks['rccl_version'] = 'unavailable (%s)' % type(exc).__name__
        ranks.update(world_size=torch.distributed.get_world_size(), backend=torch.distributed.get_backend(),
                     per_rank_images_per_s=[round(float(v), 1) for v in allr[:, 0]],
                     devices=['cuda:%d' % int(v) for v in allr[:, 1]])
    one_at_a_time = None
    if depth > 1 and world == 1:              # not `value`: the same steps with one batch on the device at a time
        torch.cuda.synchronize()
        n1, dt1 = min(a.steps, 30), 0.0
        with torch.cuda.stream(lanes[0][1]):
            while True:
                t1 = time.perf_counter()
                for _ in range(n1):
                    ex.run()
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                if dt1 >= min(a.min_seconds, 1.0) or n1 >= 4096:
                    break
                n1 *= 4
        one_at_a_time = round(a.batch * n1 / dt1, 1)

    if rank == 0:
        total_flops, per_op = conv_flops(ex.plan)
        conv_ms, covered, nconv, ideal_s, fam_flops = timed_conv_pass(ex, per_op)
        peak = covered / ideal_s / 1e12        # flop-weighted peak of the kernel mix of this step
        # Round 6 (review item 3): the solo kernel time comes from PROFILER TIMESTAMPS of a one-lane eager child of this run where
        # rocprofv3 is available (kernel_trace_leg); an event pair around a launch also times two event packets (~2.5 us per
        # launch, 5 % over 75 launches: `hip_events_over_trace`), so the HIP-event sum is kept beside it as the cross-check.
        conv_ms_events, timing = conv_ms, 'HIP events around every conv / DCN launch of an eager pass (best of 3)'
        ktrace = None
        if world == 1 and not a.no_kernel_trace:
            ktrace, ktab = kernel_trace_leg(['--workload', a.workload, '--batch', str(a.batch)], ex, per_op, table_path=a.trace_layers)
            if ktrace is None:
                ktrace = dict(error=ktab)
            else:
                ktrace['hip_events_over_trace'] = round(conv_ms_events / ktrace['kernel_ms_per_step'], 4)
                conv_ms = ktrace['kernel_ms_per_step']
                timing = ('rocprofv3 --kernel-trace timestamps of an eager one-lane child of this run (sum of End - Start of the conv / DCN '
                          'launches\' kernels, mean of %d passes)' % ktrace['passes'])
        achieved = covered / (conv_ms * 1e-3) / 1e12
        # HBM-side bytes of the conv launches from rocprofv3 PMC passes (collected separately with
        # tools/prof_run.sh; summary committed under profiles/): average per launch, like `achieved`
        traffic, traffic_src = None, None
        why_stale = '--no-pmc' if a.no_pmc else 'N > 1'
        if world == 1 and not a.no_pmc:
            traffic, traffic_src = pmc_traffic_leg(['--workload', a.workload, '--batch', str(a.batch)], nconv)
            if traffic is None:
                why_stale, traffic_src = traffic_src, None
        if traffic is None and a.workload == 'r50vd_608' and a.batch == 8:
            import glob
            files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')) if '_train_' not in os.path.basename(f))
            if files:
                with open(files[-1]) as fh:
                    rec = json.load(fh)
                traffic = round(rec['hbm_bytes_per_step'] / nconv)
                traffic_src = dict(file='profiles/' + os.path.basename(files[-1]), measured=rec.get('measured', 'round 1'), stale=True,
                                   note='NOT measured by this run (%s): committed rocprofv3 PMC summary of the named date '
                                        '(tools/prof_run.sh -> tools/pmc_traffic.py)' % why_stale)
        mfma_pmc = traffic_src.pop('mfma_busy_pmc', None) if traffic_src else None
        roof = dict(bound='mfma', achieved=round(achieved, 2), peak=round(peak, 1), unit='TFLOP/s',
                    frac=round(achieved / peak, 4), mfma_busy_pmc=mfma_pmc, traffic=traffic, traffic_source=traffic_src,
                    traffic_unit='bytes per launch (mean over the conv launches of a step; PMC 2*FETCH_SIZE+WRITE_SIZE)',
                    kernel='conv_igemm_x3_kernel<*> (fp32-in/fp32-out implicit GEMM on the 16-bit MFMA: f16x2 = 3 x '
                           'v_mfma_f32_32x32x16_f16 per product after a 2-term fp16 split, bf16x3 = 6 x ..._bf16 after a 3-term '
                           'bf16 split) / conv_igemm_glds_kernel<*> (v_mfma_f32_32x32x2_f32), %d launches/step' % nconv,
                    peak_note='achieved = algorithmic fp32 FLOPs / solo kernel time of the conv launches (kernel_ms_per_step; timing: ' + timing + '); peak = the same '
                              'FLOPs / MFMA-pipe time at peak, pricing a launch by its kernel family: f16x2 %.1f (= dense '
                              '16-bit MFMA %.1f / 3 products per multiply-add), bf16x3 %.1f (/ 6), exact fp32 %.1f TFLOP/s; '
                              'FLOP shares: %s.  Peaks are at the nominal 2.4 GHz; under dense 16-bit MFMA on real '
                              'operands the board runs at its 1400 W power cap and ~1.8-2.1 GHz (DESIGN.md 4.1)' % (
                                  F16X2_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS, X3_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS,
                                  ', '.join('%s %.1f%%' % (k, 100.0 * v / max(1, covered)) for k, v in fam_flops.items())),
                    achieved_vs_fp32_mfma_peak=round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    flops_per_step=covered, kernel_ms_per_step=round(conv_ms, 4), kernel_ms_per_step_hip_events=round(conv_ms_events, 3),
                    kernel_timing=timing, trace=ktrace,
                    whole_step_mfma_util=round(total_flops / (ms_per_step * 1e-3) / 1e12 / peak, 4))
        if one_at_a_time is not None:
            ms_one = a.batch / one_at_a_time * 1e3
            roof['frac_one_lane'] = round(total_flops / (ms_one * 1e-3) / 1e12 / peak, 4)
            roof['overlap'] = dict(ms_per_step_one_lane=round(ms_one, 3), ms_per_step_two_lanes=round(ms_per_step, 3),
                                   hidden_fraction=round(1.0 - ms_per_step / ms_one, 4), conv_kernel_ms_per_step_solo=round(conv_ms, 3),
                                   note='frac = conv FLOPs / SOLO conv kernel time (eager launches, HIP events); frac_one_lane / '
                                        'whole_step_mfma_util = the same FLOPs / whole-step wall time with one / %d batches in flight; '
                                        'hidden_fraction = share of a one-lane step the second lane hides.  Per-stream busy time and '
                                        'the overlap of the two lanes from a rocprofv3 kernel trace of this loop: '
                                        'profiles/r04_two_lane_timeline.txt (tools/two_lane_timeline.py)' % depth)
        out = dict(metric='images/sec PPYOLO R50-vd 608x608 bs=8' if a.workload == 'r50vd_608'
                   else 'images/sec %s bs=%d' % (a.workload, a.batch),
                   value=round(value, 2), unit='images/s', n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(ms_per_step, 3), ms_per_step_note=(
                       'wall time of the timed region / steps; with %d batches in flight a single batch takes about %d x this '
                       'from submit to result' % (depth, depth)) if depth > 1 else 'wall time of the timed region / steps',
                   higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='f32', data='synthetic (randn images seed 1234, deterministic random weights seed 0)',
                   config=dict(workload='%s %dx%d, %d images per GPU, device-resident input -> padded detections'
                                        % (wl['model'], wl['size'], wl['size'], a.batch),
                               global_batch=world * a.batch, parallelism='batch-sharded x%d, all-gather of detections'
                               % world if world > 1 else 'single GPU', hip_graph=not a.no_graph,
                               in_flight='%d batches of %d on the device at once (one executor, hipGraph and stream per lane; '
                                         'a step = one batch through the whole path; every step is computed in full)'
                                         % (depth, a.batch) if depth > 1 else '1 (one batch at a time)',
                               math=ex.math + {'f16x2': ' (fp32 in/out, fp32 accumulate; operands scaled by powers of two and split into 2 '
                                                         'fp16 terms, 3 partial products; error vs fp64 <= the fp32 fma chain, '
                                                         'tests/test_gpu_ops.py, tests/test_gpu_model.py::test_fp64_three_way)',
                                                'bf16x3': ' (fp32 in/out, fp32 accumulate; operands split exactly into 3 bf16 '
                                                          'terms, 6 partial products; error vs fp64 <= the fp32 fma chain)'
                                                }.get(ex.math, ''),
                               tile_table='re-measured' if a.autotune else os.path.basename(_tuned_path(ex.math))),
                   roofline=roof, ranks=ranks)
        if sustained is not None:
            out['sustained'] = sustained
            if power is not None:
                out['sustained']['power'] = power
        if one_at_a_time is not None:
            out['one_batch_at_a_time'] = dict(value=one_at_a_time, unit='images/s',
                                              note='the graph of lane 0 replayed alone (= --in-flight 1)')
        if world == 1 and not a.no_batch_scaling and a.batch == 8:
            roof['batch_scaling'] = batch_scaling_leg(model, wl, dev, ex)
        out['roofline_other'] = decode_nms_leg(ex)
        hbm_conv = hbm_conv_leg(ex)
        if hbm_conv is not None:
            out['roofline_other']['expand_1x1'] = hbm_conv
        fused = b2b_leg(ex)
        if fused is not None:
            out['roofline_other']['conv3x3_conv1x1'] = fused
        if a.layer_report:
            layer_report(ex, per_op, a.layer_report)
        if world == 1 and not a.no_alt_math:
            out['alt_math'] = alt_math_leg(wl, dev, x, ims, min(a.steps, 30), ex.math, depth)
            if ex.math != 'fp32' and 'fp32' in out['alt_math']:
                # `value` computes every product from 2-term fp16 splits (3 MFMA products, ~22-bit operands, fp32 accumulate);
                # this is the same step with every convolution on the exact-fp32 MFMA, v_mfma_f32_32x32x2_f32
                out['value_fp32_exact'] = out['alt_math']['fp32']
                out['value_fp32_exact_note'] = ('same loop, every convolution on the exact-fp32 MFMA (PPYOLO_HIP_MATH=fp32): %.2f of the '
                                                '157.3 TFLOP/s fp32-MFMA bound' % (out['alt_math']['fp32'] * total_flops / a.batch / 1e12 / FP32_MFMA_PEAK_TFLOPS))
        if world == 1 and not a.no_host_input:
            out['host_input'] = host_input_leg(ex, x, ims, min(a.steps, 30), lanes)
        if world == 1 and not a.no_host_input:
            out['roofline_other']['preprocess'] = preprocess_leg(cfg, wl['size'], a.batch, lanes, min(a.steps, 30),
                                                                 not a.no_cpu_baseline)
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sd, cfg, wl['size'], a.batch)
        if world == 1:
            try:
                pn = parity_note_leg(model, wl, a.batch, x)
            except Exception as exc:          # (a diagnostic field must not take the benchmark line with it)
                pn = dict(error='%s: %s' % (type(exc).__name__, exc))
            if pn is not None:
                out['parity_note'] = pn
        if world == 1 and not a.no_worst_case:
            out['worst_case_regime'] = worst_case_leg(wl, dev, x, ims, sd, cfg, depth, min(a.min_seconds, 1.0) or 0.2,
                                                      out['cpu_baseline']['cores'] if 'cpu_baseline' in out else 0)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
