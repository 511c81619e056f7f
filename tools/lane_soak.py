#!/usr/bin/env python
"""Soak test of runtime.InFlight: two lanes run different batches at the same time, round after round, and every
buffer of every lane (head outputs, boxes, candidate counts, detections, keep indices) must equal, bit for bit, what one
executor computes for that batch alone.  usage: lane_soak.py [workload r18|r50] [rounds] [batch] [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
from conftest import build_model  # noqa: E402
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config  # noqa: E402
from ppyolo_hip import synth  # noqa: E402


def snap(e):
    d = {'head%d' % i: e.view(a).dense().clone() for i, a in enumerate(e.plan.head_outs)}
    d.update(boxes=e.boxes.clone(), cand_count=e.cand_count.clone(), out_dets=e.out_dets.clone(),
             out_count=e.out_count.clone(), out_keep=e.out_keep.clone())
    return d


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'r18'
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    S = int(sys.argv[4]) if len(sys.argv) > 4 else 320
    cfg = PPYOLO_r18vd_Config() if wl == 'r18' else PPYOLO_2x_Config()
    model, _ = build_model(cfg, 0, 'cuda')
    NB = 6
    batches = [(synth.synth_images(N, S, seed=300 + i).cuda(), synth.synth_im_size(N).cuda()) for i in range(NB)]
    ex0 = model._plans.executor(batches[0][0])
    want = []
    for x, ims in batches:
        model(x, ims)
        want.append(snap(ex0))
    lanes = model.in_flight(2).lanes(batches[0][0])
    bad = {}
    for r in range(rounds):
        pair = [(2 * r) % NB, (2 * r + 1 + r // NB) % NB]
        for k, (e, st) in enumerate(lanes):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                e.set_inputs(*batches[pair[k]])
                e.run()
        torch.cuda.synchronize()
        for k, (e, _) in enumerate(lanes):
            got = snap(e)
            for nm, g in got.items():
                if not torch.equal(g, want[pair[k]][nm]):
                    bad[nm] = bad.get(nm, 0) + 1
    print('lane_soak %s N=%d S=%d math=%s lib=%s: %d rounds x 2 lanes, mismatching buffers: %s' % (
        wl, N, S, ex0.math, os.path.basename(os.environ.get('PPYOLO_HIP_LIB', 'default')), rounds, bad or 'none'))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
