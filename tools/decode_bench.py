"""Time the decode launch of the R50vd-608 bs-8 step under the kernel's experiment knobs (PPY_DECODE_ABL / _PER_WAVE / _STAGED).
    python tools/decode_bench.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from ppyolo_hip import ops as K, synth  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    model, sd, cfg = bench.build_model('PPYOLO_2x_Config', dev)
    x = synth.synth_images(8, 608).to(dev)
    ims = synth.synth_im_size(8).to(dev)
    ex = model._plans.executor(x)
    ex.set_inputs(x, ims)
    ex.use_graph = False
    ex.run()
    torch.cuda.synchronize()
    d = ex.plan.decode
    heads = [ex.view(a) for a in ex.plan.head_outs]
    n = d['nms']

    def decode():
        K.yolo_decode_levels(heads, [lvl['anchors'] for lvl in d['levels']], [lvl['downsample'] for lvl in d['levels']],
                             d['num_classes'], d['scale_x_y'], d['iou_aware'], d['iou_aware_factor'], d['clip_bbox'],
                             ex.im_size, ex.boxes, n['score_threshold'], ex.cand_key, ex.cand_idx, ex.cand_count)

    from ppyolo_hip._lib import lib

    def t(label, **env):
        # (the library reads its environment once: the switches go through its debug hook; PPY_DECODE_ABL only acts in a build
        # with PPY_EXTRA_HIPCC_FLAGS=-DPPY_DECODE_ABLATE)
        lib().ppy_debug_decode_mode(int(env.get('PPY_DECODE_STAGED', 0)), int(env.get('PPY_DECODE_PER_WAVE', -1)), int(env.get('PPY_DECODE_ABL', 0)))
        try:
            ex.cand_count.zero_()
            decode()
            torch.cuda.synchronize()
            best = None
            for _ in range(5):
                ex.cand_count.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    decode()
                e1.record()
                e1.synchronize()
                v = e0.elapsed_time(e1) / 20
                best = v if best is None else min(best, v)
            print('%-60s %7.1f us' % (label, best * 1e3), flush=True)
        finally:
            lib().ppy_debug_decode_mode(-1, -1, 0)
    t('staged kernel (rounds 1-2)', PPY_DECODE_STAGED=1)
    for pw in (1, 2, 3, 4, 6, 8):
        t('stream, %d group(s) per wave' % pw, PPY_DECODE_PER_WAVE=pw)
    for pw in (1, 2, 4):
        t('stream, %d per wave, no flush atomic' % pw, PPY_DECODE_PER_WAVE=pw, PPY_DECODE_ABL=1)
        t('stream, %d per wave, no pair phase' % pw, PPY_DECODE_PER_WAVE=pw, PPY_DECODE_ABL=2)
        t('stream, %d per wave, no sweep' % pw, PPY_DECODE_PER_WAVE=pw, PPY_DECODE_ABL=4)
        t('stream, %d per wave, loads only' % pw, PPY_DECODE_PER_WAVE=pw, PPY_DECODE_ABL=7)


if __name__ == '__main__':
    main()
