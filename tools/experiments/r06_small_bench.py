"""Round 6: the wave-private small-output tiles (csrc/conv_small.hip) against the table's entry, layer by layer, timed as nodes of a
captured graph (16 launches per replay) -- fp32 input, no shortcut.
usage: python tools/experiments/r06_small_bench.py [N [abs_id,abs_id,...]]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch  # noqa: E402
from ppyolo_hip import ops  # noqa: E402


def timed(run, reps=8):
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(16):
            run()
    g.replay()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            g.replay()
        e.record()
        e.synchronize()
        best = min(best, s.elapsed_time(e) / (16 * reps))
    g.reset()
    return best * 1e3      # us


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    table = json.load(open(os.path.join(ROOT, 'pytorch-ppyolo_amd', 'ppyolo_hip', 'tuned_gfx950_f16x2.json')))
    first = ops.small_first_cfg()
    ws = torch.empty(64 << 20, device='cuda')
    keys = [k for k in table if k.startswith('conv:N%d:' % N) and k.endswith(':f') and re.search(r':H(19|38|76|152):' if len(sys.argv) > 2 else r':H(19|38|76):', k)]
    for key in sorted(keys, key=lambda k: (-int(re.search(r':H(\d+)', k).group(1)), k)):
        m = re.match(r'conv:N(\d+):H(\d+):W(\d+):C(\d+):K(\d+):R(\d+):s(\d+)', key)
        n, H, W, C, K, R, stride = [int(v) for v in m.groups()]
        pad = (R - 1) // 2
        Ho, Wo = ops.conv_out_hw(H, W, R, R, stride, pad)
        x = torch.randn(n, H, W, C, device='cuda')
        w = torch.randn(K, R, R, C, device='cuda') * 0.05
        sc, sh = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
        y = torch.empty(n, Ho, Wo, K, device='cuda')
        wf = ops.split_weights_f16x2(w, sc)
        amax, amax_out = ops.amax_slots(x), ops.amax_slots(device='cuda', N=n)

        def mk(cfg, s):
            return lambda: ops.conv2d_bn_act(ops.View(x), w, sc, sh, ops.View(y), stride, pad, 'relu', cfg=cfg, splitk=s, ws=ws, w_f16=wf,
                                             amax_in=amax, amax_out=amax_out)
        ent = table[key]
        t0 = timed(mk(ent[0], ent[1]))
        row = []
        chunks = R * R * C // 32
        if len(sys.argv) > 2:          # explicit absolute ids (e.g. the persistent tiles of csrc/conv_wsp.hip), one split
            for c in [int(v) for v in sys.argv[2].split(',')]:
                try:
                    row.append((timed(mk(c, 1)), c, 1))
                except Exception:
                    pass
        else:
            for i in range(4):
                for s in (1, 2, 4, 8):
                    if s > (4 if i < 2 else 8) or s > chunks:
                        continue
                    row.append((timed(mk(first + i, s)), i, s))
        row.sort()
        print('%-44s table [%3d, %2d] %6.2f us | small: %s' % (key, ent[0], ent[1], t0, '  '.join('%d/%d %.2f' % (i, s, t) for t, i, s in row[:5])), flush=True)


if __name__ == '__main__':
    main()
