"""Round 6: the fused conv2 -> conv3 launch with the row-window main loop against the library of the commit before (PPYOLO_HIP_LIB_BASE):
the same products in the same order per output pixel -- the two must be EQUAL bit for bit at the full stage-2 size (with and without
the pooled output), and the new one repeatable.  Prints graph-timed durations of both.
usage: PPYOLO_HIP_LIB=<lib> python tools/experiments/r06_b2b_window_check.py dump|check <file>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-ppyolo_amd'))
import torch  # noqa: E402
from ppyolo_hip import ops  # noqa: E402


def run(pool, N=8, H=152, W=152):
    g = torch.Generator().manual_seed(77)
    x = torch.relu(torch.randn(N, H, W, 64, generator=g)).cuda()
    wA = (torch.randn(64, 3, 3, 64, generator=g) * (2.0 / 576) ** 0.5).cuda()
    wB = (torch.randn(256, 1, 1, 64, generator=g) * (2.0 / 64) ** 0.5).cuda()
    scA, shA = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()
    scB, shB = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
    res = torch.randn(N, H, W, 256, generator=g).cuda()
    fA, fB = ops.split_weights_f16x2(wA, scA), ops.split_weights_f16x2(wB, scB)
    # a pre-split input as a producer would write it: through a 1x1 identity-like producer launch
    w0 = torch.eye(64).view(64, 1, 1, 64).contiguous().cuda()
    one, zero = torch.ones(64).cuda(), torch.zeros(64).cuda()
    f0 = ops.split_weights_f16x2(w0, one)
    xs = torch.zeros(N, H, W, 64).cuda()
    a_x, a_xs = ops.amax_slots(x), ops.amax_slots(N=N, device='cuda')
    ys = (torch.ones(N).cuda(), 1.0 + 2.0 ** -8, 1e-30)
    ops.conv2d_bn_act(ops.View(x), w0, one, zero, ops.View(xs), 1, 0, 'relu', None, None, False, 41, 1, None, None, f0, a_x, a_xs, None, None, ys)
    y = torch.zeros(N, H, W, 256).cuda()
    pl = torch.zeros(N, H // 2, W // 2, 256).cuda() if pool else None
    a_out = ops.amax_slots(N=N, device='cuda')
    mul = float((scA.abs().double().cpu() * wA.abs().double().cpu().sum(dim=(1, 2, 3))).max()) * (1 + 2.0 ** -8)
    add = float(shA.abs().max()) * (1 + 2.0 ** -8) + 1e-30

    def launch():
        ops.conv3x3_conv1x1(ops.View(xs), ys[0], a_xs, fA, shA, fB, shB, ops.View(res), ops.View(y), mul, add, a_out,
                            None if pl is None else ops.View(pl))
    launch()
    torch.cuda.synchronize()
    first = y.clone()
    for _ in range(5):
        launch()
        torch.cuda.synchronize()
        assert torch.equal(y, first), 'not repeatable'
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(8):
            launch()
    gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        gr.replay()
    e.record()
    e.synchronize()
    us = s.elapsed_time(e) / 80 * 1e3
    return y.cpu(), None if pl is None else pl.cpu(), a_out.view(N, -1).amax(dim=1).cpu(), us


def main():
    mode, path = sys.argv[1], sys.argv[2]
    out = {}
    for pool in (False, True):
        y, pl, am, us = run(pool)
        print('%s pool=%s: %.1f us per launch' % (os.environ.get('PPYOLO_HIP_LIB', 'default lib'), pool, us), flush=True)
        out['y%d' % pool], out['p%d' % pool], out['a%d' % pool] = y, pl, am
    if mode == 'dump':
        torch.save(out, path)
    else:
        ref = torch.load(path)
        for k, v in out.items():
            if v is None:
                continue
            same = torch.equal(v, ref[k])
            print('%s: %s' % (k, 'EQUAL' if same else 'DIFFERENT (max %.3e)' % float((v - ref[k]).abs().max())))


if __name__ == '__main__':
    main()
