"""Is the fused conv2 -> conv3 launch bit-repeatable?  The same launch R times on the same inputs, outputs compared bit for bit
with the first; PPY_B2B_SKIP bits 8 / 16 replace the counted waits by full ones (experiment)."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'pytorch-ppyolo_amd'))


def child(pool, reps):
    import torch
    from ppyolo_hip import ops
    N, H, W = 8, 152, 152
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(N, H, W, 256, generator=g)).cuda()
    w1 = (torch.randn(64, 1, 1, 256, generator=g) * (2.0 / 256) ** 0.5).cuda()
    w2 = (torch.randn(64, 3, 3, 64, generator=g) * (2.0 / 576) ** 0.5).cuda()
    w3 = (torch.randn(256, 1, 1, 64, generator=g) * (2.0 / 64) ** 0.5).cuda()
    one64, one256 = torch.ones(64).cuda(), torch.ones(256).cuda()
    z64, z256 = torch.zeros(64).cuda(), torch.zeros(256).cuda()
    fs = [ops.split_weights_f16x2(w1, one64), ops.split_weights_f16x2(w2, one64), ops.split_weights_f16x2(w3, one256)]
    t1 = torch.zeros(N, H, W, 64).cuda()
    a_in, a1 = ops.amax_slots(x), ops.amax_slots(N=N, device='cuda')
    mul = float(w1.abs().double().sum(dim=(1, 2, 3)).max()) * (1 + 2.0 ** -8)
    ys = (torch.ones(N).cuda(), mul, 0.0)
    ops.conv2d_bn_act(ops.View(x), w1, one64, z64, ops.View(t1), 1, 0, 'relu', None, None, False, 44, 1, None, None, fs[0], a_in, a1, None, None, ys)
    tm = float(w2.abs().double().sum(dim=(1, 2, 3)).max()) * (1 + 2.0 ** -8)
    first, firstp, bad, badp, worst = None, None, 0, 0, 0.0
    # a second stream keeps the memory system busy beside the launch, as the other lane does
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device='cuda')
    for r in range(reps):
        out = torch.full((N, H, W, 256), float('nan')).cuda()
        pl = torch.full((N, H // 2, W // 2, 256), float('nan')).cuda() if pool else None
        a_out = ops.amax_slots(N=N, device='cuda')
        if r % 2:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        ops.conv3x3_conv1x1(ops.View(t1), ys[0], a1, fs[1], z64, fs[2], z256, ops.View(x), ops.View(out), tm, 0.0, a_out,
                            None if pl is None else ops.View(pl))
        torch.cuda.synchronize()
        if first is None:
            first, firstp = out, pl
        else:
            if not torch.equal(out, first):
                bad += 1
                worst = max(worst, float((out - first).abs().max()))
            if pool and not torch.equal(pl, firstp):
                badp += 1
    print('%d of %d runs differ from the first (pooled: %d), max |diff| %.3e' % (bad, reps - 1, badp, worst))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1] == '1', int(sys.argv[2]))
    else:
        for pool in (0, 1):
            for skip in (0, 8, 16, 24):
                env = dict(os.environ, PPY_B2B_SKIP=str(skip))
                r = subprocess.run([sys.executable, __file__, str(pool), '300'], env=env, capture_output=True, text=True, timeout=600)
                print('pool=%d PPY_B2B_SKIP=%-2d %s' % (pool, skip, r.stdout.strip() or r.stderr[-400:]), flush=True)
