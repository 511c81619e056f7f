"""Where the fused conv2 -> conv3 launch (csrc/conv_b2b.hip) spends its time: the same launch with phases switched off
(PPY_B2B_SKIP, timing only -- results are wrong by construction), one process per setting."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'pytorch-ppyolo_amd'))


def child(pool):
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
    out = torch.zeros(N, H, W, 256).cuda()
    pl = torch.zeros(N, H // 2, W // 2, 256).cuda() if pool else None
    a_in, a1, a_out = ops.amax_slots(x), ops.amax_slots(N=N, device='cuda'), ops.amax_slots(N=N, device='cuda')
    mul = float(w1.abs().double().sum(dim=(1, 2, 3)).max()) * (1 + 2.0 ** -8)
    ys = (torch.ones(N).cuda(), mul, 0.0)
    ops.conv2d_bn_act(ops.View(x), w1, one64, z64, ops.View(t1), 1, 0, 'relu', None, None, False, 44, 1, None, None, fs[0], a_in, a1, None, None, ys)
    tm = float(w2.abs().double().sum(dim=(1, 2, 3)).max()) * (1 + 2.0 ** -8)

    def run():
        ops.conv3x3_conv1x1(ops.View(t1), ys[0], a1, fs[1], z64, fs[2], z256, ops.View(x), ops.View(out), tm, 0.0, a_out,
                            None if pl is None else ops.View(pl))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(20):
                run()
    gr.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print('%.1f' % (e0.elapsed_time(e1) * 10.0))      # us per launch


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1] == '1')
    else:
        for pool in (0, 1):
            for skip, what in ((0, 'whole'), (1, 'no conv A loop'), (2, 'conv A loop only'), (4, 'no stores'), (5, 'no loop, no stores'), (3, 'prologue only')):
                env = dict(os.environ, PPY_B2B_SKIP=str(skip))
                r = subprocess.run([sys.executable, __file__, str(pool)], env=env, capture_output=True, text=True, timeout=300)
                print('pool=%d skip=%d %-22s %s us' % (pool, skip, what, r.stdout.strip() or r.stderr[-300:]), flush=True)
