"""Round 6: the pre-processing launch (8 raw 480x640 images -> 608x608 fp32 NCHW) replayed from a graph, as bench.py's preprocess leg times it.
usage: python tools/experiments/r06_pre_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from config import PPYOLO_2x_Config  # noqa: E402
from ppyolo_hip import ops as K  # noqa: E402
from ppyolo_hip.preprocess import Preprocessor  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    cfg, size, batch = PPYOLO_2x_Config(), 608, 8
    pre = Preprocessor(cfg, size, dev)
    rng = np.random.RandomState(7)
    on_dev = [torch.from_numpy(rng.randint(0, 256, size=(480, 640, 3)).astype(np.uint8)).to(dev) for _ in range(batch)]
    out = torch.empty((batch, 3, size, size), dtype=torch.float32, device=dev)
    K.preprocess_images(on_dev, size, pre.lut, out, swap_rb=pre.to_rgb)
    torch.cuda.synchronize()
    print('checksum %.6f' % float(out.double().sum()))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            K.preprocess_images(on_dev, size, pre.lut, out, swap_rb=pre.to_rgb)
    g.replay()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20.0)
    byt = sum(r.numel() for r in on_dev) + out.numel() * 4
    print('%s: %.1f us per launch = %.2f TB/s (%.1f MB)' % (os.environ.get('PPYOLO_HIP_LIB', 'default lib'), best * 1e3, byt / best / 1e9, byt / 1e6))


if __name__ == '__main__':
    main()
