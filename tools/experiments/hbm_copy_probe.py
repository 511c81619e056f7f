"""Calibration for the HBM-bound layers: what a plain device copy / fill / read reaches on THIS box (power cap included), at the sizes of the
stage-2 / stage-3 tensors.  torch ops only (ATen's vectorised elementwise kernels): nothing of the product path.
usage: python tools/experiments/hbm_copy_probe.py"""
import torch


def timed(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    dev = torch.device('cuda:0')
    for mb in (24, 47, 95, 189, 378, 757):
        n = mb * 1024 * 1024 // 4
        x = torch.randn(n, device=dev)
        y = torch.empty_like(x)
        t_copy = timed(lambda: y.copy_(x))
        t_fill = timed(lambda: y.fill_(1.0))
        t_read = timed(lambda: torch.amax(x))
        t_add = timed(lambda: torch.add(x, y, out=y))      # 2 reads + 1 write
        b = n * 4
        print('%4d MB  copy %7.1f us = %5.2f TB/s   fill %7.1f us = %5.2f TB/s   amax (read) %7.1f us = %5.2f TB/s   add (2r+1w) %7.1f us = %5.2f TB/s' % (
            mb, t_copy, 2 * b / t_copy / 1e6, t_fill, b / t_fill / 1e6, t_read, b / t_read / 1e6, t_add, 3 * b / t_add / 1e6))


if __name__ == '__main__':
    main()
