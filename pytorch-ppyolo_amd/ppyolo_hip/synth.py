"""Deterministic synthetic weights / inputs for parity tests and benchmarks.

There is no network access for real COCO checkpoints, so tests, goldens and
`bench.py` all use weights produced here.  The generator is a pure function of
(ordered {name: shape}, seed): applying it to the reference model's
`state_dict()` shapes (tools/make_goldens.py, build container only) and to this
package's model gives bit-identical tensors, which is what lets committed golden
outputs be replayed on the GPU box without the reference.

Why not the reference's default init: it zero-initialises `conv_offset`
(reference model/custom_layers.py:510-511 -> all DCN offsets 0, masks 0.5) and
leaves BN at identity, so a broken gather or a wrong BN fold would still pass;
and its head outputs make every score ~0.25 (massive sort ties).  The rules
below give non-trivial BN statistics, DCN offsets of a few pixels (including
out-of-range samples) and a spread of detection scores of which ~1 % pass the
0.01 threshold.
"""
import math
from collections import OrderedDict

import torch


def _fan_in(shape):
    n = 1
    for d in shape[1:]:
        n *= d
    return n


def synth_state_dict(shapes, seed=0, num_classes=80, iou_aware=None):
    """shapes: ordered mapping name -> shape (reference state_dict key layout,
    SURVEY.md section 8b).  Returns an OrderedDict of CPU fp32 tensors (int64 for
    num_batches_tracked)."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    out = OrderedDict()

    def randn(shape, std):
        return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std

    def rand(shape, lo, hi):
        return torch.rand(tuple(shape), generator=g, dtype=torch.float32) * (hi - lo) + lo

    # residual-branch tail + projection shortcut: (conv3, conv4) in bottleneck nets
    # (reference model/resnet_vd.py:27-33), (conv2, conv3) in BasicBlock nets (:233-241)
    bottleneck = any('.conv4.' in k for k in shapes)
    tails = ('.conv3.bn.', '.conv4.bn.') if bottleneck else ('.conv2.bn.', '.conv3.bn.')

    for name, shape in shapes.items():
        shape = tuple(shape)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[name] = torch.zeros((), dtype=torch.int64)
        elif '.bn.' in name:
            # residual-branch tails / projection shortcuts get a smaller gamma so the
            # residual stream does not blow up over 16 blocks
            tail = name.startswith('backbone.stage') and any(t in name for t in tails)
            if leaf == 'weight':
                out[name] = rand(shape, 0.25, 0.55) if tail else rand(shape, 0.7, 1.3)
            elif leaf == 'bias':
                out[name] = randn(shape, 0.1)
            elif leaf == 'running_mean':
                out[name] = randn(shape, 0.1)
            elif leaf == 'running_var':
                out[name] = rand(shape, 0.6, 1.4)
            else:
                raise KeyError(name)
        elif 'conv_offset.weight' in name:
            out[name] = randn(shape, 2.0 / math.sqrt(_fan_in(shape)))
        elif 'conv_offset.bias' in name:
            b = randn(shape, 0.5)
            out[name] = b
        elif 'yolo_output_convs' in name and leaf == 'weight':
            # gain picked per net family so the data-dependent part of the logits has
            # std ~1.2 (tip-feature rms is ~1.7 for the R50 head, ~0.4 for the r18 head)
            g_out = 0.7 if bottleneck else 3.0
            out[name] = randn(shape, g_out / math.sqrt(_fan_in(shape)))
        elif 'yolo_output_convs' in name and leaf == 'bias':
            out[name] = _head_bias(shape[0], num_classes, randn, iou_aware)
        elif leaf in ('weight', 'dcn_weight') and len(shape) == 4:
            out[name] = randn(shape, math.sqrt(2.0 / _fan_in(shape)))
        else:
            raise KeyError('no synthetic rule for %s %s' % (name, shape))
    return out


def _head_bias(nch, num_classes, randn, iou_aware):
    """Bias of a YOLO output conv.  Channel layout (reference model/head.py:83-93,
    :27-31): [A IoU logits] + A x [tx, ty, tw, th, obj, C class logits]."""
    if iou_aware is None:
        iou_aware = (nch % (num_classes + 6) == 0) and (nch % (num_classes + 5) != 0)
    per = num_classes + (6 if iou_aware else 5)
    A = nch // per
    b = torch.zeros(nch, dtype=torch.float32)
    off = 0
    if iou_aware:
        b[:A] = -1.0 + randn((A,), 0.3)
        off = A
    for a in range(A):
        s = off + a * (num_classes + 5)
        b[s:s + 2] = randn((2,), 0.3)
        b[s + 2:s + 4] = -0.3 + randn((2,), 0.2)
        b[s + 4] = (-5.0 if iou_aware else -3.6) + float(randn((1,), 0.2))
        b[s + 5:s + 5 + num_classes] = -4.4 + randn((num_classes,), 0.4)
    return b


def synth_images(n, size, seed=1234):
    """Post-normalisation images are ~N(0,1) (reference config/ppyolo_2x.py:193-198)."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    return torch.randn((n, 3, size, size), generator=g, dtype=torch.float32)


def synth_im_size(n, h=480., w=640.):
    """im_size rows are (h, w) of the ORIGINAL image (reference model/head.py:61-63)."""
    return torch.tensor([[h, w]] * n, dtype=torch.float32)
