"""CPU interpreter of a ppyolo_hip Plan -- TEST INFRASTRUCTURE.

Executes the very same plan data (ops over NHWC buffer slices, folded BN, KRSC weights,
CoordConv bias maps, concat slices, fused upsample / residual) that the HIP executor replays,
but with plain PyTorch-CPU ops and the oracle's decode / NMS.  It lets the host-side logic of
the product (plan construction) be verified without a GPU; it is never imported by the
product."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ppyolo_oracle as orc


class CpuPlanRunner(object):
    def __init__(self, plan):
        self.plan = plan
        self.bufs = []
        for i, (N, H, W, ld) in enumerate(plan.buffers):
            if i in plan.consts:
                self.bufs.append(plan.consts[i].clone().float())
            else:
                self.bufs.append(torch.full((N, H, W, ld), float('nan')))
        for op in plan.setup_ops:
            self._run(op, None)

    def get(self, a):
        return self.bufs[a.buf][..., a.coff:a.coff + a.C]

    def put(self, a, v_nhwc):
        self.bufs[a.buf][..., a.coff:a.coff + a.C] = v_nhwc

    @staticmethod
    def _act(y, act):
        if act == 'relu':
            return F.relu(y)
        if act == 'leaky':
            return F.leaky_relu(y, 0.1)
        assert act is None
        return y

    def _run(self, op, x_in):
        t = op['op']
        if t == 'stem':
            y = F.conv2d(x_in, op['w'], None, stride=2, padding=1)
            y = y * op['scale'].view(1, -1, 1, 1) + op['shift'].view(1, -1, 1, 1)
            self.put(op['y'], self._act(y, op['act']).permute(0, 2, 3, 1))
        elif t == 'conv':
            x = self.get(op['x']).permute(0, 3, 1, 2)
            assert not torch.isnan(x).any(), 'conv reads an unwritten buffer region'
            w = op['w'].permute(0, 3, 1, 2)
            y = F.conv2d(x, w, None, stride=op['stride'], padding=op['pad'])
            if op['posb'] is not None:
                y = y + self.get(op['posb']).permute(0, 3, 1, 2)
            y = y * op['scale'].view(1, -1, 1, 1) + op['shift'].view(1, -1, 1, 1)
            if op['res'] is not None:
                y = y + self.get(op['res']).permute(0, 3, 1, 2)
            y = self._act(y, op['act'])
            if op['ups']:
                y = F.interpolate(y, scale_factor=2, mode='nearest')
            self.put(op['y'], y.permute(0, 2, 3, 1))
        elif t == 'maxpool':
            self.put(op['y'], F.max_pool2d(self.get(op['x']).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1))
        elif t == 'avgpool':
            self.put(op['y'], F.avg_pool2d(self.get(op['x']).permute(0, 3, 1, 2), 2, 2, 0).permute(0, 2, 3, 1))
        elif t == 'spp':
            x = self.get(op['x']).permute(0, 3, 1, 2)
            for k, key in ((5, 'y5'), (9, 'y9'), (13, 'y13')):
                self.put(op[key], F.max_pool2d(x, k, 1, k // 2).permute(0, 2, 3, 1))
        elif t == 'dcn':
            x = self.get(op['x']).permute(0, 3, 1, 2)
            om = self.get(op['om']).permute(0, 3, 1, 2)
            val = orc.dcnv2_sample(x, om[:, :18], torch.sigmoid(om[:, 18:]), op['stride'], op['pad'], 3)
            N, oH, oW, KK, C = val.shape
            cols = val.reshape(N * oH * oW, KK * C)                       # (tap, c) order
            w = op['w'].reshape(op['w'].shape[0], -1)                     # [K][(r,s,c)]
            y = cols @ w.t()
            y = y * op['scale'].view(1, -1) + op['shift'].view(1, -1)
            self.put(op['y'], self._act(y, op['act']).reshape(N, oH, oW, -1))
        else:
            raise KeyError(t)

    def run(self, x, im_size=None):
        with torch.no_grad():
            for op in self.plan.ops:
                self._run(op, x)
            feats = [self.get(a).permute(0, 3, 1, 2).contiguous() for a in self.plan.feats]
            outs = [self.get(a).permute(0, 3, 1, 2).contiguous() for a in self.plan.head_outs]
            preds = None
            d = self.plan.decode
            if d is not None and im_size is not None:
                hcfg = dict(anchors=None)
                boxes, scores = [], []
                for lvl, o in zip(d['levels'], outs):
                    if d['iou_aware']:
                        o = orc.iou_aware_score(o, len(lvl['anchors']), d['num_classes'], d['iou_aware_factor'])
                    bx, sc = orc.yolo_box(o, np.asarray(lvl['anchors'], np.float32), lvl['downsample'],
                                          d['num_classes'], d['scale_x_y'], im_size, d['clip_bbox'])
                    boxes.append(bx)
                    scores.append(sc)
                boxes, scores = torch.cat(boxes, 1), torch.cat(scores, 1)
                preds = [orc.matrix_nms(boxes[i], scores[i], **d['nms']) for i in range(boxes.shape[0])]
            return feats, outs, preds
