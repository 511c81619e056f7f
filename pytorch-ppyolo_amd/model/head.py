"""YOLOv3 head (CoordConv / SPP / DropBlock-as-identity) of the MI355X PP-YOLO path.

Same class names, constructor arguments and `state_dict` keys as the reference's
`model/head.py` (DetectionBlock :146-231, YOLOv3Head :242-398).  The ModuleLists keep the
reference's exact entries -- including the parameter-less CoordConv / SPP / DropBlock
placeholders -- because their positions are what the checkpoint keys
(`head.detection_blocks.{i}.layers.{k}.*`) encode.  Execution is by plan emission
(ppyolo_hip/engine.py); decode + Matrix-NMS parameters are handed to the executor, which
runs ppy_yolo_decode_f32 / ppy_matrix_nms_f32 (reference :21-141, :424-469).
"""
import copy

import numpy as np
import torch

from model.custom_layers import Conv2dUnit, CoordConv, SPP, DropBlock, get_norm


class DetectionBlock(torch.nn.Module):
    def __init__(self, in_c, channel, coord_conv=True, bn=0, gn=0, af=0, norm_decay=0., conv_block_num=2,
                 is_first=False, use_spp=True, drop_block=True, block_size=3, keep_prob=0.9, is_test=True, name=''):
        super(DetectionBlock, self).__init__()
        assert channel % 2 == 0, "channel {} cannot be divided by 2".format(channel)
        self.coord_conv, self.use_spp, self.is_first = coord_conv, use_spp, is_first
        extra = 2 if coord_conv else 0
        kw = dict(bn=bn, gn=gn, af=af, act='leaky', norm_decay=norm_decay)
        seq = []
        for j in range(conv_block_num):
            seq.append(CoordConv(coord_conv))
            seq.append(Conv2dUnit(in_c + extra, channel, 1, stride=1, name='{}.{}.0'.format(name, j), **kw))
            if use_spp and is_first and j == 1:
                seq.append(SPP())
                seq.append(Conv2dUnit(channel * 4, 512, 1, stride=1, name='{}.{}.spp.conv'.format(name, j), **kw))
                seq.append(Conv2dUnit(512, channel * 2, 3, stride=1, name='{}.{}.1'.format(name, j), **kw))
            else:
                seq.append(Conv2dUnit(channel, channel * 2, 3, stride=1, name='{}.{}.1'.format(name, j), **kw))
            if drop_block and j == 0 and not is_first:
                seq.append(DropBlock(block_size=block_size, keep_prob=keep_prob, is_test=is_test))
            in_c = channel * 2
        if drop_block and is_first:
            seq.append(DropBlock(block_size=block_size, keep_prob=keep_prob, is_test=is_test))
        seq.append(CoordConv(coord_conv))
        route_in = in_c if conv_block_num == 0 else channel * 2
        seq.append(Conv2dUnit(route_in + extra, channel, 1, stride=1, name='{}.2'.format(name), **kw))
        self.layers = torch.nn.ModuleList(seq)
        self.tip_layers = torch.nn.ModuleList([
            CoordConv(coord_conv),
            Conv2dUnit(channel + extra, channel * 2, 3, stride=1, name='{}.tip'.format(name), **kw)])

    def add_param_group(self, param_groups, base_lr, base_wd):      # reference model/head.py:233-239
        for ly in list(self.layers) + list(self.tip_layers):
            if isinstance(ly, Conv2dUnit):
                ly.add_param_group(param_groups, base_lr, base_wd)

    def _walk(self, b, seq, x):
        coord = False
        for k, ly in enumerate(seq):
            if isinstance(ly, CoordConv):
                coord = ly.coord_conv
            elif isinstance(ly, DropBlock):
                if not ly.is_test:
                    raise NotImplementedError('this is the inference plan: call head.set_dropblock(is_test=True) (reference '
                                              'demo.py:93); DropBlock in training mode runs in the training step -- '
                                              'model(x, None, False, ...) / ppyolo_hip.train.TrainStep')
            elif isinstance(ly, SPP):
                x = b.spp(x)            # x is slot 0 of the 4C buffer (arranged below)
            else:
                out = None
                if k + 1 < len(seq) and isinstance(seq[k + 1], SPP):
                    wide = b.new_act(x.N, x.H, x.W, 4 * ly.filters)
                    out = b.slice(wide, 0, ly.filters)
                x = ly.emit(b, x, out=out, coord=coord)
                coord = False
        return x

    def emit(self, b, x, tip_on_side=False):
        route = self._walk(b, self.layers, x)
        if tip_on_side:
            with b.side():
                tip = self._walk(b, self.tip_layers, route)
        else:
            tip = self._walk(b, self.tip_layers, route)
        return route, tip


class YOLOv3Head(torch.nn.Module):
    def __init__(self, conv_block_num=2, num_classes=80,
                 anchors=[[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198],
                          [373, 326]],
                 anchor_masks=[[6, 7, 8], [3, 4, 5], [0, 1, 2]], norm_type="bn", norm_decay=0., coord_conv=True,
                 iou_aware=True, iou_aware_factor=0.4, block_size=3, scale_x_y=1.05, spp=True, drop_block=True,
                 keep_prob=0.9, clip_bbox=True, yolo_loss=None, downsample=[32, 16, 8],
                 in_channels=[2048, 1024, 512], nms_cfg=None, focalloss_on_obj=False, prior_prob=0.01,
                 is_train=False):
        super(YOLOv3Head, self).__init__()
        self.conv_block_num, self.num_classes = conv_block_num, num_classes
        self.coord_conv, self.iou_aware, self.iou_aware_factor = coord_conv, iou_aware, iou_aware_factor
        self.scale_x_y, self.use_spp, self.drop_block, self.clip_bbox = scale_x_y, spp, drop_block, clip_bbox
        self.anchors, self.anchor_masks = anchors, anchor_masks
        self.downsample, self.in_channels = downsample, in_channels
        self.yolo_loss, self.nms_cfg, self.is_train = yolo_loss, nms_cfg, is_train
        self.block_size, self.keep_prob, self.norm_decay = block_size, keep_prob, norm_decay
        self._anchors = np.array(copy.deepcopy(anchors)).astype(np.float32)
        assert norm_type in ['bn', 'sync_bn', 'gn', 'affine_channel']
        bn, gn, af = get_norm(norm_type)
        n_lvl = len(downsample)
        self.detection_blocks = torch.nn.ModuleList()
        self.yolo_output_convs = torch.nn.ModuleList()
        self.upsample_layers = torch.nn.ModuleList()
        for i in range(n_lvl):
            in_c = in_channels[i] + (512 // (2 ** i) if i > 0 else 0)
            width = 64 * (2 ** n_lvl) // (2 ** i)
            self.detection_blocks.append(DetectionBlock(
                in_c=in_c, channel=width, coord_conv=coord_conv, bn=bn, gn=gn, af=af, norm_decay=norm_decay,
                is_first=(i == 0), conv_block_num=conv_block_num, use_spp=spp, drop_block=drop_block,
                block_size=block_size, keep_prob=keep_prob, is_test=(not is_train),
                name="yolo_block.{}".format(i)))
            nf = len(anchor_masks[i]) * (num_classes + (6 if iou_aware else 5))
            self.yolo_output_convs.append(Conv2dUnit(width * 2, nf, 1, stride=1, bias_attr=True, act=None,
                                                     name="yolo_output.{}.conv".format(i)))
            if i < n_lvl - 1:
                self.upsample_layers.append(Conv2dUnit(width, 256 // (2 ** i), 1, stride=1, bn=bn, gn=gn, af=af,
                                                       act='leaky', norm_decay=norm_decay,
                                                       name="yolo_transition.{}".format(i)))
                self.upsample_layers.append(torch.nn.Upsample(scale_factor=2, mode='nearest'))

    def set_dropblock(self, is_test):
        # reference model/head.py:375-379
        for blk in self.detection_blocks:
            for l in blk.layers:
                if isinstance(l, DropBlock):
                    l.is_test = is_test

    # ---- plan emission -----------------------------------------------------------------------
    def route_channels(self, i):
        """Channels of the up-sampled route concatenated in front of level i's backbone map."""
        return self.upsample_layers[2 * (i - 1)].filters

    def make_out_slots(self, feature_maps):
        """Concat fusion (reference `torch.cat([route, block], dim=1)`, :391): for every level
        i > 0 the backbone writes its feature map straight into channels [route_c, route_c+C)
        of the level's input buffer and the route conv later fills [0, route_c)."""
        n_lvl = len(self.anchor_masks)
        stages = list(feature_maps)[::-1][:n_lvl]       # deepest first, e.g. [5, 4, 3]
        self._concat = {}
        slots = {}
        for i, stage in enumerate(stages):
            if i == 0:
                continue
            rc, C = self.route_channels(i), self.in_channels[i]

            def slot(b, N, H, W, i=i, rc=rc, C=C):
                wide = b.new_act(N, H, W, rc + C)
                self._concat[i] = wide
                return b.slice(wide, rc, C)
            slots[stage] = slot
        return slots

    def emit(self, b, feats):
        n_lvl = len(self.anchor_masks)
        blocks = feats[::-1][:n_lvl]
        outs = []
        for i, blk in enumerate(blocks):
            x = blk if i == 0 else self._concat[i]
            last = (i == n_lvl - 1)
            route, tip = self.detection_blocks[i].emit(b, x, tip_on_side=not last)
            if last:
                outs.append(self.yolo_output_convs[i].emit(b, tip))
            else:
                with b.side():      # tip -> output conv of this level runs beside the next level's chain
                    outs.append(self.yolo_output_convs[i].emit(b, tip))
            if i < n_lvl - 1:
                nxt = self._concat[i + 1]
                rc = self.route_channels(i + 1)
                self.upsample_layers[2 * i].emit(b, route, out=b.slice(nxt, 0, rc), ups=True)
        return outs

    def decode_params(self, outs):
        nms = copy.deepcopy(self.nms_cfg)
        if nms is None:
            raise ValueError('YOLOv3Head needs nms_cfg for inference (reference demo.py:87)')
        nms_type = nms.pop('nms_type')
        if nms_type != 'matrix_nms':
            raise NotImplementedError('only matrix_nms (the reference disables multiclass_nms, head.py:465-468)')
        levels = []
        for i, o in enumerate(outs):
            assert o.H == o.W, 'yolo_box assumes square feature maps (reference head.py:25-27)'
            levels.append(dict(anchors=[[float(v) for v in self._anchors[a]] for a in self.anchor_masks[i]],
                               downsample=int(self.downsample[i])))
        M = sum(o.H * o.W * len(self.anchor_masks[i]) for i, o in enumerate(outs))
        return dict(levels=levels, num_classes=self.num_classes, scale_x_y=self.scale_x_y, iou_aware=self.iou_aware,
                    iou_aware_factor=self.iou_aware_factor, clip_bbox=self.clip_bbox, M_total=M, nms=nms)

    def add_param_group(self, param_groups, base_lr, base_wd):      # reference model/head.py:366-373
        for blk in self.detection_blocks:
            blk.add_param_group(param_groups, base_lr, base_wd)
        for ly in self.yolo_output_convs:
            ly.add_param_group(param_groups, base_lr, base_wd)
        for ly in self.upsample_layers:
            if isinstance(ly, Conv2dUnit):
                ly.add_param_group(param_groups, base_lr, base_wd)

    def get_loss(self, input, gt_box, gt_label, gt_score, targets):
        """The reference computes the loss from backbone feature maps with autograd ops (model/head.py:400-423).  Here the
        training forward, the loss and its backward are HIP kernels driven from the whole model (the frozen backbone runs
        in training mode too): call `PPYOLO.forward(images, None, False, gt_box, gt_label, gt_score, targets)`."""
        raise NotImplementedError('call model(images, None, False, gt_box, gt_label, gt_score, targets): the training forward starts '
                                  'at the images (ppyolo_hip.train), not at torch feature maps')
