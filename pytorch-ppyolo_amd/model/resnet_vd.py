"""ResNet50-vd / ResNet18-vd backbones of the MI355X PP-YOLO path.

Same class names, constructor arguments and parameter names as the reference's
`model/resnet_vd.py` (ConvBlock :15-57, IdentityBlock :60-87, Resnet50Vd :89-168,
BasicBlock :224-267, Resnet18Vd :270-330).  The classes only hold parameters and describe
the graph: `emit()` appends kernel launches to a plan (ppyolo_hip/engine.py).  Fusions done at
emission time: the residual `x + shortcut` and the final ReLU ride in the epilogue of the
branch's last conv; the vd shortcut's AvgPool2d(2,2) feeds the 1x1 projection conv.
Training-only helpers (freeze / add_param_group) are out of scope.
"""
import os

import torch

from model.custom_layers import Conv2dUnit, get_norm


def _norm(norm_type):
    assert norm_type in ['bn', 'sync_bn', 'gn', 'affine_channel']
    return get_norm(norm_type)



class _Units(object):
    """freeze / add_param_group over the Conv2dUnits of a block, in the reference's order (model/resnet_vd.py:36-46, :71-79)."""

    def _units(self):
        return [u for u in (getattr(self, 'conv%d' % i, None) for i in (1, 2, 3, 4)) if u is not None]

    def freeze(self):
        for u in self._units():
            u.freeze()

    def add_param_group(self, param_groups, base_lr, base_wd):
        for u in self._units():
            u.add_param_group(param_groups, base_lr, base_wd)


class ConvBlock(_Units, torch.nn.Module):
    def __init__(self, in_c, filters, bn, gn, af, freeze_norm, norm_decay, lr, use_dcn=False, stride=2,
                 downsample_in3x3=True, is_first=False, block_name=''):
        super(ConvBlock, self).__init__()
        f1, f2, f3 = filters
        s1, s2 = (1, stride) if downsample_in3x3 else (stride, 1)
        self.is_first = is_first
        kw = dict(bn=bn, gn=gn, af=af, freeze_norm=freeze_norm, norm_decay=norm_decay, lr=lr)
        self.conv1 = Conv2dUnit(in_c, f1, 1, stride=s1, act='relu', name=block_name + '_branch2a', **kw)
        self.conv2 = Conv2dUnit(f1, f2, 3, stride=s2, act='relu', use_dcn=use_dcn, name=block_name + '_branch2b', **kw)
        self.conv3 = Conv2dUnit(f2, f3, 1, stride=1, act=None, name=block_name + '_branch2c', **kw)
        # vd projection shortcut: avg-pool then 1x1 (stride 1) unless this is the first stage
        self.conv4 = Conv2dUnit(in_c, f3, 1, stride=stride if is_first else 1, act=None,
                                name=block_name + '_branch1', **kw)

    # Round 4: the projection shortcut folded into conv3.  relu(bn3(W3 z) + bn4(W4 s)) is ONE 1x1 convolution over the channel
    # concatenation [z | s] with the weights [diag(scale3) W3 | diag(scale4) W4], shift3 + shift4 and a ReLU: conv2 writes z into
    # the first channels of a wide buffer, the vd average pool (or, in stage 2, the stem's max pool) writes s behind it, and the
    # shortcut tensor -- written by one launch and read back by the next as the residual, 2 x 189 MB at 152 x 152 -- never
    # exists.  One launch less per stage; the reduction of conv3 grows from f2 to f2 + in_c channels, same FLOPs in total.
    @staticmethod
    def _fold(b):
        """The switch is read ONCE per plan (Builder), so wide_input() and emit() of one plan cannot disagree."""
        f = getattr(b, 'fold_shortcut', None)
        if f is None:
            f = b.fold_shortcut = os.environ.get('PPYOLO_HIP_FOLD_SHORTCUT', '1') == '1'
        return f

    def wide_input(self, b, N, H, W):
        """Stage 2 (no pooling in front of the projection): the block's INPUT is the shortcut operand, so its producer writes it
        straight into the wide buffer -> the slice to hand to that producer (None when the fold is off)."""
        if not (self._fold(b) and self.is_first):
            return None
        f2, cs = self.conv2.filters, self.conv4.conv.in_channels
        wide = b.new_buf(N, H, W, f2 + cs)
        b.wide_of_block[id(self)] = wide          # (kept on the builder: a plan under construction, not module state)
        from ppyolo_hip.engine import A
        return A(wide, f2, cs, N, H, W)

    def emit(self, b, x, out=None):
        # (a stage-2 block whose input was NOT placed into a wide buffer by its producer -- stand-alone emission, another stem -- takes
        # the unfolded path: the fold needs the producer's cooperation there, wide_input())
        if self._fold(b) and not (self.is_first and not (id(self) in b.wide_of_block and x.buf == b.wide_of_block[id(self)])):
            return self._emit_folded(b, x, out)
        b.wide_of_block.pop(id(self), None)
        with b.side():              # projection shortcut: independent of conv1 -> conv2
            s = x if self.is_first else b.avgpool(x)
            s = self.conv4.emit(b, s)
        y = self.conv1.emit(b, x)
        y = self.conv2.emit(b, y)
        y = self.conv3.emit(b, y, res=s, out=out, post_act='relu')      # relu(bn(conv) + shortcut)
        return y

    def _emit_folded(self, b, x, out):
        from ppyolo_hip.engine import A
        f2, cs, f3 = self.conv2.filters, self.conv4.conv.in_channels, self.conv3.filters
        Ho, Wo = (x.H, x.W) if self.is_first else (x.H // 2, x.W // 2)
        wide = b.wide_of_block.pop(id(self), None)
        if self.is_first:      # x already sits in its slice of the wide buffer (wide_input; emit() checked it)
            assert wide is not None and x.buf == wide
        else:
            wide = b.new_buf(x.N, Ho, Wo, f2 + cs)
            b.avgpool(x, out=A(wide, f2, cs, x.N, Ho, Wo))
        y = self.conv1.emit(b, x)
        self.conv2.emit(b, y, out=A(wide, 0, f2, x.N, Ho, Wo))
        skel = getattr(b, 'skeleton', False)
        if skel:
            w = torch.empty((f3, f2 + cs, 1, 1), dtype=torch.float32, device='meta')
            one = shift = None
        else:
            s3, b3 = self.conv3.folded(b.device)
            s4, b4 = self.conv4.folded(b.device)
            w = torch.cat([self.conv3.conv.weight.detach().float() * s3.view(-1, 1, 1, 1),
                           self.conv4.conv.weight.detach().float() * s4.view(-1, 1, 1, 1)], dim=1)
            one, shift = torch.ones_like(s3), b3 + b4
        return b.conv(A(wide, 0, f2 + cs, x.N, Ho, Wo), w, one, shift, stride=1, act='relu', out=out)


class IdentityBlock(_Units, torch.nn.Module):
    def __init__(self, in_c, filters, bn, gn, af, freeze_norm, norm_decay, lr, use_dcn=False, block_name=''):
        super(IdentityBlock, self).__init__()
        f1, f2, f3 = filters
        kw = dict(bn=bn, gn=gn, af=af, freeze_norm=freeze_norm, norm_decay=norm_decay, lr=lr)
        self.conv1 = Conv2dUnit(in_c, f1, 1, stride=1, act='relu', name=block_name + '_branch2a', **kw)
        self.conv2 = Conv2dUnit(f1, f2, 3, stride=1, act='relu', use_dcn=use_dcn, name=block_name + '_branch2b', **kw)
        self.conv3 = Conv2dUnit(f2, f3, 1, stride=1, act=None, name=block_name + '_branch2c', **kw)

    def emit(self, b, x, out=None):
        y = self.conv1.emit(b, x)
        y = self.conv2.emit(b, y)
        return self.conv3.emit(b, y, res=x, out=out, post_act='relu')


class _Backbone(torch.nn.Module):
    """Shared stem + stage walking.  `out_slots` lets the head ask for a feature map to be
    written straight into the channel slice of its concat buffer."""

    def _emit_stem(self, b):
        x = self.stage1_conv1_1.emit(b, None)
        x = self.stage1_conv1_2.emit(b, x)
        x = self.stage1_conv1_3.emit(b, x)
        first = self._stage_blocks(2)[0]
        slot = first.wide_input(b, x.N, (x.H + 1) // 2, (x.W + 1) // 2) if hasattr(first, 'wide_input') else None
        return b.maxpool(x, out=slot)

    def emit(self, b, out_slots=None):
        """out_slots: {stage: callable(builder, N, H, W) -> A}: feature maps the head wants
        produced in place inside its concat buffer (the last block of every stage is stride 1,
        so its output has its input's spatial size).  Returns the feature-map activations."""
        out_slots = out_slots or {}
        x = self._emit_stem(b)
        feats = {}
        for stage in (2, 3, 4, 5):
            blocks = self._stage_blocks(stage)
            for i, blk in enumerate(blocks):
                out = None
                if i == len(blocks) - 1 and stage in out_slots:
                    out = out_slots[stage](b, x.N, x.H, x.W)
                x = blk.emit(b, x, out=out)
            feats[stage] = x
        return [feats[s] for s in (2, 3, 4, 5) if s in self.feature_maps]

    def forward(self, x):
        from ppyolo_hip.runtime import run_backbone
        return run_backbone(self, x)

    # ---- training-loop surface (reference model/resnet_vd.py:174-222) ----
    def _stage_units(self, stage):
        if stage == 1:
            return [self.stage1_conv1_1, self.stage1_conv1_2, self.stage1_conv1_3]
        return self._stage_blocks(stage)

    def freeze(self):
        """Stages 1 .. freeze_at stop training (their BatchNorm keeps normalising with batch statistics: the reference's
        loop never leaves train mode)."""
        for stage in range(1, self.freeze_at + 1):
            for u in self._stage_units(stage):
                u.freeze()

    def add_param_group(self, param_groups, base_lr, base_wd):
        for stage in (1, 2, 3, 4, 5):
            for u in self._stage_units(stage):
                u.add_param_group(param_groups, base_lr, base_wd)


class Resnet50Vd(_Backbone):
    def __init__(self, norm_type='bn', feature_maps=[3, 4, 5], dcn_v2_stages=[5], downsample_in3x3=True, freeze_at=0,
                 freeze_norm=False, norm_decay=0., lr_mult_list=[1., 1., 1., 1.]):
        super(Resnet50Vd, self).__init__()
        self.norm_type, self.feature_maps, self.freeze_at = norm_type, list(feature_maps), freeze_at
        assert freeze_at in [0, 1, 2, 3, 4, 5]
        assert len(lr_mult_list) == 4
        bn, gn, af = _norm(norm_type)
        kw = dict(bn=bn, gn=gn, af=af, freeze_norm=freeze_norm, norm_decay=norm_decay)
        self.stage1_conv1_1 = Conv2dUnit(3, 32, 3, stride=2, act='relu', name='conv1_1', **kw)
        self.stage1_conv1_2 = Conv2dUnit(32, 32, 3, stride=1, act='relu', name='conv1_2', **kw)
        self.stage1_conv1_3 = Conv2dUnit(32, 64, 3, stride=1, act='relu', name='conv1_3', **kw)
        args = (bn, gn, af, freeze_norm, norm_decay)
        width = {2: (64, [64, 64, 256]), 3: (256, [128, 128, 512]), 4: (512, [256, 256, 1024]),
                 5: (1024, [512, 512, 2048])}
        depth = {2: 3, 3: 4, 4: 6, 5: 3}
        self._depth = depth
        for stage in (2, 3, 4, 5):
            in_c, filt = width[stage]
            lr = lr_mult_list[stage - 2]
            dcn = stage in dcn_v2_stages
            tag = 'res%d' % stage
            first = ConvBlock(in_c, filt, *args, lr, use_dcn=dcn, stride=1 if stage == 2 else 2,
                              downsample_in3x3=downsample_in3x3, is_first=(stage == 2), block_name=tag + 'a')
            setattr(self, 'stage%d_0' % stage, first)
            for i in range(1, depth[stage]):
                setattr(self, 'stage%d_%d' % (stage, i),
                        IdentityBlock(filt[2], filt, *args, lr, use_dcn=dcn, block_name=tag + 'abcdef'[i]))

    def _stage_blocks(self, stage):
        return [getattr(self, 'stage%d_%d' % (stage, i)) for i in range(self._depth[stage])]


class BasicBlock(_Units, torch.nn.Module):
    def __init__(self, in_c, filters, bn, gn, af, freeze_norm, norm_decay, lr, stride=1, is_first=False,
                 block_name=''):
        super(BasicBlock, self).__init__()
        f1, f2 = filters
        self.is_first, self.stride = is_first, stride
        kw = dict(bn=bn, gn=gn, af=af, freeze_norm=freeze_norm, norm_decay=norm_decay, lr=lr)
        self.conv1 = Conv2dUnit(in_c, f1, 3, stride=stride, act='relu', name=block_name + '_branch2a', **kw)
        self.conv2 = Conv2dUnit(f1, f2, 3, stride=1, act=None, name=block_name + '_branch2b', **kw)
        self.conv3 = None
        if stride == 2 or is_first:
            self.conv3 = Conv2dUnit(in_c, f2, 1, stride=stride if is_first else 1, act=None,
                                    name=block_name + '_branch1', **kw)

    def emit(self, b, x, out=None):
        if self.conv3 is not None:
            with b.side():
                s = x if self.is_first else b.avgpool(x)
                s = self.conv3.emit(b, s)
        else:
            s = x
        y = self.conv1.emit(b, x)
        return self.conv2.emit(b, y, res=s, out=out, post_act='relu')


class Resnet18Vd(_Backbone):
    def __init__(self, norm_type='bn', feature_maps=[4, 5], dcn_v2_stages=[], freeze_at=0, freeze_norm=False,
                 norm_decay=0., lr_mult_list=[1., 1., 1., 1.]):
        super(Resnet18Vd, self).__init__()
        self.norm_type, self.feature_maps, self.freeze_at = norm_type, list(feature_maps), freeze_at
        assert freeze_at in [0, 1, 2, 3, 4, 5]
        assert len(lr_mult_list) == 4
        if dcn_v2_stages:
            raise NotImplementedError('the reference Resnet18Vd ignores dcn_v2_stages (model/resnet_vd.py:270-300)')
        bn, gn, af = _norm(norm_type)
        kw = dict(bn=bn, gn=gn, af=af, freeze_norm=freeze_norm, norm_decay=norm_decay)
        self.stage1_conv1_1 = Conv2dUnit(3, 32, 3, stride=2, act='relu', name='conv1_1', **kw)
        self.stage1_conv1_2 = Conv2dUnit(32, 32, 3, stride=1, act='relu', name='conv1_2', **kw)
        self.stage1_conv1_3 = Conv2dUnit(32, 64, 3, stride=1, act='relu', name='conv1_3', **kw)
        args = (bn, gn, af, freeze_norm, norm_decay)
        chans = {2: (64, 64), 3: (64, 128), 4: (128, 256), 5: (256, 512)}
        for stage in (2, 3, 4, 5):
            in_c, c = chans[stage]
            lr = lr_mult_list[stage - 2]
            setattr(self, 'stage%d_0' % stage,
                    BasicBlock(in_c, [c, c], *args, lr, stride=1 if stage == 2 else 2, is_first=(stage == 2),
                               block_name='res%da' % stage))
            setattr(self, 'stage%d_1' % stage, BasicBlock(c, [c, c], *args, lr, stride=1, block_name='res%db' % stage))

    def _stage_blocks(self, stage):
        return [getattr(self, 'stage%d_0' % stage), getattr(self, 'stage%d_1' % stage)]
