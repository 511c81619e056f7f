"""Layer containers of the MI355X PP-YOLO path.

Same class names, constructor signatures and `state_dict` keys as the reference's
`model/custom_layers.py` (Conv2dUnit :65-139, CoordConv :256-272, SPP :275-290,
DropBlock :293-342, DCNv2 :486-523) so checkpoints and calling code are interchangeable --
but the classes here hold PARAMETERS ONLY.  Nothing is computed by torch: `emit()` records
the layer into a kernel plan (ppyolo_hip/engine.py) and `forward()` of a single layer
builds and runs a one-layer plan through the HIP kernels (used by the per-layer parity
tests).  Eval mode only; GroupNorm / AffineChannel / Mish are not instantiated by either
PP-YOLO config and raise NotImplementedError.
"""
import torch

from ppyolo_hip import engine as _engine


def get_norm(norm_type):
    # reference model/custom_layers.py:22-34 ('sync_bn' is plain BN there too)
    bn = 1 if norm_type in ('bn', 'sync_bn') else 0
    gn = 1 if norm_type == 'gn' else 0
    af = 1 if norm_type == 'affine_channel' else 0
    return bn, gn, af


class DCNv2(torch.nn.Module):
    """Parameter holder of the deformable conv (reference :486-523): `conv_offset`
    (3x3, 27 channels, bias) + `dcn_weight` [K,C,3,3] (+ optional `dcn_bias`)."""

    def __init__(self, input_dim, filters, filter_size, stride=1, padding=0, bias_attr=False,
                 distribution='normal', gain=1):
        super(DCNv2, self).__init__()
        if filter_size != 3 or padding != 1:
            raise NotImplementedError('HIP DCNv2 supports the 3x3 / pad 1 form PP-YOLO uses')
        self.input_dim, self.filters, self.filter_size = input_dim, filters, filter_size
        self.stride, self.padding = stride, padding
        self.conv_offset = torch.nn.Conv2d(input_dim, filter_size * filter_size * 3, kernel_size=filter_size,
                                           stride=stride, padding=padding, bias=True)
        torch.nn.init.constant_(self.conv_offset.weight, 0.0)
        torch.nn.init.constant_(self.conv_offset.bias, 0.0)
        self.dcn_weight = torch.nn.Parameter(torch.randn(filters, input_dim, filter_size, filter_size))
        torch.nn.init.xavier_normal_(self.dcn_weight, gain=gain)
        self.dcn_bias = None
        if bias_attr:
            self.dcn_bias = torch.nn.Parameter(torch.zeros(filters))


class Conv2dUnit(torch.nn.Module):
    def __init__(self, input_dim, filters, filter_size, stride=1, bias_attr=False, bn=0, gn=0, af=0, groups=32,
                 act=None, freeze_norm=False, is_test=False, norm_decay=0., lr=1., bias_lr=None, weight_init=None,
                 bias_init=None, use_dcn=False, name=''):
        super(Conv2dUnit, self).__init__()
        if gn or af:
            raise NotImplementedError('GroupNorm / AffineChannel are outside the PP-YOLO inference path')
        if act not in (None, 'relu', 'leaky'):
            raise NotImplementedError("Activation '%s' is not on the PP-YOLO inference path" % act)
        self.filters, self.filter_size, self.stride = filters, filter_size, stride
        self.padding = (filter_size - 1) // 2
        self.act_name = act
        self.use_dcn = use_dcn
        self.name = name
        self.freeze_norm, self.norm_decay, self.lr = freeze_norm, norm_decay, lr      # (reference :94-100)
        if bias_attr and not use_dcn:
            self.blr = bias_lr if bias_lr else lr                                     # (reference :112-116)
        if use_dcn:
            self.conv = DCNv2(input_dim, filters, filter_size=filter_size, stride=stride,
                              padding=(filter_size - 1) // 2, bias_attr=False)
        else:
            self.conv = torch.nn.Conv2d(input_dim, filters, kernel_size=filter_size, stride=stride,
                                        padding=(filter_size - 1) // 2, bias=bias_attr)
        self.bn = torch.nn.BatchNorm2d(filters) if bn else None

    # ---- training-loop surface (reference :141-243): which tensors train, and their optimizer groups ----
    def _own_params(self):
        """[(parameter, lr multiplier, weight decay applies)] in the reference's group order."""
        if self.use_dcn:
            c = self.conv
            ps = [(c.conv_offset.weight, self.lr, True), (c.conv_offset.bias, self.lr, True), (c.dcn_weight, self.lr, True)]
        else:
            ps = [(self.conv.weight, self.lr, True)]
            if self.conv.bias is not None:
                ps.append((self.conv.bias, self.blr, False))
        if self.bn is not None:
            ps += [(self.bn.weight, self.lr, False), (self.bn.bias, self.lr, False)]
        return ps

    def freeze(self):
        for prm, _, _ in self._own_params():
            prm.requires_grad = False
        if self.use_dcn and self.conv.dcn_bias is not None:
            self.conv.dcn_bias.requires_grad = False

    def add_param_group(self, param_groups, base_lr, base_wd):
        """One group per trainable tensor: lr = base_lr * multiplier, L2 decay on convolution weights only (norm scales /
        offsets and convolution biases: 0) -- what the reference hands to torch.optim.SGD (train.py:270-279)."""
        for prm, mult, decay in self._own_params():
            if prm.requires_grad:
                param_groups.append({'params': [prm], 'lr': base_lr * mult, 'base_lr': base_lr * mult,
                                     'weight_decay': base_wd if decay else 0.0})

    # ---- plan emission -----------------------------------------------------------------------
    def folded(self, device):
        bias = self.conv.dcn_bias if self.use_dcn else self.conv.bias
        return _engine.fold_bn(self.bn, bias, self.filters, device)

    def emit(self, b, x, res=None, out=None, ups=False, coord=False, post_act=None):
        """Record conv -> BN -> [+res] -> act into builder `b`; returns the output activation.
        `post_act` is the block-level activation applied after the residual add (the unit's own
        act is None there: reference model/resnet_vd.py:27, :55-56)."""
        act = self.act_name
        if res is not None:
            assert act is None
            act = post_act
        skel = getattr(b, 'skeleton', False)      # shape-only plan: the executor takes the folded weights from their owner
        scale, shift = (None, None) if skel else self.folded(b.device)
        if x is None:
            # first backbone conv: reads the plan's NCHW input directly (3 -> K, 3x3, stride 2)
            assert (self.conv.in_channels, self.filter_size, self.stride) == (3, 3, 2) and not self.use_dcn
            return b.stem(self.conv.weight, scale, shift, self.act_name)
        if self.use_dcn:
            assert res is None and not ups and not coord
            co = self.conv.conv_offset
            one = None if skel else torch.ones(27, dtype=torch.float32, device=b.device)
            om = b.conv(x, co.weight, one, None if skel else co.bias.detach().float().clone(), stride=self.stride, act=None)
            return b.dcn(x, om, self.conv.dcn_weight, scale, shift, self.stride, self.act_name, out=out)
        return b.conv(x, self.conv.weight, scale, shift, stride=self.stride, act=act, res=res, out=out,
                      ups=ups, coord=coord)

    def forward(self, x):
        """Single-layer HIP execution on an NCHW tensor (parity tests / drop-in use of one
        layer).  The whole-model path never calls this."""
        return _engine.run_single(self, x)


class CoordConv(torch.nn.Module):
    """Marker for the CoordConv concat (reference :256-272).  The plan folds the two appended
    coordinate channels into a per-position bias of the NEXT conv (engine.Builder._coord_bias),
    so there is nothing to execute here."""

    def __init__(self, coord_conv=True):
        super(CoordConv, self).__init__()
        self.coord_conv = coord_conv


class SPP(torch.nn.Module):
    """Marker for SPP (reference :275-290, 'asc' order [x, pool5, pool9, pool13])."""

    def __init__(self, seq='asc'):
        super(SPP, self).__init__()
        if seq != 'asc':
            raise NotImplementedError('only the asc order used by PP-YOLO')
        self.seq = seq


class DropBlock(torch.nn.Module):
    """Identity at inference (reference :304-305); in the training step the masks come from ppy_dropblock_mask_f32
    (ppyolo_hip/train.py)."""

    def __init__(self, block_size=3, keep_prob=0.9, is_test=False):
        super(DropBlock, self).__init__()
        self.block_size, self.keep_prob, self.is_test = block_size, keep_prob, is_test
