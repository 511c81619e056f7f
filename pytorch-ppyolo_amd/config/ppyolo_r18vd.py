"""PP-YOLO tiny (ResNet18-vd, no DCN / CoordConv / SPP / IoU-aware) inference
hyper-parameters.  Data-only mirror of the reference's
`config/ppyolo_r18vd.py:13-226` (`PPYOLO_r18vd_Config`), inference attributes only.
"""
from .ppyolo_2x import _matrix_nms_defaults


class PPYOLO_r18vd_Config(object):
    def __init__(self):
        self.num_classes = 80
        self.classes_path = 'data/coco_classes.txt'
        self.eval_cfg = dict(model_path='ppyolo_r18vd.pt', target_size=416,
                             draw_image=False, draw_thresh=0.15, eval_batch_size=4)
        self.test_cfg = dict(model_path='ppyolo_r18vd.pt', target_size=416,
                             draw_image=True, draw_thresh=0.15)
        # reference: config/ppyolo_r18vd.py:94-101
        self.backbone_type = 'Resnet18Vd'
        self.backbone = dict(norm_type='bn', feature_maps=[4, 5], dcn_v2_stages=[],
                             freeze_at=5, freeze_norm=False, norm_decay=0.)
        # reference: config/ppyolo_r18vd.py:103-120
        self.head_type = 'YOLOv3Head'
        self.head = dict(num_classes=self.num_classes, conv_block_num=0, norm_type='bn',
                         anchor_masks=[[3, 4, 5], [0, 1, 2]],
                         anchors=[[10, 14], [23, 27], [37, 58],
                                  [81, 82], [135, 169], [344, 319]],
                         coord_conv=False, iou_aware=False, iou_aware_factor=0.4,
                         scale_x_y=1.05, spp=False, drop_block=True, keep_prob=0.9,
                         downsample=[32, 16], in_channels=[512, 256])
        # training step -- reference: config/ppyolo_r18vd.py:46-66, :122-134
        self.iou_loss_type = 'IouLoss'
        self.iou_loss = dict(loss_weight=2.5, max_height=608, max_width=608, ciou_term=False)
        self.use_ema = True            # reference config: lines 92-94
        self.ema_decay = 0.9998
        self.yolo_loss_type = 'YOLOv3Loss'
        self.yolo_loss = dict(ignore_thresh=0.7, scale_x_y=1.05, label_smooth=False, use_fine_grained_loss=True)
        self.learningRate = dict(base_lr=0.0001, PiecewiseDecay=dict(gamma=0.1, milestones=[150000, 200000]),
                                 LinearWarmup=dict(start_factor=0., steps=4000))
        self.optimizerBuilder = dict(optimizer=dict(momentum=0.9, type='Momentum'), regularizer=dict(factor=0.0005, type='L2'))
        self.nms_cfg = _matrix_nms_defaults()
        self.context = {'fields': ['image']}
        self.decodeImage = dict(to_rgb=True)
        self.normalizeImage = dict(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225],
                                   is_scale=True, is_channel_first=False)
        self.permute = dict(to_bgr=False, channel_first=True)
        self.resizeImage = dict(target_size=416, interp=2)
