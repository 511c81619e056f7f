"""PP-YOLO (ResNet50-vd + DCNv2) inference hyper-parameters.

Data-only mirror of the reference's `config/ppyolo_2x.py:13-234` (class
`PPYOLO_2x_Config`): only the attributes the inference hot path reads are kept
(`backbone_type/backbone`, `head_type/head`, `nms_cfg`, `eval_cfg/test_cfg`,
`num_classes`).  Training / data-augmentation attributes are out of scope
(SURVEY.md section 8f).
"""

_COCO_ANCHORS_9 = [[10, 13], [16, 30], [33, 23], [30, 61], [62, 45],
                   [59, 119], [116, 90], [156, 198], [373, 326]]


def _matrix_nms_defaults():
    # reference: config/ppyolo_2x.py:143-151
    return dict(nms_type='matrix_nms', score_threshold=0.01, post_threshold=0.01,
                nms_top_k=500, keep_top_k=100, use_gaussian=False, gaussian_sigma=2.)


class PPYOLO_2x_Config(object):
    def __init__(self):
        self.num_classes = 80
        self.classes_path = 'data/coco_classes.txt'
        self.eval_cfg = dict(model_path='ppyolo_2x.pt', target_size=608,
                             draw_image=False, draw_thresh=0.15, eval_batch_size=4)
        self.test_cfg = dict(model_path='ppyolo_2x.pt', target_size=608,
                             draw_image=True, draw_thresh=0.15)
        # reference: config/ppyolo_2x.py:95-104
        self.backbone_type = 'Resnet50Vd'
        self.backbone = dict(norm_type='bn', feature_maps=[3, 4, 5], dcn_v2_stages=[5],
                             downsample_in3x3=True, freeze_at=5, freeze_norm=False,
                             norm_decay=0.)
        # reference: config/ppyolo_2x.py:105-123
        self.head_type = 'YOLOv3Head'
        self.head = dict(num_classes=self.num_classes, norm_type='bn',
                         anchor_masks=[[6, 7, 8], [3, 4, 5], [0, 1, 2]],
                         anchors=[list(a) for a in _COCO_ANCHORS_9],
                         coord_conv=True, iou_aware=True, iou_aware_factor=0.4,
                         scale_x_y=1.05, spp=True, drop_block=True, keep_prob=0.9,
                         downsample=[32, 16, 8], in_channels=[2048, 1024, 512])
        # training step -- reference: config/ppyolo_2x.py:47-67, :123-142
        self.iou_loss_type = 'IouLoss'
        self.iou_loss = dict(loss_weight=2.5, max_height=608, max_width=608, ciou_term=False)
        self.iou_aware_loss_type = 'IouAwareLoss'
        self.iou_aware_loss = dict(loss_weight=1.0, max_height=608, max_width=608)
        self.use_ema = True            # reference config: lines 92-94
        self.ema_decay = 0.9998
        self.yolo_loss_type = 'YOLOv3Loss'
        self.yolo_loss = dict(ignore_thresh=0.7, scale_x_y=1.05, label_smooth=False, use_fine_grained_loss=True)
        self.learningRate = dict(base_lr=0.0001, PiecewiseDecay=dict(gamma=0.1, milestones=[400000, 450000]),
                                 LinearWarmup=dict(start_factor=0., steps=4000))
        self.optimizerBuilder = dict(optimizer=dict(momentum=0.9, type='Momentum'), regularizer=dict(factor=0.0005, type='L2'))
        self.nms_cfg = _matrix_nms_defaults()
        # pre-processing constants the harness (decode_np.Decode) reads
        self.context = {'fields': ['image']}
        self.decodeImage = dict(to_rgb=True)
        self.normalizeImage = dict(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225],
                                   is_scale=True, is_channel_first=False)
        self.permute = dict(to_bgr=False, channel_first=True)
        self.resizeImage = dict(target_size=608, interp=2)
