"""Where does the 2.2e-3 error of the level-0 tip's weight gradient (R50vd-608, 8 images, head on identical features) enter?
Captures d loss / d (unit output) for every head unit in the HIP tape and in the fp32 / float64 oracles."""
import sys
ROOT = '/root/repo'
for p in (ROOT, ROOT + '/pytorch-ppyolo_amd', ROOT + '/tests'):
    sys.path.insert(0, p)
import numpy as np
import torch
from config import PPYOLO_2x_Config
from conftest import build_model
from oracle import ppyolo_oracle as orc, train_oracle as trn
from ppyolo_hip import ops as K, synth
from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
from ppyolo_hip.train import TrainStep, Act
import train_parity_util as tp

cfg = PPYOLO_2x_Config()
N, S = 8, 608
model, sd = build_model(cfg, 0, 'cuda')
cfg.head['drop_active'] = False
g = torch.Generator().manual_seed(21)
feats = [torch.relu(torch.randn(N, c, S // d, S // d, generator=g)) for c, d in ((512, 8), (1024, 16), (2048, 32))]
bb, cc, ss = synth_ground_truth(N, 50)
hc = cfg.head
gt = torch.from_numpy(bb)
targets = [torch.from_numpy(t) for t in gt2yolo_target(bb, cc, ss, hc['anchors'], hc['anchor_masks'], hc['downsample'], 80, S)]
torch.set_num_threads(16)
real_cu, real_db = orc.conv_unit, orc.drop_block_train
orc.drop_block_train = lambda t, *a, **k: t
res = {}
for tag, dt in (('ref32', torch.float32), ('ref64', torch.float64)):
    state = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in trn.trainable_keys(sd):
        state[k].requires_grad_(True)
    rec = {}

    def spy(sd_, prefix, x_, stride=1, act=None, rec=rec):
        y = real_cu(sd_, prefix, x_, stride, act)
        if y.requires_grad:
            y.retain_grad()
            rec[prefix] = y
        return y
    orc.conv_unit = spy
    orc.TRAIN_MODE[0] = True
    try:
        outs = orc.head_outputs(state, [f.to(dt) for f in feats], cfg.head)
    finally:
        orc.TRAIN_MODE[0] = False
        orc.conv_unit = real_cu
    for o in outs:
        o.retain_grad()
    losses = trn.yolov3_loss(outs, [t.to(dt) for t in targets], gt.to(dt), cfg)
    sum(losses.values()).backward()
    res[tag] = dict(g={k: v.grad for k, v in rec.items()}, w={k: state[k].grad for k in trn.trainable_keys(sd)})
orc.drop_block_train = real_db
ts = TrainStep(model, cfg)
ts.tape, ts._nbt = [], []
hip_acts = {}
real_unit = ts.conv_unit

def unit_spy(prefix, *a, **k):
    y = real_unit(prefix, *a, **k)
    hip_acts[prefix] = y
    return y
ts.conv_unit = unit_spy
fa = []
for f in feats:
    t = f.permute(0, 2, 3, 1).contiguous().cuda()
    fa.append(Act(t, 0, t.shape[3], False, K.amax_slots(t) if ts.f16 else None))
ts.head_loss_backward(fa, gt.cuda(), [t.cuda() for t in targets])
torch.cuda.synchronize()
grads = ts.grads()
print('unit: d loss / d output (relative L2 vs float64)   |   conv weight gradient')
for k in res['ref64']['g']:
    a = hip_acts.get(k)
    if a is None or a.g is None:
        continue
    e_h = tp.rel_l2(a.g.dense_nchw(), res['ref64']['g'][k])
    e_r = tp.rel_l2(res['ref32']['g'][k], res['ref64']['g'][k])
    wk = k + '.conv.weight'
    print('%-45s dy: hip %.2e ref %.2e   dw: hip %.2e ref %.2e' % (k, e_h, e_r, tp.rel_l2(grads[wk], res['ref64']['w'][wk]),
                                                                  tp.rel_l2(res['ref32']['w'][wk], res['ref64']['w'][wk])))
