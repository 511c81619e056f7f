"""Level-0 tip of the R50vd head at 608 / 8 images: the HIP BatchNorm backward and weight gradient against float64 evaluations of
the SAME formulas on the HIP path's own saved tensors (kernel arithmetic), and HIP's batch statistics against float64 ones."""
import sys
ROOT = '/root/repo'
for p in (ROOT, ROOT + '/pytorch-ppyolo_amd', ROOT + '/tests'):
    sys.path.insert(0, p)
import torch
from config import PPYOLO_2x_Config
from conftest import build_model
from ppyolo_hip import ops as K, synth
from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
from ppyolo_hip.train import TrainStep, Act

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
ts = TrainStep(model, cfg)
ts.tape, ts._nbt = [], []
cap = {}
real_bwd, real_wg = K.bn_train_bwd, K.conv2d_wgrad

def bwd_spy(x, y, dy, mean, invstd, gamma, dx, dgamma, dbeta, act=None, ws=None, amax_dx=None):
    real_bwd(x, y, dy, mean, invstd, gamma, dx, dgamma, dbeta, act, ws, amax_dx)
    if x.C == 1024 and x.H == 19:
        cap.setdefault('bn', []).append(dict(x=x.dense().clone(), y=y.dense().clone(), dy=dy.dense().clone(), mean=mean.clone(), invstd=invstd.clone(),
                                             gamma=gamma.clone(), dx=dx.dense().clone(), dgamma=dgamma.clone(), dbeta=dbeta.clone(), act=act))

def wg_spy(x, dy, dw, stride=1, pad=0, ws=None, amax_x=None, amax_dy=None):
    real_wg(x, dy, dw, stride, pad, ws, amax_x, amax_dy)
    if dy.C == 1024 and dy.H == 19 and dw.shape[1] == 3:
        cap.setdefault('wg', []).append(dict(x=x.dense().clone(), dy=dy.dense().clone(), dw=dw.clone(), pad=pad))
K.bn_train_bwd, K.conv2d_wgrad = bwd_spy, wg_spy
fa = []
for f in feats:
    t = f.permute(0, 2, 3, 1).contiguous().cuda()
    fa.append(Act(t, 0, t.shape[3], False, K.amax_slots(t) if ts.f16 else None))
ts.head_loss_backward(fa, gt.cuda(), [t.cuda() for t in targets])
torch.cuda.synchronize()
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
for i, c in enumerate(cap.get('bn', [])):
    x, y, dy = c['x'].double(), c['y'].double(), c['dy'].double()
    P = x.shape[0] * x.shape[1] * x.shape[2]
    xf = x.reshape(P, -1)
    m64, v64 = xf.mean(0), xf.var(0, unbiased=False)
    is64 = 1.0 / torch.sqrt(v64 + 1e-5)
    slope = {'leaky': 0.1, 'relu': 0.0}.get(c['act'], 1.0)
    dz = (dy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))).reshape(P, -1)

    def formula(mean, invstd):
        xh = (xf - mean) * invstd
        sa, sb = dz.sum(0), (dz * xh).sum(0)
        return c['gamma'].double() * invstd * (dz - (sa + xh * sb) / P), sa, sb
    dx_own, sa, sb = formula(c['mean'].double(), c['invstd'].double())       # HIP's statistics, float64 arithmetic
    dx_64, _, _ = formula(m64, is64)                                          # float64 statistics too
    print('BN bwd %d (act %s): mean rel err %.2e, invstd rel err %.2e | dx vs f64 formula on HIP stats %.2e, vs f64 stats %.2e | dbeta %.2e dgamma %.2e'
          % (i, c['act'], rel(c['mean'], m64), rel(c['invstd'], is64), rel(c['dx'].reshape(P, -1), dx_own), rel(c['dx'].reshape(P, -1), dx_64),
             rel(c['dbeta'], sa), rel(c['dgamma'], sb)))
    print('   |mean|/sigma median %.2f; mean|dz| / rms(dx_64) per channel median %.1f' % (
        float((m64.abs() * is64).median()), float((dz.mean(0).abs() / (dx_64 / (c['gamma'].double() * is64)).pow(2).mean(0).sqrt().clamp_min(1e-300)).median())))
for i, c in enumerate(cap.get('wg', [])):
    x, dy = c['x'].double().permute(0, 3, 1, 2), c['dy'].double().permute(0, 3, 1, 2)
    dw = torch.nn.grad.conv2d_weight(x, (dy.shape[1], x.shape[1], 3, 3), dy, padding=c['pad'])
    print('wgrad %d: vs float64 on the same operands %.2e' % (i, rel(c['dw'].permute(0, 3, 1, 2)[:, :x.shape[1]], dw)))
