"""debug: activations / gradients layer by layer, TrainStep vs the oracle (R50vd 320 px, freeze_at 3, DropBlock off)"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from conftest import build_model
from config import PPYOLO_2x_Config
from oracle import train_oracle as trn, ppyolo_oracle as orc
from ppyolo_hip import synth
from ppyolo_hip.train import TrainStep
g = np.load(os.path.join(ROOT, 'tests/golden/g16_train_r50vd_320_fa3.npz'))
S, N, wseed, iseed, fa = [int(v) for v in g['meta']]
cfg = PPYOLO_2x_Config()
model, sd = build_model(cfg, wseed, 'cuda')
cfg.backbone['freeze_at'] = fa
cfg.head['drop_active'] = False
orc.drop_block_train = lambda x, *a, **k: x
T = lambda a: torch.from_numpy(np.asarray(a))
x = synth.synth_images(N, S, seed=iseed)
targets = [T(g['target%d' % i]) for i in range(3)]
torch.set_num_threads(16)
rec = {}
real = orc.conv_unit
def spy(s_, prefix, xx, stride=1, act=None):
    y = real(s_, prefix, xx, stride, act)
    if y.requires_grad:
        y.retain_grad()
    rec[prefix] = y
    return y
orc.conv_unit = spy
r = trn.train_step(sd, cfg, x, T(g['gt_bbox']), targets)
orc.conv_unit = real
ts = TrainStep(model, cfg)
ts.acts = {}
loss6 = ts.forward_backward(x.cuda(), T(g['gt_bbox']).cuda(), [t.cuda() for t in targets])
torch.cuda.synchronize()
print('loss mine', loss6.cpu().numpy())
print('loss ref ', [round(float(v), 5) for v in r['losses'].values()])
for k, a in ts.acts.items():
    if k not in rec or not (k.startswith('head.') or k.startswith('backbone.stage5')):
        continue
    o = rec[k]
    mine = a.dense_nchw().cpu()
    e = float((mine - o.detach()).abs().max() / o.detach().abs().max())
    line = '%-46s act err %.1e' % (k, e)
    if a.g is not None and o.grad is not None:
        mg = a.g.dense_nchw().cpu()
        line += '   grad norm mine/ref %.4f  cos %.4f' % (float(mg.norm() / o.grad.norm()),
                                                        float(torch.nn.functional.cosine_similarity(mg.reshape(1, -1).double(), o.grad.reshape(1, -1).double())))
    print(line)
A = 3
for i in range(3):
    mine = ts.outs[i].g.dense_nchw().cpu()
    ref = r['douts'][i]
    bm, br = mine[:, A:].reshape(N, A, 85, *mine.shape[2:]), ref[:, A:].reshape(N, A, 85, *ref.shape[2:])
    print('level %d dout: norm ratio %.4f' % (i, float(mine.norm() / ref.norm())))
    print('   ioup norms %.4f / %.4f' % (float(mine[:, :A].norm()), float(ref[:, :A].norm())))
    for nm, sl in (('xy', slice(0, 2)), ('wh', slice(2, 4)), ('obj', slice(4, 5)), ('cls', slice(5, 85))):
        a, b = bm[:, :, sl], br[:, :, sl]
        print('   %-4s norms %.4f / %.4f   max abs diff %.3e at %s' % (nm, float(a.norm()), float(b.norm()), float((a - b).abs().max()),
                                                                      tuple(int(v) for v in np.unravel_index(int((a - b).abs().argmax()), a.shape))))
    if i == 0:
        d = (bm - br).abs()
        idx = np.unravel_index(int(d.argmax()), d.shape)
        n_, a_, c_, h_, w_ = [int(v) for v in idx]
        mo = ts.outs[0].dense_nchw().cpu()[:, A:].reshape(N, A, 85, *mine.shape[2:])
        ro = r['outs'][0].detach()[:, A:].reshape(N, A, 85, *mine.shape[2:])
        print('   worst cell', idx, 'outputs mine', mo[n_, a_, :5, h_, w_].numpy(), 'ref', ro[n_, a_, :5, h_, w_].numpy())
        print('   dout mine', bm[n_, a_, :5, h_, w_].numpy(), 'ref', br[n_, a_, :5, h_, w_].numpy())
        print('   target', targets[0][n_, a_, :6, h_, w_].numpy())
