"""debug: per-tensor gradient-norm ratio (this package / reference golden g16) for R50vd freeze_at 3 at 320 px"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from conftest import build_train_model
from config import PPYOLO_2x_Config
from ppyolo_hip import synth
g = np.load(os.path.join(ROOT, 'tests/golden/g16_train_r50vd_320_fa3.npz'))
S, N, wseed, iseed, fa = [int(v) for v in g['meta']]
cfg = PPYOLO_2x_Config(); cfg.backbone['freeze_at'] = fa
m = build_train_model(cfg, wseed, 'cuda'); m.head.set_dropblock(is_test=True)
T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
x = synth.synth_images(N, S, seed=iseed).cuda()
losses = m(x, None, False, T(g['gt_bbox']), None, None, [T(g['target%d' % i]) for i in range(3)])
sum(losses.values()).backward()
sd = dict(m.named_parameters())
gn = [str(v) for v in g['grad_names']]
for i, k in enumerate(gn):
    gr = sd[k].grad.double()
    smp = gr.reshape(-1).cpu()
    step = max(1, smp.numel() // 64)
    mine = smp[::step][:64].numpy(); ref = g['grad_samples'][i][:len(mine)]
    cs = float(np.dot(mine, ref) / (np.linalg.norm(mine) * np.linalg.norm(ref) + 1e-30))
    print('%-58s ratio %.3f  cos(64 samples) %.3f' % (k, float(gr.norm()) / g['grad_digest'][i][2], cs))
