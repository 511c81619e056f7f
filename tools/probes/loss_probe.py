"""debug: the loss kernel on the ORACLE's head outputs (R50vd 320 px, golden g16 inputs): d loss / d outputs per channel group"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from conftest import build_model
from config import PPYOLO_2x_Config
from oracle import train_oracle as trn, ppyolo_oracle as orc
from ppyolo_hip import synth, ops as K
g = np.load(os.path.join(ROOT, 'tests/golden/g16_train_r50vd_320_fa3.npz'))
S, N, wseed, iseed, fa = [int(v) for v in g['meta']]
cfg = PPYOLO_2x_Config()
_, sd = build_model(cfg, wseed, 'cpu')
orc.drop_block_train = lambda x, *a, **k: x
T = lambda a: torch.from_numpy(np.asarray(a))
x = synth.synth_images(N, S, seed=iseed)
targets = [T(g['target%d' % i]) for i in range(3)]
torch.set_num_threads(16)
r = trn.train_step(sd, cfg, x, T(g['gt_bbox']), targets)
hc = cfg.head
loss6 = torch.zeros(6, device='cuda')
for i, (o, d) in enumerate(zip(r['outs'], r['douts'])):
    oh = o.detach().permute(0, 2, 3, 1).contiguous().cuda()
    dout = torch.zeros_like(oh)
    anchors = [hc['anchors'][m] for m in hc['anchor_masks'][i]]
    K.yolov3_loss(K.View(oh), targets[i].float().contiguous().cuda(), T(g['gt_bbox']).float().contiguous().cuda(), anchors, 80, hc['downsample'][i],
                  cfg.yolo_loss['scale_x_y'], cfg.yolo_loss['ignore_thresh'], cfg.iou_loss['loss_weight'], True, cfg.iou_aware_loss['loss_weight'],
                  K.View(dout), loss6, accumulate=i > 0)
    mine = dout.permute(0, 3, 1, 2).cpu()
    ref = d
    A = 3
    grp = {'ioup': slice(0, A)}
    body_m, body_r = mine[:, A:].reshape(N, A, 85, *mine.shape[2:]), ref[:, A:].reshape(N, A, 85, *ref.shape[2:])
    print('level %d: whole dout norm mine/ref %.4f  max abs diff %.3e (max |ref| %.3e)' % (i, float(mine.norm() / ref.norm()), float((mine - ref).abs().max()), float(ref.abs().max())))
    print('   ioup  norm ratio %.4f' % float(mine[:, :A].norm() / ref[:, :A].norm()))
    for nm, sl in (('xy', slice(0, 2)), ('wh', slice(2, 4)), ('obj', slice(4, 5)), ('cls', slice(5, 85))):
        print('   %-4s norm ratio %.4f  max abs diff %.3e' % (nm, float(body_m[:, :, sl].norm() / body_r[:, :, sl].norm()), float((body_m[:, :, sl] - body_r[:, :, sl]).abs().max())))
print('loss6', loss6.cpu().numpy(), [float(v) for v in r['losses'].values()])
