"""Round 6: which table entry makes test_full_size_batch_properties[PPYOLO_2x_Config-608] fail (call d1: a batch-1 run of image 3 lands
115 px from its batch-8 row)?  One process: the batch-8 rows with the committed table, then batch-1 runs of images 3 and 7 with the
batch-1 entries of tools/probes/r06_pre_kp2_entries.json put back (a) all together, (b) one at a time.
usage: python tools/experiments/r06_kp2_bisect.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd')):
    sys.path.insert(0, p)
from config import PPYOLO_2x_Config, select_backbone, select_head      # noqa: E402
from model.ppyolo import PPYOLO      # noqa: E402
from ppyolo_hip import engine, synth      # noqa: E402


def build():
    cfg = PPYOLO_2x_Config()
    bb = select_backbone(cfg.backbone_type)(**cfg.backbone)
    hd = select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head)
    m = PPYOLO(bb, hd)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict(sd)
    m.eval()
    hd.set_dropblock(is_test=True)
    return m.cuda()


def box_gap(a, b):
    if a.shape != b.shape:
        return float('inf')
    return float((a[:, 2:] - b[:, 2:]).abs().max())


def main():
    model = build()
    N, S = 8, 608
    x = synth.synth_images(N, S)
    ims = torch.tensor([[480., 640.], [375., 500.], [608., 608.], [1080., 1920.]] * 2)      # (tests/test_gpu_model.py _FULL_IMS)
    preds = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    again = [p.cpu() for p in model(x.cuda(), ims.cuda())]
    print('batch 8 repeatable:', all(torch.equal(a, b) for a, b in zip(preds, again)))
    perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4])
    pp = [p.cpu() for p in model(x[perm].cuda(), ims[perm].cuda())]
    print('permuted batch, max box gap per image:', ['%.2e' % box_gap(pp[j], preds[i]) for j, i in enumerate(perm.tolist())])
    table = engine.tuned_table()
    committed = dict(table)
    old = json.load(open(os.path.join(ROOT, 'tools', 'probes', 'r06_pre_kp2_entries.json')))
    n1 = [k for k in old if ':N1:' in k]

    def solo(tag):
        model._plans.clear()
        out = []
        for i in (3, 7):
            s = model(x[i:i + 1].cuda(), ims[i:i + 1].cuda())[0].cpu()
            out.append(box_gap(s, preds[i]))
        print('%-60s solo image 3 / 7 box gap: %.3e / %.3e' % (tag, out[0], out[1]), flush=True)
        return max(out)

    solo('committed table')
    for k in n1:
        table[k] = old[k]
    solo('all batch-1 entries put back')
    for k in n1:
        table.update(committed)
        table[k] = old[k]
        solo('only %s back to %s' % (k, old[k][:2]))
    # and the other way round: only ONE new entry on top of the old ones
    for k in n1:
        table.update(committed)
        for q in n1:
            table[q] = old[q]
        table[k] = committed[k]
        solo('old entries, only %s on %s' % (k, committed[k][:2]))
    table.update(committed)


if __name__ == '__main__':
    main()
