#!/usr/bin/env python
"""Process-start cost of the weights, before / after the native blob (ppyolo_hip/blob.py).  Each variant
runs in a fresh process:  load the .pt checkpoint -> (load the blob) -> first forward (plan + weight preparation + graph
capture) -> second forward.   usage: startup_time.py [r50|r18] [S] [N]     (writes /tmp/ppy_startup.*; prints JSON)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-ppyolo_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)


def child(which, S, N, mode):
    import torch
    from conftest import build_model
    from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config
    from ppyolo_hip import synth
    cfg = {'r50': PPYOLO_2x_Config, 'r18': PPYOLO_r18vd_Config}[which]()
    pt, bp = '/tmp/ppy_startup_%s.pt' % which, '/tmp/ppy_startup_%s.ppyblob' % which
    if mode == 'prepare':
        model, sd = build_model(cfg, 0, 'cuda')
        torch.save(model.state_dict(), pt)
        size = model.save_native_blob(bp)
        print(json.dumps(dict(pt_bytes=os.path.getsize(pt), blob_bytes=size)))
        return
    torch.zeros(1, device='cuda')                      # HIP context up before the clock starts
    x, ims = synth.synth_images(N, S).cuda(), synth.synth_im_size(N).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from config import select_backbone, select_head
    from model.ppyolo import PPYOLO
    if mode == 'blob_only':          # no checkpoint at all: modules on the meta device, every weight from the blob
        with torch.device('meta'):
            m = PPYOLO(select_backbone(cfg.backbone_type)(**cfg.backbone),
                       select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head))
        m.eval()
        m.head.set_dropblock(is_test=True)
        t1 = time.perf_counter()
        m.attach_native_blob(bp)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out = m(x, ims)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        m(x, ims)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(json.dumps(dict(mode=mode, construct_s=round(t1 - t0, 3), blob_s=round(t2 - t1, 3), first_forward_s=round(t3 - t2, 3),
                              second_forward_s=round(t4 - t3, 4), total_to_first_result_s=round(t3 - t0, 3),
                              detections=[int(o.shape[0]) for o in out])))
        return
    m = PPYOLO(select_backbone(cfg.backbone_type)(**cfg.backbone),
               select_head(cfg.head_type)(yolo_loss=None, nms_cfg=cfg.nms_cfg, **cfg.head))
    m.load_state_dict(torch.load(pt))
    m.eval()
    m.head.set_dropblock(is_test=True)
    m = m.cuda()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if mode == 'blob':
        m.load_native_blob(bp)
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    m(x, ims)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    m(x, ims)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(json.dumps(dict(mode=mode, checkpoint_s=round(t1 - t0, 3), blob_s=round(t2 - t1, 3), first_forward_s=round(t3 - t2, 3),
                          second_forward_s=round(t4 - t3, 4), total_to_first_result_s=round(t3 - t0, 3),
                          hbm_allocated_gb=round(torch.cuda.memory_allocated() / 1e9, 2))))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        which, S, N = (sys.argv[1:] + ['r50', '608', '8'])[:3]
        for mode in ('prepare', 'plain', 'blob', 'blob_only', 'plain', 'blob', 'blob_only'):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', which, S, N, mode], stdout=subprocess.PIPE,
                               universal_newlines=True)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED rc %d' % r.returncode)
