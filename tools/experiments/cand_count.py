import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/pytorch-ppyolo_amd')
import torch, bench
from ppyolo_hip import synth
for wl in ['r50vd_608', 'r18vd_416']:
    w = bench.WORKLOADS[wl]
    model, sd, cfg = bench.build_model(w['cfg'], torch.device('cuda'))
    x = synth.synth_images(8, w['size'], seed=1234).cuda(); ims = synth.synth_im_size(8).cuda()
    ex = model._plans.executor(x); ex.set_inputs(x, ims); ex.use_graph = False; ex.run(); torch.cuda.synchronize()
    print(wl, 'cand_count', ex.cand_count.tolist(), 'out_count', ex.out_count.tolist())
