import sys, os, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/pytorch-ppyolo_amd')
import bench
from ppyolo_hip import synth
from ppyolo_hip.targets import gt2yolo_target, synth_ground_truth
from ppyolo_hip.train import TrainStep, lr_at
dev=torch.device('cuda:0')
model, sd, cfg = bench.build_model('PPYOLO_2x_Config', dev)
S=608; hc=cfg.head
x=synth.synth_images(8,S,seed=1234).to(dev)
bb,cc,ss=synth_ground_truth(8,50)
targets=[torch.from_numpy(t).to(dev) for t in gt2yolo_target(bb,cc,ss,hc['anchors'],hc['anchor_masks'],hc['downsample'],80,S)]
gt=torch.from_numpy(bb).to(dev)
ts=TrainStep(model,cfg,1)
for _ in range(2): ts.step(x,gt,targets,1e-4)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ts.step(x,gt,targets,1e-4)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
