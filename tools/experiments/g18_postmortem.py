#!/usr/bin/env python
"""Build container: post-mortem of the HIP box errors at R50vd-608 batch 8 from gpurun_out/r04_g18_dump.npz (tools/g18_dump.py on the GPU box)
against /root/reference -> profiles/r04_g18_postmortem.txt: HIP decode vs the reference decode on identical logits, and the worst boxes with
the reference's own re-evaluations of the same box."""
import sys
sys.argv=['x']
import importlib.util, torch, numpy as np
spec=importlib.util.spec_from_file_location('mg','/root/repo/tools/make_goldens.py'); mg=importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
from model.head import get_iou_aware_score as gi, yolo_box as yb
d=np.load('/root/repo/gpurun_out/r04_g18_dump.npz'); g=np.load('/root/repo/tests/golden/g18_r50vd_608.npz')
cfg=mg.PPYOLO_2x_Config(); m,_=mg.build_ref(cfg,0)
x=mg.synth.synth_images(8,608)
torch.set_num_threads(8)
with torch.no_grad():
    feats=m.backbone(x); outs=m.head._get_outputs(feats)
IMG=7
for math in ('f16x2','fp32'):
    print('=====',math)
    hh=[torch.from_numpy(d['%s_head%d_img%d'%(math,lv,IMG)]) for lv in range(3)]
    for lv in range(3):
        e=(hh[lv]-outs[lv][IMG]).abs()
        print('head',lv,'max |hip-ref|',float(e.max()),'rms',float(e.pow(2).mean().sqrt()))
    for k in ('a','b'):
        ims=torch.from_numpy(g['im_size_'+k])
        # reference decode applied to HIP head outputs (image IMG only)
        bs=[]; bs_ref=[]
        with torch.no_grad():
            for i,(o,oh) in enumerate(zip(outs,hh)):
                for src,dst in ((o[IMG:IMG+1],bs_ref),(oh[None],bs)):
                    t=gi(src,3,80,m.head.iou_aware_factor)
                    b,_s=yb(t,m.head._anchors[m.head.anchor_masks[i]],m.head.downsample[i],80,m.head.scale_x_y,ims[IMG:IMG+1],True,0.01)
                    dst.append(b)
        refdec_on_hip=torch.cat(bs,1)[0]; refdec=torch.cat(bs_ref,1)[0]
        hipboxes=torch.from_numpy(d['%s_%s_boxes%d'%(math,k,IMG)])
        # decode-only difference: hip decode vs reference decode, both on hip logits (boxes with conf<thr are zeroed by reference)
        nz=(refdec_on_hip.abs().sum(1)>0)
        dd=(hipboxes[nz]-refdec_on_hip[nz]).abs()
        print(k,'decode-only |hipdecode(hiplogits) - refdecode(hiplogits)| max px %.3e over %d boxes'%(float(dd.max()),int(nz.sum())))
        # rows
        cnt=int(d['%s_%s_cnt'%(math,k)][IMG]); rows=torch.from_numpy(d['%s_%s_dets'%(math,k)][IMG,:cnt]); keep=d['%s_%s_keep'%(math,k)][IMG,:cnt]
        ref=torch.from_numpy(g['t8_%s_pred%d'%(k,IMG)]); rkeep=g['t8_%s_keep%d'%(k,IMG)]
        pos={int(q):j for j,q in enumerate(rkeep)}
        worst=[]
        for j,q in enumerate(keep):
            if int(q) in pos:
                r=ref[pos[int(q)]]; e=(rows[j,2:]-r[2:]).abs()
                side=max(float(r[4]-r[2]),float(r[5]-r[3]))
                boxi=int(q)//80
                # network-noise part: refdecode(hip logits) vs refdecode(ref logits) at that box
                nn=(refdec_on_hip[boxi]-refdec[boxi]).abs().max()
                alt=[]
                for run in ('t1','native','bs1','f64'):
                    rr=torch.from_numpy(g['%s_%s_pred%d'%(run,k,IMG)]); kk=g['%s_%s_keep%d'%(run,k,IMG)]
                    pp={int(z):jj for jj,z in enumerate(kk)}
                    alt.append(float((rr[pp[int(q)],2:].float()-r[2:]).abs().max()) if int(q) in pp else -1)
                worst.append((float(e.max()),side,float(nn),alt,[float(v) for v in r[2:]],[float(v) for v in rows[j,2:]]))
        worst.sort(reverse=True)
        for w in worst[:6]:
            print('   err %.3e px side %.1f rel %.2e | via ref decode of hip logits %.3e | ref alts (t1,native,bs1,f64) %s | ref box %s'%(w[0],w[1],w[0]/max(w[1],1),w[2],['%.2e'%v for v in w[3]],['%.3f'%v for v in w[4]]))
