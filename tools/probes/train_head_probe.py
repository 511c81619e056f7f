import sys, os
ROOT='/root/repo'
for p in (ROOT, ROOT+'/pytorch-ppyolo_amd', ROOT+'/tests'):
    sys.path.insert(0,p)
import numpy as np
from config import PPYOLO_2x_Config
import train_parity_util as tp
res = tp.three_way(PPYOLO_2x_Config, 608, 8, 5, True)
ho = res['head_only']
names = list(ho['grads'].keys())
print(os.environ.get('PPYOLO_HIP_TRAIN_MATH'), 'PPY_WGRAD_FP32', os.environ.get('PPY_WGRAD_FP32'))
for k in names:
    print('%-55s hip %.2e ref %.2e' % (k, ho['grads'][k][0], ho['grads'][k][1]))
hh = np.array([v[0] for v in ho['grads'].values()]); hr = np.array([v[1] for v in ho['grads'].values()])
print('median', np.median(hh), np.median(hr), 'max', hh.max(), hr.max())
