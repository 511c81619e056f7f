#!/usr/bin/env python
"""Merge autotune outputs (tools/retune.sh) into the committed per-mode table.
usage: merge_tuned.py <mode> file.json [file.json ...]"""
import json
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1]
dst = os.path.join(ROOT, 'pytorch-ppyolo_amd', 'ppyolo_hip', 'tuned_gfx950%s.json' % ('' if mode == 'fp32' else '_' + mode))
tab = json.load(open(dst)) if os.path.exists(dst) else {}
for f in sys.argv[2:]:
    tab.update(json.load(open(f)))
json.dump(tab, open(dst, 'w'), indent=0, sort_keys=True)
print(dst, len(tab), 'entries; configs used:', Counter(v[0] for v in tab.values()).most_common())
