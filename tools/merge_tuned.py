#!/usr/bin/env python
"""Merge autotune outputs (tools/retune.sh) into the committed per-mode table.
usage: merge_tuned.py <mode> [--only PREFIX | --match A,B] file.json [file.json ...]
(--only dcnf: just the fused-DCNv2 keys; --match ":C64:,:R1:": keys that contain all of the pieces)"""
import json
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1]
dst = os.path.join(ROOT, 'pytorch-ppyolo_amd', 'ppyolo_hip', 'tuned_gfx950%s.json' % ('' if mode == 'fp32' else '_' + mode))
tab = json.load(open(dst)) if os.path.exists(dst) else {}
files = sys.argv[2:]
only, match = None, []
if files and files[0] == '--only':
    only, files = files[1], files[2:]
if files and files[0] == '--match':
    match, files = files[1].split(','), files[2:]
for f in files:
    tab.update({k: v for k, v in json.load(open(f)).items()
                if (only is None or k.startswith(only + ':')) and all(m in k for m in match)})
json.dump(tab, open(dst, 'w'), indent=0, sort_keys=True)
print(dst, len(tab), 'entries; configs used:', Counter(v[0] for v in tab.values()).most_common())
