"""Native weight blob: the folded / re-laid / pre-split weights of a model exactly as the kernels read them, in one file.

The reference's checkpoint is a `torch.save`d `state_dict` (reference 1_ppyolo_2x_2pytorch.py:321, read at demo.py:91);
from it every process start derives, per convolution: BatchNorm folded to (scale, shift) (engine.fold_bn), the weights in
KRSC order, and -- in the default math mode -- two scaled fp16 planes in chunk-major order plus three bf16 planes
(csrc/conv_x3.hip).  None of that depends on the input: the blob stores the results, so a
later start uploads one contiguous region and builds its plans shape-only (engine.Builder(skeleton=True)).

File layout (little endian):
    8 B   magic  'PPYBLOB1'
    8 B   u64    length of the JSON header
    JSON  {'math', 'fingerprint', 'version', 'lists': {'setup_ops': [...], 'ops': [...]}}; an entry is null for an op
          without weights, else {key: {'dtype', 'shape', 'offset'}} for key in w / scale / shift / w3 / wf16_planes /
          wf16_scale; offsets are relative to the data region and 256-byte aligned
    pad to a multiple of 4096, then the data region.

A blob belongs to one set of parameters: `fingerprint(state_dict)` is stored and checked on load.
"""
import hashlib
import json
import os
import struct

import numpy as np
import torch

from ._lib import PPYoloHipError

MAGIC = b'PPYBLOB1'
VERSION = 2          # 2: the header names the plan-shaping switches the weights were folded under (round 5)
_KEYS = ('w', 'scale', 'shift', 'w3')
_DT = {'float32': torch.float32, 'int16': torch.int16}


def plan_switches():
    """The environment switches that change WHICH weight tensors a plan holds (not how they are run): a blob written under other
    settings does not fit the plans this process builds -- with the shortcut fold conv3 and conv4 of a stage's first block are one
    [f3, f2 + in_c] tensor, with the K padding the 258-filter output convolutions carry 260."""
    return dict(fold_shortcut=os.environ.get('PPYOLO_HIP_FOLD_SHORTCUT', '1') == '1', pad_k=os.environ.get('PPYOLO_HIP_PAD_K', '1') == '1')


def fingerprint(state_dict):
    """Cheap identity of a parameter set: names, shapes, dtypes and two float64 moments per tensor (one device->host
    transfer when the tensors live on the GPU)."""
    h = hashlib.sha256()
    moms = []
    for k in sorted(state_dict):
        v = state_dict[k]
        h.update(('%s|%s|%s;' % (k, tuple(v.shape), v.dtype)).encode())
        d = v.detach().double().reshape(-1)
        idx = torch.arange(1, d.numel() + 1, dtype=torch.float64, device=d.device)
        moms.append(torch.stack([d.sum(), (d * idx).sum()]) if d.numel() else torch.zeros(2, dtype=torch.float64, device=d.device))
    h.update(torch.stack(moms).cpu().numpy().tobytes())
    return h.hexdigest()


class _Plan(object):
    def __init__(self):
        self.setup_ops, self.ops = [], []


class BlobOwner(object):
    """What HipExecutor(share=...) needs of a weight owner: `.math`, `.plan.setup_ops`, `.plan.ops` (dicts with the
    device tensors w / scale / shift / w3 / wf16)."""

    def __init__(self, math, device):
        self.math, self.device, self.plan = math, torch.device(device), _Plan()
        self.nbytes = 0


def save(ex, path, fp):
    """Write the weights of executor `ex` (a fully built HipExecutor, any input shape) to `path`."""
    lists, chunks, off = {}, [], 0
    for name in ('setup_ops', 'ops'):
        ents = []
        for op in getattr(ex.plan, name):
            if op.get('w') is None:
                ents.append(None)
                continue
            rec = {}
            items = [(k, op.get(k)) for k in _KEYS]
            if op.get('wf16') is not None:
                items += [('wf16_planes', op['wf16'][0]), ('wf16_scale', op['wf16'][1])]
            for k, t in items:
                if t is None:
                    continue
                a = t.detach().contiguous().cpu().numpy()
                rec[k] = dict(dtype=str(a.dtype), shape=list(a.shape), offset=off)
                chunks.append((off, a))
                off += (a.nbytes + 255) // 256 * 256
            ents.append(rec)
        lists[name] = ents
    header = json.dumps(dict(version=VERSION, math=ex.math, fingerprint=fp, data_bytes=off, lists=lists, switches=plan_switches())).encode()
    start = (16 + len(header) + 4095) // 4096 * 4096
    tmp = path + '.tmp.%d' % os.getpid()
    with open(tmp, 'wb') as fh:
        fh.write(MAGIC + struct.pack('<Q', len(header)) + header)
        fh.write(b'\0' * (start - 16 - len(header)))
        pos = 0
        for o, a in chunks:
            fh.write(b'\0' * (o - pos))
            fh.write(a.tobytes())
            pos = o + a.nbytes
        fh.write(b'\0' * (off - pos))
    os.replace(tmp, path)
    return start + off


def read_header(path):
    with open(path, 'rb') as fh:
        head = fh.read(16)
        if len(head) < 16 or head[:8] != MAGIC:
            raise PPYoloHipError('%s is not a PPYBLOB1 weight blob' % path)
        n, = struct.unpack('<Q', head[8:])
        hdr = json.loads(fh.read(n).decode())
    if hdr.get('version') != VERSION:
        raise PPYoloHipError('%s: blob version %r, this build reads %d -- rebuild it with model.save_native_blob()' % (path, hdr.get('version'), VERSION))
    if hdr.get('switches') != plan_switches():
        raise PPYoloHipError('%s was written under other plan-shaping switches (%r) than this process runs with (%r): rebuild it with '
                             'model.save_native_blob(), or set PPYOLO_HIP_FOLD_SHORTCUT / PPYOLO_HIP_PAD_K as they were'
                             % (path, hdr.get('switches'), plan_switches()))
    return hdr, (16 + n + 4095) // 4096 * 4096


def read(path, expect_fingerprint=None):
    """-> (header, data region as a host uint8 tensor); checks magic, version, size and (optionally) the fingerprint."""
    hdr, start = read_header(path)
    if expect_fingerprint is not None and hdr['fingerprint'] != expect_fingerprint:
        raise PPYoloHipError('%s was written for other parameters than the model holds (fingerprint mismatch): rebuild it '
                             'with model.save_native_blob()' % path)
    if os.path.getsize(path) < start + hdr['data_bytes']:
        raise PPYoloHipError('%s is truncated' % path)
    return hdr, torch.from_numpy(np.fromfile(path, dtype=np.uint8, count=hdr['data_bytes'], offset=start))


def views(hdr, data, math_device):
    """BlobOwner over `data` (a uint8 tensor holding the data region): every weight tensor is a view into it."""
    own = BlobOwner(hdr['math'], math_device)
    own.nbytes = hdr['data_bytes']

    def view(r):
        dt = _DT[r['dtype']]
        n = int(np.prod(r['shape'])) * (4 if dt == torch.float32 else 2)
        return data[r['offset']:r['offset'] + n].view(dt).view(r['shape'])
    for name in ('setup_ops', 'ops'):
        out = getattr(own.plan, name)
        for rec in hdr['lists'][name]:
            if rec is None:
                out.append({})
                continue
            op = {k: view(rec[k]) for k in _KEYS if k in rec}
            if 'wf16_planes' in rec:
                op['wf16'] = (view(rec['wf16_planes']), view(rec['wf16_scale']))
            out.append(op)
    return own


def load(path, device, expect_fingerprint=None):
    """-> BlobOwner whose tensors are views into ONE device allocation filled by one host->device copy."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise PPYoloHipError('a weight blob is loaded onto a ROCm device (got %s); there is no CPU path' % device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    hdr, host = read(path, expect_fingerprint)
    dev = torch.empty(hdr['data_bytes'], dtype=torch.uint8, device=device)
    dev.copy_(host)
    own = views(hdr, dev, device)
    own.storage = dev
    return own
