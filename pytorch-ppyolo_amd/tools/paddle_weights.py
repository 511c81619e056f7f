"""PaddleDetection PP-YOLO checkpoints -> this package's modules (the optional `.pdparams`
mapping of the reference's converter scripts, 1_ppyolo_2x_2pytorch.py:63-317 and 1_ppyolo_r18vd_2pytorch.py).

The reference runs Paddle to read the checkpoint and then copies array by array; which variable goes where follows from the
`name=` every Conv2dUnit was constructed with (the modules here keep the reference's constructor arguments), so the mapping is
derived from the model instead of being spelled out per layer -- and is pinned against the reference's own scripts (golden
g17: the scripts executed on a recording dict, tests/test_paddle_names.py).  Paddle itself is not needed: a `.pdparams` file
of Paddle 1.8 is a pickled {variable name: ndarray}; any other container with that mapping (np.load of an .npz) works too."""
import pickle

import numpy as np
import torch


def paddle_name_map(model):
    """{state_dict key: PaddleDetection variable name} for every parameter and BatchNorm statistic of `model`."""
    from model.custom_layers import Conv2dUnit
    out = {}
    for prefix, m in model.named_modules():
        if not isinstance(m, Conv2dUnit):
            continue
        n = m.name
        if prefix.startswith('backbone.'):
            # resnet_vd naming: conv1_1 / res2a_branch2a ... with BatchNorm 'bn' + the name minus its first three letters
            bn = 'bn' + n[3:]
            if m.use_dcn:
                out[prefix + '.conv.conv_offset.weight'] = n + '_conv_offset.w_0'
                out[prefix + '.conv.conv_offset.bias'] = n + '_conv_offset.b_0'
                out[prefix + '.conv.dcn_weight'] = n + '_weights'
            else:
                out[prefix + '.conv.weight'] = n + '_weights'
            suffix = {'weight': '_scale', 'bias': '_offset', 'running_mean': '_mean', 'running_var': '_variance'}
        else:
            # head naming: yolo_block.i.j.k / yolo_transition.i with '.conv.weights' + '.bn.*'; the output convolutions are
            # constructed as 'yolo_output.i.conv' and carry '.weights' / '.bias' directly
            if m.bn is None:
                out[prefix + '.conv.weight'] = n + '.weights'
                if m.conv.bias is not None:
                    out[prefix + '.conv.bias'] = n + '.bias'
            else:
                out[prefix + '.conv.weight'] = n + '.conv.weights'
            bn = n
            suffix = {'weight': '.bn.scale', 'bias': '.bn.offset', 'running_mean': '.bn.mean', 'running_var': '.bn.var'}
        if m.bn is not None:
            for k, sfx in suffix.items():
                out[prefix + '.bn.' + k] = bn + sfx
    return out


def read_pdparams(path):
    """{name: ndarray} from a Paddle 1.8 `.pdparams` (a pickle) or an `.npz`."""
    if str(path).endswith('.npz'):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    with open(path, 'rb') as fh:
        obj = pickle.load(fh, encoding='latin1')
    return {k: np.asarray(v) for k, v in obj.items() if isinstance(v, np.ndarray)}


def load_paddle_state(model, arrays, strict=True):
    """Copy a PaddleDetection PP-YOLO checkpoint ({variable name: ndarray}) into `model` (parameters and BatchNorm running
    statistics; convolution weights are [K, C, R, S] in both frameworks).  Returns the list of state_dict keys filled."""
    names = paddle_name_map(model)
    sd = model.state_dict()
    missing = [k for k in sd if not k.endswith('num_batches_tracked') and k not in names]
    if missing:
        raise KeyError('no Paddle name for %s' % missing[:4])
    done = []
    with torch.no_grad():
        for k, pname in names.items():
            if pname not in arrays:
                if strict:
                    raise KeyError('%s: variable %r is not in the checkpoint' % (k, pname))
                continue
            a = np.asarray(arrays[pname])
            if tuple(a.shape) != tuple(sd[k].shape):
                raise ValueError('%s <- %s: shape %s, expected %s' % (k, pname, tuple(a.shape), tuple(sd[k].shape)))
            sd[k].copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)))
            done.append(k)
    return done
