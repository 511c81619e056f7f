"""The PaddleDetection variable <-> state_dict mapping (tools/paddle_weights.py) against the reference's converter scripts
(golden g17: 1_ppyolo_2x_2pytorch.py / 1_ppyolo_r18vd_2pytorch.py executed on a recording dict), and a round trip through the
loader with a synthetic checkpoint."""
import numpy as np
import pytest
import torch

from conftest import build_model
from config import PPYOLO_2x_Config, PPYOLO_r18vd_Config


@pytest.mark.parametrize('tag,cfgc', [('r50vd', PPYOLO_2x_Config), ('r18vd', PPYOLO_r18vd_Config)])
def test_name_map_matches_the_reference_scripts(golden, tag, cfgc):
    from tools.paddle_weights import paddle_name_map
    g = golden('g17_paddle_names')
    m, sd = build_model(cfgc())
    want = dict(zip((str(k) for k in g[tag + '.keys']), (str(v) for v in g[tag + '.paddle'])))
    got = paddle_name_map(m)
    assert got == want
    assert set(got) == {k for k in sd if not k.endswith('num_batches_tracked')}


def test_loader_round_trip(tmp_path):
    from tools.paddle_weights import paddle_name_map, load_paddle_state, read_pdparams
    import pickle
    cfg = PPYOLO_r18vd_Config()
    m, sd = build_model(cfg)
    names = paddle_name_map(m)
    ckpt = {p: sd[k].numpy().copy() for k, p in names.items()}
    path = tmp_path / 'ppyolo_r18vd.pdparams'
    with open(path, 'wb') as fh:
        pickle.dump(dict(ckpt, **{'StructuredToParameterName@@': {}}), fh, protocol=2)
    m2, _ = build_model(cfg, seed=7)
    assert not torch.equal(m2.state_dict()['head.yolo_output_convs.0.conv.bias'], sd['head.yolo_output_convs.0.conv.bias'])
    done = load_paddle_state(m2, read_pdparams(path))
    assert len(done) == len(names)
    for k in names:
        assert torch.equal(m2.state_dict()[k], sd[k]), k
    bad = dict(ckpt)
    first = names['backbone.stage1_conv1_1.conv.weight']
    bad[first] = bad[first][:, :2]
    with pytest.raises(ValueError):
        load_paddle_state(m2, bad)
    del bad[first]
    with pytest.raises(KeyError):
        load_paddle_state(m2, bad)
