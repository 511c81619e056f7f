"""N>1 path on CPU: world_size-2 gloo processes shard a batch, all-gather packed detection
records, and every rank reconstructs the global result list."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ppyolo_hip.dist import DetectionGatherer, shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_dets(global_idx, keep):
    """Deterministic per-image record: image g has (g % 4) detections."""
    k = global_idx % 4
    d = torch.full((keep, 6), -1.0)
    for r in range(k):
        d[r] = torch.tensor([float(global_idx), 0.9 - 0.1 * r, 1., 2., 3. + r, 4. + global_idx])
    return d, k


def _worker(rank, world, port, n_global, keep, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_bounds(n_global, rank, world)
    dets = torch.stack([_fake_dets(g, keep)[0] for g in range(lo, hi)])
    cnt = torch.tensor([_fake_dets(g, keep)[1] for g in range(lo, hi)], dtype=torch.int32)
    gat = DetectionGatherer(hi - lo, keep, 'cpu')
    gat.gather(dets, cnt)
    out = gat.unpack()
    ok = len(out) == n_global
    for g, o in enumerate(out):
        d, k = _fake_dets(g, keep)
        ok = ok and tuple(o.shape) == (max(k, 1), 6) and torch.equal(o, d[:max(k, 1)])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_all_gather_of_detections_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
