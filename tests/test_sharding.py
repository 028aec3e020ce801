"""Multi-GPU layout of the path on CPU: image sharding is a static partition, and the one
start-up collective (flat weight broadcast) leaves every rank with rank 0's weights.
World size 2 over gloo (the GPU run uses the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from centernet_amd import sharding, synth
from centernet_amd.model import create_model


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == [len(c) for c in np.array_split(np.arange(n), world)]


def test_flatten_roundtrip():
    m = create_model("resdcn_18", {"hm": 80, "wh": 2, "reg": 2}, 64)
    synth.fill_state_dict_(m, 1)
    flat, layout = sharding.flatten_state(m)
    assert flat.numel() == sum(int(np.prod(s)) for _, s in layout)
    m2 = create_model("resdcn_18", {"hm": 80, "wh": 2, "reg": 2}, 64)
    sharding.unflatten_state_(m2, flat, layout)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        if a.is_floating_point():
            assert torch.equal(a, b), k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = create_model("res_18", {"hm": 80, "wh": 2, "reg": 2}, 64)
    synth.fill_state_dict_(m, 100 + rank)          # ranks start with DIFFERENT weights
    nbytes = sharding.broadcast_weights(m, src=0)
    flat, _ = sharding.flatten_state(m)
    lo, hi = sharding.shard_range(5, rank, world)   # 5 images over 2 ranks: 3 + 2
    dets = torch.full((hi - lo, 4, 6), float(rank))
    # uneven shards: pad to the max shard for the all_gather, as a caller would
    pad = torch.zeros((3 - (hi - lo), 4, 6))
    allg = sharding.gather_detections(torch.cat([dets, pad], 0))
    q.put((rank, nbytes, float(flat.double().sum()), (lo, hi), tuple(allg.shape),
           float(allg[3:, 0, 0].max())))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = create_model("res_18", {"hm": 80, "wh": 2, "reg": 2}, 64)
    synth.fill_state_dict_(ref, 100)
    want = float(sharding.flatten_state(ref)[0].double().sum())
    assert res[0][2] == pytest.approx(want, rel=0, abs=0)
    assert res[1][2] == res[0][2], "rank 1 must hold rank 0's weights after the broadcast"
    assert res[0][1] == res[1][1] > 50e6            # ~63 MB for res_18
    assert res[0][3] == (0, 3) and res[1][3] == (3, 5)
    assert res[0][4] == (6, 4, 6) and res[0][5] == 1.0
