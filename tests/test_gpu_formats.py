"""f32s (fp16 high/low pair) tensors through the format-agnostic ops of the engine: channel
concatenation (Root.forward, pose_dla_dcn.py:159), max pooling (pose_dla_dcn.py:200) and the
plain <-> f32s converters -- each against torch on the decoded values."""
import pytest
import torch
import torch.nn.functional as F

from centernet_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _act(x, dev):
    from centernet_amd.engine import Act
    B, C, H, W = x.shape
    return Act(x.permute(0, 2, 3, 1).contiguous().to(dev), B, H, W, C)


def _run(pb):
    for op in pb.ops:
        op()
    torch.cuda.synchronize()


def _rand(shape, seed):
    return torch.from_numpy(synth.normal(shape, 1.5, seed))


def test_pack_unpack_round_trip_keeps_22_bits(dev):
    from centernet_amd.engine import PlanBuilder
    x = _rand((2, 96, 9, 11), 1)
    pb = PlanBuilder(dev, 2, 9, 11, split=True)
    p = pb.packed(_act(x, dev))
    q = pb.plain(p)
    _run(pb)
    assert p.fmt == "f32s" and q.fmt == "f32"
    back = q.t.permute(0, 3, 1, 2).cpu()
    assert torch.equal(p.to_float().permute(0, 3, 1, 2).cpu(), back)   # host decoding == device decoding
    assert float((back - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())


@pytest.mark.parametrize("chans", [(64, 64), (128, 32, 96), (64, 40)])
def test_concat_of_f32s_tensors(dev, chans):
    """whole 32-channel groups stay f32s (a group is 128 bytes in either format); a ragged
    channel count falls back to plain floats.  Values equal torch.cat of the decoded inputs."""
    from centernet_amd.engine import PlanBuilder
    B, H, W = 2, 12, 20
    xs = [_rand((B, c, H, W), 10 + i) for i, c in enumerate(chans)]
    pb = PlanBuilder(dev, B, H, W, split=True)
    packed = [pb.packed(_act(x, dev)) for x in xs]
    y = pb.concat(packed)
    _run(pb)
    whole = all(c % 32 == 0 for c in chans)
    assert y.fmt == ("f32s" if whole else "f32")
    ref = torch.cat([p.to_float() for p in packed], dim=3)
    assert torch.equal(y.to_float(), ref)
    # and the concatenated tensor is a valid conv input
    w = torch.from_numpy(synth.normal((32, sum(chans), 1, 1), 0.1, 3))
    z = pb.plain(pb.conv(y, w))
    for op in pb.ops[-2:]:
        op()
    torch.cuda.synchronize()
    zr = F.conv2d(torch.cat(xs, 1), w)
    assert float((z.t[..., :32].permute(0, 3, 1, 2).cpu() - zr).abs().max()) <= 2e-5 * (1 + float(zr.abs().max()))


@pytest.mark.parametrize("cfg", [(2, 64, 17, 19, 2, 2, 0), (1, 128, 16, 16, 3, 2, 1), (2, 32, 9, 9, 3, 1, 1)])
def test_maxpool_reads_f32s(dev, cfg):
    from centernet_amd.engine import PlanBuilder
    B, C, H, W, k, s, p = cfg
    x = _rand((B, C, H, W), 5)
    pb = PlanBuilder(dev, B, H, W, split=True)
    xs = pb.packed(_act(x, dev))
    n_before = len(pb.ops)
    y = pb.maxpool(xs, k, s, p)
    assert len(pb.ops) == n_before + 1 and y.fmt == "f32"      # no converter launch in between
    _run(pb)
    ref = F.max_pool2d(xs.to_float().permute(0, 3, 1, 2), k, s, p)
    assert torch.equal(y.t.permute(0, 3, 1, 2), ref)
