"""oracle/net_oracle.py against goldens produced by the reference's own network classes
(msra_resnet.PoseResNet, resnet_dcn.PoseResNet) on CPU; and state-dict compatibility of
the product's parameter containers with the reference's."""
import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import net_oracle


CASES = ["res_18", "resdcn_18", "resdcn_101", "dla_34", "dla_34_pose", "hourglass"]


def _model(case, gen):
    from centernet_amd.model import create_model
    heads = gen.POSE_HEADS if case.endswith("_pose") else gen.NET_HEADS
    arch = case.replace("_pose", "")
    return create_model(arch, dict(heads), 256 if arch.startswith("dla") else 64), arch, heads


def _ref_keys(sd):
    # base.fc.* (ImageNet classifier) only exists in the reference when it loads the
    # pretrained DLA (pose_dla_dcn.py:294-305); the golden was built with pretrained=False
    return {k: v for k, v in sd.items() if not k.startswith("base.fc.")}


@pytest.mark.parametrize("case", CASES)
def test_state_dict_names_match_reference(gen, net_golden, case):
    _, meta = net_golden
    m, arch, heads = _model(case, gen)
    sd = _ref_keys(m.state_dict())
    ref = meta[case]["keys"]
    assert set(sd.keys()) == set(ref.keys())
    for k, shape in ref.items():
        assert list(sd[k].shape) == shape, k


@pytest.mark.parametrize("case", CASES)
def test_net_oracle_matches_reference(gen, net_golden, case):
    z, meta = net_golden
    m, arch, heads = _model(case, gen)
    synth.fill_state_dict_(m, gen.NET_SEED)
    sd = _ref_keys(m.state_dict())
    assert gen.sha(*[sd[k].numpy() for k in sorted(sd) if not k.endswith("num_batches_tracked")]) \
        == meta[case]["weights_sha"], "synthetic weights differ from the ones the golden was made with"
    B, H, W = gen.NET_INPUT
    x = synth.images(B, H, W, seed=0)
    assert gen.sha(x.numpy()) == meta["input_sha"]
    out = net_oracle.forward(arch, sd, x, list(heads))
    for h in heads:
        ref = z["%s/%s" % (case, h)]
        got = out[h].numpy()
        assert got.shape == ref.shape
        # same torch build + same op sequence -> identical; allow a few ulp for threading
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6), (h, np.abs(got - ref).max())
