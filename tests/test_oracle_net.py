"""oracle/net_oracle.py against goldens produced by the reference's own network classes
(msra_resnet.PoseResNet, resnet_dcn.PoseResNet) on CPU; and state-dict compatibility of
the product's parameter containers with the reference's."""
import numpy as np
import pytest
import torch

from centernet_amd import synth
from oracle import net_oracle


def _model(arch, heads):
    from centernet_amd.model import create_model
    return create_model(arch, dict(heads), 64)


@pytest.mark.parametrize("arch", ["res_18", "resdcn_18"])
def test_state_dict_names_match_reference(gen, net_golden, arch):
    _, meta = net_golden
    m = _model(arch, gen.NET_HEADS)
    sd = m.state_dict()
    ref = meta[arch]["keys"]
    assert set(sd.keys()) == set(ref.keys())
    for k, shape in ref.items():
        assert list(sd[k].shape) == shape, k


@pytest.mark.parametrize("arch", ["res_18", "resdcn_18"])
def test_net_oracle_matches_reference(gen, net_golden, arch):
    z, meta = net_golden
    m = _model(arch, gen.NET_HEADS)
    synth.fill_state_dict_(m, gen.NET_SEED)
    sd = m.state_dict()
    assert gen.sha(*[sd[k].numpy() for k in sorted(sd) if not k.endswith("num_batches_tracked")]) \
        == meta[arch]["weights_sha"], "synthetic weights differ from the ones the golden was made with"
    B, H, W = gen.NET_INPUT
    x = synth.images(B, H, W, seed=0)
    assert gen.sha(x.numpy()) == meta["input_sha"]
    out = net_oracle.forward(arch, sd, x, list(gen.NET_HEADS))
    for h in gen.NET_HEADS:
        ref = z["%s/%s" % (arch, h)]
        got = out[h].numpy()
        assert got.shape == ref.shape
        # same torch build + same op sequence -> identical; allow a few ulp for threading
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6), (h, np.abs(got - ref).max())
