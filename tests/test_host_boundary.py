"""Host-side contract of the drop-in module (CPU): the checkpoint forms the reference writes load, and every way of
replacing parameters through the nn.Module API invalidates the packed engine.

Reference behaviour: checkpoints are saved from the nn.DataParallel wrapper (scripts/test_rpn_3d.py:50-54) with a leading
'module.' on every key; lib/core.py:489-499 (load_weights(remove_module=True)) strips it."""
import collections
import copy

import pytest
import torch
from torch import nn

from m3dssd_amd import synth
from model.M3d_inference_align import build

CROP = (128, 320)


@pytest.fixture(scope="module")
def net_and_sd():
    conf = synth.synth_conf(CROP, 0, batch_size=1, device="cpu")
    sd = synth.synth_state_dict(0)
    return build(conf, "test"), sd


def test_module_prefixed_checkpoint_loads_into_the_bare_module(net_and_sd):
    net, sd = net_and_sd
    wrapped = collections.OrderedDict(("module." + k, v) for k, v in sd.items())
    res = net.load_state_dict(wrapped, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    got = net.state_dict()
    assert len(got) == len(sd) == 542
    for k in ("base.base.level2.tree1.conv1.weight", "cls.6.bias", "shape_align.align.weight"):
        assert torch.equal(got[k], sd[k])
    # a checkpoint taken from a DataParallel wrapper carries _metadata with 'module.'-prefixed module names too
    dp_sd = nn.DataParallel(net).state_dict()
    assert all(k.startswith("module.") for k in dp_sd)
    net2 = build(synth.synth_conf(CROP, 0, batch_size=1, device="cpu"), "test")
    net2.load_state_dict(dp_sd, strict=True)
    assert torch.equal(net2.state_dict()["bbox_z3d_gl.1.running_var"], sd["bbox_z3d_gl.1.running_var"])
    # a mixed dict is not a wrapper checkpoint: it must fail like the reference's nn.Module would
    mixed = collections.OrderedDict(sd)
    mixed["module.cls.6.bias"] = mixed.pop("cls.6.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(mixed, strict=True)


class _Sentinel:
    pass


def test_every_nn_module_load_path_invalidates_the_packed_engine(net_and_sd):
    net, sd = net_and_sd
    net.load_state_dict(sd)
    net._engine = _Sentinel()                      # stands for a packed engine (no GPU here)
    net.load_state_dict(sd)
    assert net._engine is None
    # through a wrapper: nn.Module.load_state_dict recurses via child._load_from_state_dict, never child.load_state_dict
    net._engine = _Sentinel()
    nn.DataParallel(net).load_state_dict(collections.OrderedDict(("module." + k, v) for k, v in sd.items()))
    assert net._engine is None
    # through a container
    net._engine = _Sentinel()
    holder = nn.ModuleDict({"det": net})
    holder.load_state_dict(collections.OrderedDict(("det." + k, v) for k, v in sd.items()))
    assert net._engine is None
    # _apply (.float() / .to())
    net._engine = _Sentinel()
    net.float()
    assert net._engine is None
    net._engine = _Sentinel()
    assert net.refresh_engine() is net and net._engine is None


def test_data_parallel_replica_does_not_share_the_engine(net_and_sd):
    net, _ = net_and_sd
    net._engine = _Sentinel()
    rep = net._replicate_for_data_parallel()
    assert rep._engine is None and net._engine is not None
    assert rep._src is net and "_src" not in rep._modules          # the source is referenced, not registered as a child
    assert rep._replicate_for_data_parallel()._src is net          # a replica of a replica still points at the source
    net._engine = None
    # what torch.nn.parallel.replicate makes of the module: no parameters, children replaced by replicas -- asking such a
    # replica for its engine must reach the source's per-device cache (here: a CPU "device", refused with the CPU message)
    rep._parameters, rep._former_parameters = {}, {"w": next(net.parameters()).detach()}
    import pytest
    with pytest.raises(NotImplementedError):
        rep.engine()
    with pytest.raises(NotImplementedError):
        rep.engine("cpu")
    copy.deepcopy(net)                             # the post-hook must not break copying / pickling
    import pickle
    pickle.dumps(net._load_state_dict_post_hooks)


def test_lib_rpn_util_keeps_the_reference_name_for_the_test_driver():
    import lib.rpn_util as r
    assert callable(r.test_kitti_3d) and r.test_kitti_3d.__module__ == "m3dssd_amd.host.kitti_test"


def test_bf16_packers_refuse_weights_outside_the_fp16_range():
    """The round-5 bf16 kernels compute in fp16 inside a launch: a folded weight beyond +-65504 has no faithful copy and must not be
    packed silently (INTEGRATION.md "Limits worth knowing")."""
    import pytest
    import torch
    from m3dssd_amd import engine_bf16 as E
    w1, wp = torch.randn(64, 32, 3, 3), torch.randn(64, 32, 1, 1)
    ok = E.pack_tree_entry(w1, torch.ones(64), wp, torch.ones(64), "cpu")
    assert ok.dtype == torch.float16 and tuple(ok.shape) == (2, 1, 10, 2, 64, 8)
    with pytest.raises(RuntimeError, match="fp16 range"):
        E.pack_tree_entry(w1, torch.full((64,), 1e6), wp, torch.ones(64), "cpu")
    with pytest.raises(RuntimeError, match="fp16 range"):
        E._head2_frag(torch.full((256, 256), 7e4), torch.float16)
    assert E._head2_frag(torch.full((256, 256), 7e4), E.BF16).dtype == E.BF16          # bf16 has fp32's range
