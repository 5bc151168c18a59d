"""The drop-in boundary without its Python host: one whole D+G update (adversarial.lua:51-275) recorded as a sequence of
C-ABI calls and replayed by tools/abi_replay - a plain C++ program that owns its device memory (cg_malloc), its stream
(cg_stream_create) and dispatches every call by name through a table generated from include/catgan.h.  No interpreter,
no PyTorch in that process.

The recorded host is the module layer with every fast path off (nn.fusion = False, no grouped / stacked branches, one
stream): one C call per nn.Module method - the call sequence lua/catgan/nn.lua issues, class for class (the LuaJIT layer
cannot run in this image; scripts/check_lua_binding.py checks its C calls against the header statically).  The replay must
reproduce the Python host's parameter vectors exactly."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
f32 = np.float32


@pytest.mark.gpu
def test_whole_step_replayed_through_the_c_abi_without_python(tmp_path):
    from abi_record import Recorder
    cg = importlib.import_module("cat-generator_amd")
    exe = os.path.join(ROOT, "tools", "abi_replay")
    assert os.path.exists(exe), "tools/abi_replay is built by __graft_entry__.build()"
    nn = cg.nn
    saved = (nn.fusion, nn.Concat.grouped, nn.Concat.overlap_groups, nn._Stackable.stacking)
    nn.fusion, nn.Concat.grouped, nn.Concat.overlap_groups, nn._Stackable.stacking = False, False, False, False
    try:
        cg.manual_seed(91)
        G, D = cg.models.create_G((3, 32, 32), 100), cg.models.create_D((3, 32, 32))
        S = cg.adversarial.State(dict(batchSize=8), G, D)
        S.device_rng = True      # real-batch indices from the device generator: nothing but ABI calls touches the device
        data = cg.adversarial.TrainData(np.random.RandomState(4).rand(32, 3, 32, 32).astype(f32))
        cg.adversarial.iteration(S, data, 8)      # warm-up: every buffer exists, weights are packed
        rec = Recorder(cg)
        rec.start()
        cg.adversarial.iteration(S, data, 8)      # the recorded update
        torch.cuda.synchronize()
        pD, pG = S.PARAMETERS_D, S.PARAMETERS_G
        trace, blob = str(tmp_path / "step.trace"), str(tmp_path / "step.blob")
        ncalls = rec.stop({str(tmp_path / "pD.bin"): (pD.ptr, pD.nElement() * 4), str(tmp_path / "pG.bin"): (pG.ptr, pG.nElement() * 4)},
                          trace, blob)
        assert ncalls > 300, ncalls
        names = {l.split("|")[1] for l in open(trace) if l.startswith("call|")}
        for must in ("cg_conv2d_forward", "cg_conv2d_wgrad", "cg_conv2d_dgrad_ups2", "cg_prelu_backward", "cg_bn_forward", "cg_bn_backward",
                     "cg_bilinear_sampler_backward", "cg_adam_step", "cg_pack_conv_weight", "cg_rng_bernoulli_dev", "cg_bce_backward"):
            assert must in names, f"{must} missing from the recorded step"
        assert not any("grouped" in n or n in ("cg_conv2d_forward_ex", "cg_act_pool2_mask_forward", "cg_bn_act_forward") for n in names), \
            "the recorded host must be the plain one-call-per-module layer"
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "cat-generator_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([exe, trace, blob], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"{ncalls} calls replayed" in r.stdout, r.stdout
        rD = np.fromfile(str(tmp_path / "pD.bin"), dtype=f32); rG = np.fromfile(str(tmp_path / "pG.bin"), dtype=f32)
        np.testing.assert_array_equal(rD, pD.numpy())       # D: no atomics of any kind on the path
        dG = np.abs(rG - pG.numpy())                        # G: batch-norm column sums use fp64 atomics (order-dependent in the last bit)
        assert dG.max() <= 2.5e-3 and np.mean(dG > 0) < 1e-3, (dG.max(), np.mean(dG > 0))
    finally:
        nn.fusion, nn.Concat.grouped, nn.Concat.overlap_groups, nn._Stackable.stacking = saved


def test_lua_binding_calls_match_the_header():
    """scripts/check_lua_binding.py: every C.cg_* call in lua/ names a declared entry point with the right arity, the block
    structure of every file balances, and every class models.lua / adversarial.lua instantiate is defined."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_lua_binding.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 problems" in r.stdout


def test_replayer_dispatch_covers_every_compute_entry_point():
    """tools/abi_dispatch.inc (generated from the header) has an entry for every int-returning entry point whose arguments
    are device pointers and scalars - i.e. everything a training step can call."""
    cg = importlib.import_module("cat-generator_amd")
    protos = cg._abi.parse_header()
    inc = open(os.path.join(ROOT, "tools", "abi_dispatch.inc")).read()
    host_only = {"cg_comm_available", "cg_comm_init", "cg_comm_size", "cg_device_count", "cg_get_option", "cg_malloc", "cg_set_option",
                 "cg_stream_create", "cg_abi_version", "cg_pack_conv_weight_batch", "cg_concat_channels",
                 "cg_split_channels",   # the last three: host int arrays (fused hosts only)
                 "cg_conv2d_wgrad_pending", "cg_host_alloc", "cg_event_create"}
    for name, (ret, _) in protos.items():
        if ret == "int" and name not in host_only:
            assert f'"{name}"' in inc, f"{name} has no dispatch entry (re-run scripts/gen_abi_dispatch.py)"
