"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/catgan.h declares, argument
validation fails loudly with a message, and the host-side module layer reproduces the reference's graph structure
(parameter counts, getParameters() ordering and aliasing, weight-init scoping) bit-for-bit against the oracle."""
import ctypes
import importlib
import os

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cg():
    return importlib.import_module("cat-generator_amd")


def test_library_exports_every_declared_symbol(cg):
    abi = importlib.import_module("cat-generator_amd._abi")
    protos = abi.parse_header()
    assert len(protos) >= 60
    dll = ctypes.CDLL(abi.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/catgan.h but not exported"
    L = cg.lib()
    assert L.abi_version() == 1
    # nothing but extern "C" cg_* entry points with plain C types
    for name, (ret, args) in protos.items():
        assert ret in ("int", "size_t", "const char*")
        for typ, _ in args:
            assert typ in abi._CTYPES, f"{name}: non-C-ABI type {typ}"


def test_argument_validation_fails_loudly(cg):
    L = cg.lib()
    with pytest.raises(cg.CatganError, match="null pointer"):
        L.conv2d_forward(None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, None, 0)
    with pytest.raises(cg.CatganError, match="bad geometry"):
        L.conv2d_forward(None, 16, 16, None, 16, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, None, 0)
    with pytest.raises(cg.CatganError, match="out of range"):
        L.copy_channels(None, 16, 16, 4, 8, 6, 8, 0, 4)
    assert L.conv2d_workspace_bytes(128, 1, 1, 20480, 256, 1, 1, 0, 0, 0) > 0          # split-K linear head
    assert L.conv2d_workspace_bytes(128, 16, 16, 256, 128, 5, 5, 2, 2, 1) == 0           # big conv: no split


def test_missing_library_has_no_fallback(cg, monkeypatch):
    abi = importlib.import_module("cat-generator_amd._abi")
    monkeypatch.setattr(abi, "LIB_PATH", "/nonexistent/libcatgan_hip.so")
    with pytest.raises(abi.CatganError, match="no CPU fallback"):
        abi.Lib()


def test_models_match_oracle_structure_bit_for_bit(cg):
    cg.manual_seed(1); rng = O.RNG(1)
    G = cg.models.create_G((3, 32, 32), 100); D = cg.models.create_D((3, 32, 32))
    Go = O.create_G32up_c(3, 100, rng); Do = O.create_D32_st3(3, 32, rng)
    assert cg.nn_utils.getNumberOfParameters(G) == 5191687
    assert cg.nn_utils.getNumberOfParameters(D) == 6664777
    pG, gG = G.getParameters(); pD, gD = D.getParameters()
    a, _ = O.get_parameters(Go); b, _ = O.get_parameters(Do)
    np.testing.assert_array_equal(pG.numpy(), a)
    np.testing.assert_array_equal(pD.numpy(), b)
    assert cg.tensor.rng().offset == rng.offset  # both consumed the same number of draws
    # getParameters(): module tensors alias the flat vector (train.lua:184-185)
    conv = G.modules[4]
    assert conv.typename == "cudnn.SpatialConvolution" and conv.weight.t.data_ptr() >= pG.t.data_ptr()
    pG.t[conv.weight.t.data_ptr() // 4 - pG.t.data_ptr() // 4] = 42.0
    assert conv.weight.numpy().reshape(-1)[0] == 42.0
    # weight-init scoping (weight-init.lua:52): G conv bias zeroed; D Concat children keep default bias
    assert np.all(conv.bias.numpy() == 0)
    assert np.any(D.modules[7].modules[0].modules[1].bias.numpy() != 0)
    # spatial transformer initialised to the identity (models.lua:859-860)
    cls = D.modules[0].modules[0].modules[1].modules[0].modules[-1]
    assert np.all(cls.weight.numpy() == 0) and np.all(cls.bias.numpy() == [0.0])


def test_g32up_and_64px_variants(cg):
    cg.manual_seed(2)
    G1 = cg.models.create_G_decoder_upsampling32((1, 32, 32), 100)
    assert cg.nn_utils.getNumberOfParameters(G1) == 2468100
    G64 = cg.models.create_G((3, 64, 64), 100)
    D64 = cg.models.create_D((3, 64, 64))
    assert cg.nn_utils.getNumberOfParameters(D64) == 22737481        # SURVEY.md Appendix A.3 (S=64)
    assert G64.modules[0].weight.size(1) == 512 * 8 * 8


def test_transpose_is_a_relabeling_of_nhwc_storage(cg):
    x = np.arange(2 * 3 * 4 * 5, dtype=np.float32).reshape(2, 3, 4, 5)
    t = cg.Tensor.from_numpy(x)
    tr = cg.nn.Transpose((3, 4), (2, 4))
    y = tr.updateOutput(t)
    assert y.shape == (2, 4, 5, 3) and y.fmt == "plain" and y.t.data_ptr() == t.t.data_ptr()
    np.testing.assert_array_equal(y.numpy(), x.transpose(0, 2, 3, 1))
    back = cg.nn.Transpose((2, 4), (3, 4)).updateOutput(y)
    assert back.fmt == "nhwc" and back.shape == (2, 3, 4, 5)
    np.testing.assert_array_equal(back.numpy(), x)
    up = cg.nn.SpatialUpSamplingNearest(2).updateOutput(t)
    assert up.ups == 1 and up.shape == (2, 3, 8, 10) and up.t.data_ptr() == t.t.data_ptr()  # never materialised
    np.testing.assert_array_equal(up.numpy(), np.repeat(np.repeat(x, 2, 2), 2, 3))


def test_host_rng_is_the_oracle_stream(cg):
    r = cg.tensor.SplitMix(77); ro = O.RNG(77)
    np.testing.assert_array_equal(r.uniform((1000,), -0.3, 0.7), ro.uniform((1000,), -0.3, 0.7))
    np.testing.assert_array_equal(r.u01(17), O.u01(17, 77, 1000))




def test_generated_tables_are_fresh(tmp_path):
    """cat-generator_amd/csrc/net_ktable.inc and tools/abi_dispatch.inc are GENERATED from include/catgan.h by build.py (they are
    not tracked): what the built library / tools/abi_replay were compiled from must be what the generators produce from the header
    as it stands, or a stale table would shadow an entry point that changed."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, rel in (("gen_net_ktable.py", "cat-generator_amd/csrc/net_ktable.inc"), ("gen_abi_dispatch.py", "tools/abi_dispatch.inc")):
        have = os.path.join(root, rel)
        assert os.path.exists(have), f"{rel} missing: run python __graft_entry__.py (build.py generates it)"
        spec = importlib.util.spec_from_file_location(script[:-3], os.path.join(root, "scripts", script))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
        out = tmp_path / os.path.basename(rel)
        gen.main(str(out))
        assert open(have).read() == out.read_text(), f"{rel} is stale against include/catgan.h: rebuild"


def test_workspace_query_covers_the_position_major_plan_and_its_fallback(cg):
    """cg_conv2d_workspace_bytes is what a host sizes its buffer with BEFORE the launch decides between position-major tiles (round 5,
    CG_PAD_SKIP: K units of the valid taps, partial sums per unit) and the image-major plan it falls back to for unaligned operands or an
    epilogue with statistics: the query returns the larger of the two.  Host code only - no device needed."""
    L = cg.lib()
    d77 = (128, 8, 8, 128, 128, 7, 7, 3, 3, 0)        # D32_st3's last convolution (models.lua:685) at batch 128: 38 % padding
    d55 = (128, 16, 16, 64, 128, 5, 5, 2, 2, 0)       # models.lua:680: 14 % padding, below the default threshold of 20 %
    try:
        L.set_option(b"CG_PAD_SKIP", 0)
        image_major = L.conv2d_workspace_bytes(*d77)
        assert image_major == 2 * 128 * 64 * 128 * 4                          # two K splits of [8192 x 128] partial sums
        L.set_option(b"CG_PAD_SKIP", 20)
        assert L.conv2d_workspace_bytes(*d77) == 7 * 128 * 64 * 128 * 4       # an interior tile's 49 taps in seven units of 28 K tiles
        assert L.conv2d_workspace_bytes(*d77) >= image_major
        assert L.conv2d_workspace_bytes(*d55) == 0                            # stays image-major, unsplit: direct output
        # round 6, the weight gradient (igemm_tng_kernel mode 2: position-major K tiles from HALF the share, any split count): 49 row tiles x
        # 15 splits fill three workgroups per CU, 13 x 59 for the 5x5 layer; the image-major plans (16 / 64 power-of-two splits) are the
        # fallback for unaligned operands, and the query covers both
        plane = lambda taps, cin, cout: (taps * cin + 1) * cout * 4
        assert L.conv2d_wgrad_workspace_bytes(*d77) == 16 * plane(49, 128, 128)
        assert L.conv2d_wgrad_workspace_bytes(*d55) == 64 * plane(25, 64, 128)
        L.set_option(b"CG_TN_SPLITS", 40)                                     # forced: position-major takes any count, image-major 8192 / 40 -> 39
        assert L.conv2d_wgrad_workspace_bytes(*d77) == 40 * plane(49, 128, 128)
        L.set_option(b"CG_TN_SPLITS", -1)
        L.set_option(b"CG_PAD_SKIP", 0)
        assert L.conv2d_wgrad_workspace_bytes(*d77) == 16 * plane(49, 128, 128)
    finally:
        L.set_option(b"CG_PAD_SKIP", -1)
        L.set_option(b"CG_TN_SPLITS", -1)


def test_bench_work_model_matches_the_survey_and_the_dispatch():
    """bench.py::step_work is the numerator of every utilisation figure in the bench line.  Direct count: SURVEY.md 8d (F_G = 2592.4,
    F_D = 374.0 MFLOP per image, W = 3.5 F_G + 5 F_D = 10.94 GFLOP per image at configs[1]).  Executed count: what the dispatch issues -
    phase folding and the Winograd forms in G, and in D the padding taps the position-major kernels leave out (forward / data gradient:
    the 7x7 layer; weight gradient: every plain layer with >= 10 % padding) and the fused-Winograd layers' 20 / 36."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    w2 = B.step_work(B.CONFIGS[2], 128)
    assert abs(w2["F_G"] / 1e6 - 2592.4) < 0.1 and abs(w2["F_D"] / 1e6 - 374.0) < 0.1
    assert abs(w2["W"] / 1e9 - 10.943) < 0.001
    assert abs(w2["W_executed"] / 1e9 - 3.507) < 0.001
    conv = lambda ci, co, k, ho: 2.0 * ci * co * k * k * ho * ho
    share = lambda k, w: 1.0 - (sum(min(w - 1, x + k // 2) - max(0, x - k // 2) + 1 for x in range(w)) / (w * k)) ** 2
    assert abs(share(7, 8) - 0.383) < 1e-3 and abs(share(5, 16) - 0.144) < 1e-3 and abs(share(3, 8) - 0.160) < 1e-3 and share(3, 16) < 0.10
    fwd_skip = share(7, 8) * conv(128, 128, 7, 8) + 20 / 36 * (conv(64, 64, 3, 32) + 3 * conv(64, 64, 3, 16))
    wg_skip = share(7, 8) * conv(128, 128, 7, 8) + share(5, 16) * conv(64, 128, 5, 16) + share(3, 8) * 3 * conv(64, 64, 3, 8)
    assert abs(w2["D_skipped_per_pass"] - fwd_skip) < 1.0 and abs(w2["D_skipped_weight_gradient"] - wg_skip) < 1.0
    for num, N in ((2, 128), (3, 256), (5, 64), (2, 16)):
        w = B.step_work(B.CONFIGS[num], N)
        assert 0 < w["W_executed"] < w["W"] and w["D_skipped_per_pass"] < w["F_D"] and w["D_skipped_weight_gradient"] < w["F_D"]
    assert B.step_work(B.CONFIGS[2], 8)["D_skipped_weight_gradient"] == 0          # batch 8: no 16-image K tiles, nothing is skipped
