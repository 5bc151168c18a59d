"""Oracle-compared parity at the BENCHMARKED sizes (BASELINE.json configs[1] batch 128, configs[2] batch 256) and
for every compiled kernel variant the dispatch can select.

tests/test_gpu_parity.py compares with the oracle at batch <= 8, where `pick_tile` / `plan_nn` / the Winograd K-step
rule select small-grid variants.  Here every convolution and linear layer of G32up-c / G32up / D32_st3
(models.lua:138-160,196-228,640-711,814-906) runs at the batch the bench runs it (N and N/2: the fake-generation
forward of adversarial.lua:232 is a half batch) through forward, data gradient and weight gradient against the CPU
oracle, and `cg_set_option` forces each block tile / split-K / K-step / Winograd variant on small oracle-checked
shapes, so that every kernel in profiles/*_bench_kernel_stats.csv also occurs inside an oracle comparison
(scripts/kernel_coverage.py lists the two sets side by side).

Tolerance (fp32, different summation orders): |d| <= 2e-5 * sqrt(K/1024) * max(1, max|ref|), K = reduction length
(4e-5 for the weight gradient, whose K = N*H*W reaches 131072).
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def cg():
    mod = importlib.import_module("cat-generator_amd")
    assert torch.cuda.is_available(), "these tests need the MI355X"
    mod.lib()
    return mod


def close(a, b, K=1024, tol=2e-5, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    s = max(1.0, np.sqrt(K / 1024.0)) * max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    assert err <= tol * s, f"{what}: max|d|={err:.3e} > {tol * s:.3e} (K={K})"
    return err


class options:
    """with options(cg, CG_NN_TILE=128064, ...): force dispatch tunables through the C ABI, restore afterwards."""

    def __init__(self, cg, **kv):
        self.cg, self.kv = cg, kv

    def __enter__(self):
        for k, v in self.kv.items():
            self.cg.lib().set_option(k.encode(), int(v))

    def __exit__(self, *exc):
        for k in self.kv:
            self.cg.lib().set_option(k.encode(), -1)


def run_conv(cg, N, Cin, H, W, Cout, k, ups, seed=0, wino=True, check_dgrad=True):
    """One SpatialConvolution (optionally behind the lazy 2x upsampling) forward / backward against the oracle."""
    rs = np.random.RandomState(seed)
    pad = (k - 1) // 2
    cg.nn.SpatialConvolution.winograd = wino
    try:
        m = cg.nn.SpatialConvolution(Cin, Cout, k, k, 1, 1, pad)
        w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(f32)
        b = rs.randn(Cout).astype(f32)
        m.weight.copy(w); m.bias.copy(b)
        x = rs.randn(N, Cin, H, W).astype(f32)
        xin = cg.Tensor.from_numpy(x)
        xl = x
        if ups:
            up = cg.nn.SpatialUpSamplingNearest(2)
            xin = up.forward(xin)
            xl = np.repeat(np.repeat(x, 2, axis=2), 2, axis=3)
        y = m.forward(xin).numpy()
        close(y, O.conv2d_forward(xl, w, b, pad), K=Cin * k * k, what="updateOutput")
        dy = rs.randn(*y.shape).astype(f32)
        m.gradWeight.zero(); m.gradBias.zero()
        gi_t = m.backward(xin, cg.Tensor.from_numpy(dy))
        if check_dgrad:
            ref = O.conv2d_backward_data(dy, w, xl.shape, pad)
            if ups:
                gi = up.updateGradInput(None, gi_t).numpy()
                close(gi, O.UpSample2().backward(ref), K=4 * Cout * k * k, what="updateGradInput (+2x2 block sum)")
            else:
                close(gi_t.numpy(), ref, K=Cout * k * k, what="updateGradInput")
        gw, gb = np.zeros_like(w), np.zeros_like(b)
        O.conv2d_backward_weight(xl, dy, gw, gb, pad)
        P = y.shape[0] * y.shape[2] * y.shape[3]
        close(m.gradWeight.numpy(), gw, K=P, tol=4e-5, what="gradWeight")
        close(m.gradBias.numpy(), gb, K=P, tol=4e-5, what="gradBias")
        return m
    finally:
        cg.nn.SpatialConvolution.winograd = True


# ------------------------------------------------------------------ every conv layer at the benchmarked batch
# (name, N, Cin, H, W, Cout, k, ups): H, W are the convolution's INPUT dims before the folded upsampling
FULL_CONVS = [
    # config #2: G32up-c, batch 128 (G-step) and 64 (fake generation for the D-step)
    ("G32up-c conv 512->512 3x3 @4->8 (models.lua:205-206)", 128, 512, 4, 4, 512, 3, 1),
    ("G32up-c conv 512->512, half batch", 64, 512, 4, 4, 512, 3, 1),
    ("G32up-c conv 512->256 3x3 @8->16 (:211-212)", 128, 512, 8, 8, 256, 3, 1),
    ("G32up-c conv 512->256, half batch", 64, 512, 8, 8, 256, 3, 1),
    ("G32up-c conv 256->128 5x5 @16->32 (:217-218), Winograd", 128, 256, 16, 16, 128, 5, 1),
    ("G32up-c conv 256->128 5x5, half batch", 64, 256, 16, 16, 128, 5, 1),
    ("G32up-c conv 128->3 3x3 @32 (:222), skinny", 128, 128, 32, 32, 3, 3, 0),
    # D32_st3, batch 128
    ("D conv 3->64 3x3 @32 (:646)", 128, 3, 32, 32, 64, 3, 0),
    ("D conv 64->64 3x3 @32 (:648)", 128, 64, 32, 32, 64, 3, 0),
    ("D branch conv 64->64 3x3 @16 (:655)", 128, 64, 16, 16, 64, 3, 0),
    ("D branch conv 64->64 3x3 @8 (:659)", 128, 64, 8, 8, 64, 3, 0),
    ("D branch-4 conv 64->128 5x5 @16 (:681)", 128, 64, 16, 16, 128, 5, 0),
    ("D branch-4 conv 128->128 7x7 @8 (:685)", 128, 128, 8, 8, 128, 7, 0),
    ("ST0 loc conv 3->16 @16 (:844)", 128, 3, 16, 16, 16, 3, 0),
    ("ST0 loc conv 16->16 @16 (:846)", 128, 16, 16, 16, 16, 3, 0),
    ("branch ST loc conv 64->16 @8 (:844), stacked x3", 384, 64, 8, 8, 16, 3, 0),
    ("branch ST loc conv 16->16 @8 (:846), stacked x3", 384, 16, 8, 8, 16, 3, 0),
    # config #3: G32up grayscale, batch 256 / 128
    ("G32up conv 128->256 5x5 @8->16 (:143-144), bs256", 256, 128, 8, 8, 256, 5, 1),
    ("G32up conv 256->128 5x5 @16->32 (:148-149), bs256", 256, 256, 16, 16, 128, 5, 1),
    ("G32up conv 128->1 3x3 @32 (:153), bs256", 256, 128, 32, 32, 1, 3, 0),
    ("D conv 1->64 3x3 @32, bs256", 256, 1, 32, 32, 64, 3, 0),
    ("D conv 64->64 3x3 @32, bs256", 256, 64, 32, 32, 64, 3, 0),
    ("D branch-4 conv 64->128 5x5 @16, bs256", 256, 64, 16, 16, 128, 5, 0),
    ("D branch-4 conv 128->128 7x7 @8, bs256", 256, 128, 8, 8, 128, 7, 0),
    # config #5's per-GPU share: G32up-c@64 / D32_st3@64 (models.lua:645,654,696-697 scale D with `dimensions`), 64 images per
    # GPU (G-step) and 32 (fake generation)
    ("c5 G conv 512->512 3x3 @8->16, bs64", 64, 512, 8, 8, 512, 3, 1),
    ("c5 G conv 512->256 3x3 @16->32, bs64", 64, 512, 16, 16, 256, 3, 1),
    ("c5 G conv 512->256 3x3 @16->32, half batch", 32, 512, 16, 16, 256, 3, 1),
    ("c5 G conv 256->128 5x5 @32->64, bs64, Winograd", 64, 256, 32, 32, 128, 5, 1),
    ("c5 G conv 256->128 5x5 @32->64, half batch", 32, 256, 32, 32, 128, 5, 1),
    ("c5 G conv 128->3 3x3 @64, skinny", 64, 128, 64, 64, 3, 3, 0),
    ("c5 D conv 3->64 3x3 @64", 64, 3, 64, 64, 64, 3, 0),
    ("c5 D conv 64->64 3x3 @64", 64, 64, 64, 64, 64, 3, 0),
    ("c5 D branch conv 64->64 3x3 @32", 64, 64, 32, 32, 64, 3, 0),
    ("c5 D branch conv 64->64 3x3 @16", 64, 64, 16, 16, 64, 3, 0),
    ("c5 D branch-4 conv 64->128 5x5 @32", 64, 64, 32, 32, 128, 5, 0),
    ("c5 D branch-4 conv 128->128 7x7 @16", 64, 128, 16, 16, 128, 7, 0),
    ("c5 ST0 loc conv 3->16 @32", 64, 3, 32, 32, 16, 3, 0),
    ("c5 branch ST loc conv 64->16 @16, stacked x3", 192, 64, 16, 16, 16, 3, 0),
]


@pytest.mark.parametrize("name,N,Cin,H,W,Cout,k,ups", FULL_CONVS, ids=[c[0] for c in FULL_CONVS])
def test_conv_layer_at_benchmarked_batch(cg, name, N, Cin, H, W, Cout, k, ups):
    m = run_conv(cg, N, Cin, H, W, Cout, k, ups, seed=len(name))
    if "Winograd" in name:
        assert getattr(m, "_wino", False), "the benchmarked dispatch runs this layer on the Winograd kernels"


# (N, Cin, H, W, Cout): 3x3 layers with <= 3 planes on ONE side.  Cout <= 3: forward + weight gradient run the skinny kernels
# (csrc/skinny.hip on the MFMA for widths % 32 == 0, gemm.hip's VALU kernels otherwise / with CG_SKINNY=2); Cin <= 3: the DATA gradient
# does (D's first layer seen from its output, models.lua:646).  Batches that are not a multiple of 8 (block -> image map), heights that
# are not a multiple of the strip, every compiled (planes, width) combination, a width the MFMA path refuses (8, 40).
SKINNY_CASES = [(3, 128, 32, 32, 3), (9, 64, 40, 32, 3), (2, 128, 64, 64, 3), (16, 128, 32, 32, 1), (2, 64, 24, 96, 1), (5, 128, 6, 128, 3),
                (2, 64, 8, 8, 1), (3, 128, 5, 40, 3), (3, 3, 32, 32, 64), (2, 1, 64, 64, 64), (24, 3, 20, 32, 64)]


@pytest.mark.parametrize("mode", [1, 2], ids=["mfma", "valu"])
@pytest.mark.parametrize("N,Cin,H,W,Cout", SKINNY_CASES)
def test_skinny_3x3_layers_on_both_kernel_families(cg, mode, N, Cin, H, W, Cout):
    with options(cg, CG_SKINNY=mode):
        run_conv(cg, N, Cin, H, W, Cout, 3, 0, seed=N + Cin + H + W + Cout)


# (N, H, W): plain 64 -> 64 plane 3x3 layers (models.lua:648,655,664,673) on the fused-transform Winograd kernel (csrc/wino3.hip; CG_WINO3 = 2
# takes it wherever the geometry fits, 0 never) and on the direct implicit GEMM: forward and data gradient (flipped filters) run the
# fused kernel, the weight gradient stays direct.  One block per image, several block rows / columns, image borders on every side of a
# block, a batch that is not a multiple of anything, the benchmarked layer itself on the DIRECT kernel (the default takes the fused one).
WINO3_CASES = [(2, 8, 16), (3, 16, 16), (2, 24, 32), (5, 32, 48), (1, 64, 64)]


@pytest.mark.parametrize("mode", [2, 0], ids=["fused", "direct"])
@pytest.mark.parametrize("N,H,W", WINO3_CASES)
def test_fused_winograd_for_the_64_plane_3x3_layers(cg, mode, N, H, W):
    with options(cg, CG_WINO3=mode):
        run_conv(cg, N, 64, H, W, 64, 3, 0, seed=N + H + W)


def test_direct_kernel_still_covers_the_benchmarked_64_plane_layer(cg):
    with options(cg, CG_WINO3=0):
        run_conv(cg, 128, 64, 32, 32, 64, 3, 0, seed=7)


FULL_LINEARS = [
    ("G32up-c Linear 100->8192 (models.lua:199)", 128, 100, 8192),
    ("G32up-c Linear 100->8192, half batch", 64, 100, 8192),
    ("G32up Linear 100->8192 (:139), bs256", 256, 100, 8192),
    ("D Linear 20480->256 (:697)", 128, 20480, 256),
    ("D Linear 20480->256, bs256", 256, 20480, 256),
    ("D Linear 256->1 (:700)", 128, 256, 1),
    ("ST0 loc Linear 1024->64 (:850)", 128, 1024, 64),
    ("ST0 loc Linear 64->1 (:853)", 128, 64, 1),
    ("branch ST loc Linear 256->64, stacked x3", 384, 256, 64),
    ("branch ST loc Linear 64->4", 128, 64, 4),
    # config #5 (64x64, 64 images per GPU)
    ("c5 G Linear 100->32768", 64, 100, 32768),
    ("c5 D Linear 81920->256", 64, 81920, 256),
    ("c5 ST0 loc Linear 4096->64", 64, 4096, 64),
    ("c5 branch ST loc Linear 1024->64, stacked x3", 192, 1024, 64),
]


@pytest.mark.parametrize("name,N,i,o", FULL_LINEARS, ids=[c[0] for c in FULL_LINEARS])
def test_linear_layer_at_benchmarked_batch(cg, name, N, i, o):
    rs = np.random.RandomState(N + i + o)
    m = cg.nn.Linear(i, o)
    w = (rs.randn(o, i) / np.sqrt(i)).astype(f32); b = rs.randn(o).astype(f32)
    m.weight.copy(w); m.bias.copy(b)
    x = rs.randn(N, i).astype(f32); dy = rs.randn(N, o).astype(f32)
    close(m.forward(cg.Tensor.from_numpy(x)).numpy(), O.linear_forward(x, w, b), K=i, what="forward")
    m.gradWeight.zero(); m.gradBias.zero()
    gi = m.backward(cg.Tensor.from_numpy(x), cg.Tensor.from_numpy(dy)).numpy()
    close(gi, O.linear_backward_data(dy, w), K=o, what="gradInput")
    gw, gb = np.zeros_like(w), np.zeros_like(b)
    O.linear_backward_weight(x, dy, gw, gb)
    close(m.gradWeight.numpy(), gw, K=N, what="gradWeight"); close(m.gradBias.numpy(), gb, K=N, what="gradBias")


def test_grouped_branch_convs_at_benchmarked_batch(cg):
    """D32_st3's three identical transformer branches (models.lua:653-678) run their convolutions as ONE grouped launch
    (blockIdx.z = branch) inside the planned pass: an nn.Concat of three structurally identical conv -> PReLU branches at batch
    128 - forward (grouped GEMM with the activation in its epilogue), data gradient and weight gradient of the group."""
    rs = np.random.RandomState(12)
    N, C, H = 128, 64, 16
    net = cg.nn.Sequential()
    cat = cg.nn.Concat(2)
    convs, ws, bs = [], [], []
    for g in range(3):
        m = cg.nn.SpatialConvolution(C, C, 3, 3, 1, 1, 1)
        w = (rs.randn(C, C, 3, 3) / 24).astype(f32); b = rs.randn(C).astype(f32)
        m.weight.copy(w); m.bias.copy(b); m.gradWeight.zero(); m.gradBias.zero()
        cat.add(cg.nn.Sequential().add(m).add(cg.nn.PReLU()))
        convs.append(m); ws.append(w); bs.append(b)
    net.add(cat)
    x = rs.randn(N, C, H, H).astype(f32); dy = rs.randn(N, 3 * C, H, H).astype(f32)
    xin = cg.nn.as_nhwc(cg.Tensor.from_numpy(x))
    out = net.forward(xin).numpy()
    assert net._planned_last
    gin = cg.nn.as_nhwc(net.backward(xin, cg.nn.as_nhwc(cg.Tensor.from_numpy(dy)))).numpy()
    gref = np.zeros_like(x)
    for g in range(3):
        y = O.conv2d_forward(x, ws[g], bs[g], 1)
        A = O.PReLU()
        close(out[:, g * C:(g + 1) * C], A.forward(y), K=C * 9, what=f"branch {g} output")
        dyg = A.backward(np.ascontiguousarray(dy[:, g * C:(g + 1) * C]))
        gref = gref + O.conv2d_backward_data(dyg, ws[g], x.shape, 1)
        gw, gb = np.zeros_like(ws[g]), np.zeros_like(bs[g])
        O.conv2d_backward_weight(x, dyg, gw, gb, 1)
        close(convs[g].gradWeight.numpy(), gw, K=N * H * H, tol=4e-5, what=f"branch {g} gradWeight")
        close(convs[g].gradBias.numpy(), gb, K=N * H * H, tol=4e-5, what=f"branch {g} gradBias")
    close(gin, gref, K=3 * C * 9, what="gradInput (sum over the branches)")


# ------------------------------------------------------------------ every compiled variant, forced
NN_TILES = [128128, 64128, 128064, 64064, 128032]


@pytest.mark.parametrize("tile", NN_TILES)
@pytest.mark.parametrize("bk32", [0, 1])
@pytest.mark.parametrize("splits", [0, 3])
@pytest.mark.parametrize("stage", ["b32", "glds"])
def test_forced_nn_tile_variants(cg, tile, bk32, splits, stage):
    """igemm_nn_kernel<BM,BN,...,BK> / igemm_nng_kernel<BM,BN,..,BK> for every block tile, K step 16 / 32, with and
    without split-K (+ reduce kernel), on plain, upsample-folded (4 phases) and folded data-gradient (4 tap groups)
    geometries; ragged M and Cout.  How a K tile reaches the MFMAs: b32 = registers -> transposed LDS tile -> ds_read_b32
    fragments (the fallback family); glds = LDS-direct loads (buffer_load ... lds)."""
    how = {"b32": dict(CG_NN_GLDS=0), "glds": dict(CG_NN_GLDS=3)}[stage]
    with options(cg, CG_NN_TILE=tile, CG_GEMM_BK32=bk32, CG_NN_SPLITS=splits, CG_SKINNY=0, **how):
        run_conv(cg, 3, 64, 10, 6, 72, 3, 0, seed=tile % 97)          # ragged M = 180, Cout = 72
        run_conv(cg, 2, 32, 8, 8, 128, 3, 1, seed=tile % 89, wino=False)
        run_conv(cg, 2, 64, 4, 4, 32, 5, 1, seed=tile % 83, wino=False)


@pytest.mark.parametrize("tile", NN_TILES)
@pytest.mark.parametrize("splits", [0, 3])
@pytest.mark.parametrize("glds", [0, 3])
def test_lean_epilogue_on_full_tiles(cg, tile, splits, glds):
    """nn_store_lean (gemm.hip): FULL tiles with consecutive output rows take the buffer-store epilogue (direct output and split-K
    partials), everything else the generic one.  Shapes whose M and Cout are multiples of every block tile, both staging families,
    with and without split-K, against the oracle; then conv -> PReLU through the planned executor (activation in the epilogue, two
    outputs) against the per-module walk (convolution, then cg_prelu_forward), which must agree bit for bit."""
    with options(cg, CG_NN_TILE=tile, CG_NN_SPLITS=splits, CG_NN_GLDS=glds, CG_SKINNY=0):
        run_conv(cg, 4, 64, 16, 16, 128, 3, 0, seed=11)          # M = 1024, Cout = 128
        run_conv(cg, 2, 32, 16, 16, 64, 5, 0, seed=12)           # M = 512, Cout = 64
        outs = []
        for planned in (True, False):
            cg.nn.planned = planned
            try:
                cg.manual_seed(3)
                net = cg.nn.Sequential()
                net.add(cg.nn.SpatialConvolution(64, 64, 3, 3, 1, 1, 1)); net.add(cg.nn.PReLU(None, None, True))
                net.add(cg.nn.SpatialConvolution(64, 128, 3, 3, 1, 1, 1)); net.add(cg.nn.LeakyReLU(0.2))
                p, g = net.getParameters()
                x = cg.nn.as_nhwc(cg.Tensor.from_numpy(np.random.RandomState(5).randn(4, 64, 16, 16).astype(f32)))
                y = cg.nn.as_plain(net.forward(x)).numpy()
                dy = cg.Tensor.from_numpy(np.random.RandomState(6).randn(*y.shape).astype(f32))
                g.zero()
                gi = cg.nn.as_plain(net.backward(x, dy)).numpy()
                outs.append((y, gi, g.numpy().copy()))
            finally:
                cg.nn.planned = True
        np.testing.assert_array_equal(outs[0][0], outs[1][0])
        np.testing.assert_array_equal(outs[0][1], outs[1][1])
        close(outs[0][2], outs[1][2], K=1024, tol=4e-5, what="flat gradient planned vs per-module")


@pytest.mark.parametrize("splits", [0, 1, 3, 8])
@pytest.mark.parametrize("shape", [(64, 128, 8, 128, 7), (128, 64, 16, 128, 5), (64, 64, 8, 64, 3), (64, 32, 4, 64, 5)])
def test_position_major_tiles_skip_the_padding_taps(cg, shape, splits):
    """igemm_nng_kernel<64, BN, ..., PM> (gemm.hip, Geom::pmn; round 5): a plain stride-1 convolution whose zero padding is a real share
    of its MACs runs with POSITION-major row tiles - 64 images at one pixel position - whose rows share their padding taps, and the K
    loop walks the valid taps only (D32_st3's 7x7 layer at 8x8, models.lua:685: 38 % of the MACs; the 5x5 layer at 16x16: 14 %).
    Forward and data gradient (= the same kernel on dy with the flipped kernel) against the oracle, unsplit and with the valid K range
    of every tile cut into 3 / 8 splits (more splits than a corner tile has K tiles included); then against the image-major form
    (CG_PAD_SKIP = 0): equal up to the re-association of the K splits, and BIT-equal unsplit - a skipped product is an exact zero."""
    N, Cin, H, Cout, k = shape
    with options(cg, CG_PAD_SKIP=1, CG_NN_TILE=64000 + (128 if Cout >= 128 else 64), CG_NN_SPLITS=splits, CG_SKINNY=0):
        run_conv(cg, N, Cin, H, H, Cout, k, 0, seed=5 + k, wino=False)
    if splits > 1:
        return
    outs = []
    for thr in (1, 0):
        with options(cg, CG_PAD_SKIP=thr, CG_NN_TILE=64000 + (128 if Cout >= 128 else 64), CG_NN_SPLITS=1, CG_SKINNY=0):
            rs = np.random.RandomState(3)
            m = cg.nn.SpatialConvolution(Cin, Cout, k, k, 1, 1, (k - 1) // 2)
            m.weight.copy((rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(f32)); m.bias.copy(rs.randn(Cout).astype(f32))
            x = cg.Tensor.from_numpy(rs.randn(N, Cin, H, H).astype(f32))
            y = m.forward(x).numpy().copy()
            gi = m.updateGradInput(x, cg.Tensor.from_numpy(rs.randn(*y.shape).astype(f32))).numpy().copy()
            outs.append((y, gi))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("splits", [0, 3, 16])
@pytest.mark.parametrize("shape", [(32, 128, 8, 128, 7), (16, 64, 16, 128, 5), (48, 64, 8, 64, 3), (16, 16, 16, 16, 3), (32, 64, 6, 64, 5)])
def test_position_major_weight_gradient_skips_the_padding(cg, shape, splits):
    """igemm_tng_kernel mode 2 (gemm.hip; round 6): the weight gradient's K tiles are ONE grid position of 16 images, and a workgroup -
    whose dW rows belong to one or two taps (eight with 16 input planes) - visits only the rectangle of positions at which one of its
    taps reads inside the image (models.lua:681,685: 14 % / 38 % of the MACs multiply padding).  gradWeight and gradBias (which rides on
    the centre tap's tile) against the oracle: the plan's splits and 3 / 16 splits of every workgroup's own tile list (more
    splits than a corner tap has tiles included); a 6 x 6 map (no power-of-two geometry needed); then against the image-major tiles
    (CG_PAD_SKIP = 0) - equal up to the order of the pixel sum."""
    N, Cin, H, Cout, k = shape
    gws = []
    for thr in (1, 0):
        with options(cg, CG_PAD_SKIP=thr, CG_TN_SPLITS=splits, CG_SKINNY=0):
            m = run_conv(cg, N, Cin, H, H, Cout, k, 0, seed=9 + k, wino=False, check_dgrad=False)
            gws.append((m.gradWeight.numpy().copy(), m.gradBias.numpy().copy()))
    close(gws[0][0], gws[1][0], K=N * H * H, tol=4e-5, what="gradWeight position-major vs image-major")
    close(gws[0][1], gws[1][1], K=N * H * H, tol=4e-5, what="gradBias position-major vs image-major")


@pytest.mark.parametrize("N", [16, 64, 128])
def test_view_linear_head_weight_gradient_straight_into_gradweight(cg, N):
    """models.lua:696-697: View(320*8*8) -> Linear(20480, 256) on the NHWC map runs as an 8 x 8 convolution with a 1 x 1 output grid
    (cg_pack_conv_weight_map); its weight gradient is ONE kernel that adds into the canonical gradWeight[co][c][tap] (headwg.hip: the
    batch is the K dimension, an accumulator register holds 16 taps of one plane).  Against a float64 product, accumulate semantics
    included, and against CG_PAD_SKIP = 0 (the register-staged igemm_tn_kernel + the transposing reduction it replaces)."""
    C, H, Co = 320, 8, 256
    rs = np.random.RandomState(21)
    w = (rs.randn(Co, C * H * H) / np.sqrt(C * H * H)).astype(f32)
    b = rs.randn(Co).astype(f32)
    x = rs.randn(N, C, H, H).astype(f32)
    dy = rs.randn(N, Co).astype(f32)
    gw_ref = dy.T.astype(np.float64) @ x.reshape(N, -1).astype(np.float64)
    got = []
    for thr in (20, 0):
        with options(cg, CG_PAD_SKIP=thr):
            net = cg.nn.Sequential()
            net.add(cg.nn.View(C * H * H)); lin = cg.nn.Linear(C * H * H, Co); net.add(lin)
            lin.weight.copy(w); lin.bias.copy(b)
            p, g = net.getParameters()
            xin = cg.nn.as_nhwc(cg.Tensor.from_numpy(x))
            y = net.forward(xin).numpy()
            close(y, x.reshape(N, -1) @ w.T + b, K=C * H * H, what="View -> Linear forward")
            g.zero()
            net.backward(xin, cg.Tensor.from_numpy(dy))
            gw = lin.gradWeight.numpy().copy()
            close(gw, gw_ref, K=N, tol=4e-5, what="gradWeight")
            close(lin.gradBias.numpy(), dy.sum(0), K=N, tol=4e-5, what="gradBias")
            net.backward(xin, cg.Tensor.from_numpy(dy))          # accGradParameters accumulates
            close(lin.gradWeight.numpy(), 2.0 * gw_ref, K=N, tol=4e-5, what="gradWeight accumulated twice")
            close(lin.gradBias.numpy(), 2.0 * dy.sum(0), K=N, tol=4e-5, what="gradBias accumulated twice")
            got.append(gw)
    close(got[0], got[1], K=N, tol=4e-5, what="gradWeight position-major vs register-staged")


@pytest.mark.parametrize("swizzle", [0, 1, 3, 7])
def test_xcd_tile_orders(cg, swizzle):
    """CG_XCD_SWIZZLE: which tile a workgroup takes is a pure relabelling - natural order (0), contiguous row ranges per XCD (bit 0; bit 1
    = pixel chunks per XCD in the weight-gradient kernels), weights-stationary XCDs in igemm_nng_kernel (bit 2, default since round 5:
    conv1's forward 161 -> 68 MB of fabric traffic, profiles/r05_pmc_kernels.json).  Shapes with 2 and 4 column tiles whose tile counts are
    multiples of 8, phase-folded and plain, with the batch-norm statistics rows of the epilogue in the plan's order."""
    with options(cg, CG_XCD_SWIZZLE=swizzle, CG_NN_TILE=64128, CG_SKINNY=0):
        run_conv(cg, 32, 64, 4, 4, 256, 3, 1, seed=21, wino=False)     # 4 phases x (8 row tiles x 2 column tiles)
        run_conv(cg, 16, 32, 8, 8, 512, 3, 0, seed=22)                 # 16 row tiles x 4 column tiles


def run_linear(cg, N, i, o, seed=0):
    rs = np.random.RandomState(seed)
    m = cg.nn.Linear(i, o)
    w = (rs.randn(o, i) / np.sqrt(i)).astype(f32); b = rs.randn(o).astype(f32)
    m.weight.copy(w); m.bias.copy(b)
    x = rs.randn(N, i).astype(f32); dy = rs.randn(N, o).astype(f32)
    close(m.forward(cg.Tensor.from_numpy(x)).numpy(), O.linear_forward(x, w, b), K=i, what="forward")
    m.gradWeight.zero(); m.gradBias.zero()
    gi = m.backward(cg.Tensor.from_numpy(x), cg.Tensor.from_numpy(dy)).numpy()
    close(gi, O.linear_backward_data(dy, w), K=o, what="gradInput")
    gw, gb = np.zeros_like(w), np.zeros_like(b)
    O.linear_backward_weight(x, dy, gw, gb)
    close(m.gradWeight.numpy(), gw, K=N, what="gradWeight"); close(m.gradBias.numpy(), gb, K=N, what="gradBias")


@pytest.mark.parametrize("tile", NN_TILES)
@pytest.mark.parametrize("splits", [0, 5])
@pytest.mark.parametrize("stage", ["b32", "glds"])
def test_forced_tn_tile_variants(cg, tile, splits, stage):
    """igemm_tn_kernel<BM,BN> / igemm_tng_kernel<BM,BN> (weight gradient) for every block tile, default and forced pixel splits;
    the lean power-of-two addressing (16x16 grid, and the 4 phases of a folded upsampling), the generic one (10x6 grid: always
    the b32 kernel) and the flat rows of a linear layer.  glds: LDS-direct loads."""
    how = {"b32": dict(CG_TN_GLDS=0), "glds": dict(CG_TN_GLDS=1)}[stage]
    with options(cg, CG_TN_TILE=tile, CG_TN_SPLITS=splits, CG_SKINNY=0, **how):
        run_conv(cg, 3, 64, 10, 6, 72, 3, 0, seed=tile % 97, check_dgrad=False)
        run_conv(cg, 2, 64, 16, 16, 64, 3, 0, seed=tile % 89, check_dgrad=False)
        run_conv(cg, 2, 32, 8, 8, 128, 3, 1, seed=tile % 83, wino=False, check_dgrad=False)
        run_linear(cg, 96, 136, 72, seed=tile % 79)


def test_generic_gather_variants(cg):
    """The non-FAST (scalar gather) and non-vector-B template instances: Cin % 16 != 0, Cout % 4 != 0."""
    run_conv(cg, 2, 24, 8, 8, 10, 3, 0, seed=1)
    run_conv(cg, 2, 16, 8, 8, 10, 3, 0, seed=2)
    run_conv(cg, 2, 24, 8, 8, 64, 3, 0, seed=3)


@pytest.mark.parametrize("bk,stage", [(16, "b32"), (32, "b32"), (16, "glds"), (32, "glds")])
def test_forced_winograd_variants(cg, bk, stage):
    """wino_gemm_kernel<8,16>, <8,32> (register staging, the fallback) and the LDS-direct-load kernels wino_gemm_g_kernel<16>, <32> on
    the F(2x2,3x3) path of upsample2 -> conv5x5 (models.lua:217-218), forward + data gradient + weight gradient (the Winograd-domain
    weight gradient runs on the TN kernels: glds there too), ragged tile count."""
    how = {"b32": dict(CG_WINO_GLDS=0, CG_TN_GLDS=0), "glds": dict(CG_WINO_GLDS=1, CG_TN_GLDS=1)}[stage]
    cg.nn.SpatialConvolution.winograd_min_tiles = 0
    try:
        with options(cg, CG_WINO_BK=bk, **how):
            m = run_conv(cg, 3, 128, 6, 4, 128, 5, 1, seed=8 + bk)
            assert getattr(m, "_wino", False)
            m = run_conv(cg, 2, 256, 8, 8, 128, 5, 1, seed=8 * bk)
            assert getattr(m, "_wino", False)
    finally:
        cg.nn.SpatialConvolution.winograd_min_tiles = 2048


# ------------------------------------------------------------------ the training step: gradients and Adam moments
def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _grad_report(tag, a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b)
    scale = max(float(np.abs(b).max()), 1e-30)
    q = np.quantile(d, [0.5, 0.99, 0.999])
    print(f"[grad] {tag}: rel-l2 {_rel(a, b):.2e} median {q[0] / scale:.2e} p99 {q[1] / scale:.2e} p99.9 {q[2] / scale:.2e} "
          f"max {d.max() / scale:.2e} (scale {scale:.2e})")
    return q / scale, d.max() / scale


def _outliers(tag, a, b, slices):
    """{threshold: number of entries with |a - b| > threshold * max|b|} for 1e-4 and 1e-3, the worst ones listed per parameter tensor."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b) / max(float(np.abs(b).max()), 1e-30)
    out = {t: int((d > t).sum()) for t in (1e-4, 1e-3)}
    if out[1e-4]:
        per = []
        for off, n, shape in slices:
            c4, c3 = int((d[off:off + n] > 1e-4).sum()), int((d[off:off + n] > 1e-3).sum())
            if c4:
                per.append(f"{shape}@{off}: {c4} > 1e-4" + (f", {c3} > 1e-3" if c3 else "") + f" of {n} (max {d[off:off + n].max():.1e})")
        print(f"[outliers] {tag}: {out[1e-4]} entries > 1e-4, {out[1e-3]} > 1e-3 of {d.size}; " + "; ".join(per[:12]))
    return out


def _teacher_force(cg, S, T, G, Go):
    """Put the engine into the oracle's state: parameters, Adam m / v / t and the BN running statistics.  With this in front
    of every step each one is a "step 0": engine and oracle see identical parameters, so the tight single-step bounds apply
    to all of them and a bug that only shows with non-initial parameters / moments / statistics cannot hide behind the
    drift Adam's sign flips cause."""
    S.PARAMETERS_D.copy(T.pD); S.PARAMETERS_G.copy(T.pG)
    for key, st_o in (("D", T.stD), ("G", T.stG)):
        st_e = S.OPTSTATE["adam"][key]
        st_e["m"].copy(st_o["m"]); st_e["v"].copy(st_o["v"])
        assert st_e["t"] == st_o["t"]
    bns_e = [m for m in G.listModules() if isinstance(m, cg.nn.SpatialBatchNormalization)]
    bns_o = [m for m in Go.modules() if isinstance(m, O.SBN)]
    assert len(bns_e) == len(bns_o) > 0
    for e, o in zip(bns_e, bns_o):
        e.running_mean.copy(o.running_mean); e.running_var.copy(o.running_var)


@pytest.mark.parametrize("cfg,N", [("c2", 8), ("c2", 128), ("c3", 256), ("c5", 64)])
def test_training_steps_gradients_and_adam_state(cg, cfg, N):
    """adversarial.lua:51-275 x3 against the oracle Trainer on identical batches / noise / masks, comparing what a
    wrong gradient cannot hide in: the flat gradient optim.adam receives (after penalty and clamp, :92-112) and Adam's
    m / v after the update, for D and for G, at EVERY step, tight on the bulk of the entries and per parameter tensor.
    c2 / N = 128 is BASELINE configs[1] itself (the oracle needs ~20 s per step there); c3 / N = 256 is configs[2]: G32up on
    one grey plane at batch 256; c5 / N = 64 is the per-GPU share of configs[4]: G32up-c@64 + D32_st3@64 (models.lua:645,654,
    696-697) at 64 images.

    Every step is teacher-forced (`_teacher_force`): before steps 1.. the oracle's parameters, Adam moments and BN running
    statistics are copied into the engine, so all steps are held to the bounds of a first step while running on
    non-initial parameters / moments / statistics (engine and oracle trajectories would otherwise separate by +-2 lr on the
    few weights whose gradient was ~0 - Adam's first update is lr*sign(g) - and later steps could only be checked loosely).

    Why not exact: engine and oracle are both fp32 with different summation orders, so activations differ by ~1e-6
    relative; an activation within that distance of a PReLU / max-pool / clamp kink takes the other branch, which moves
    a few gradient entries by O(1e-3) of the scale (bounded below: p99.9 and max)."""
    seed = 31
    cg.manual_seed(seed); rng = O.RNG(seed)
    C, size = (1 if cfg == "c3" else 3), (64 if cfg == "c5" else 32)
    if cfg == "c2":
        G, Go = cg.models.create_G((C, 32, 32), 100), O.create_G32up_c(C, 100, rng)
    elif cfg == "c5":
        G, Go = cg.models.create_G((C, 64, 64), 100), O.create_G32up_c(C, 100, rng, base=8)
    else:
        G, Go = cg.models.create_G_decoder_upsampling32((C, 32, 32), 100), O.create_G32up(C, 100, rng)
    D, Do = cg.models.create_D((C, size, size)), O.create_D32_st3(C, size, rng)
    S = cg.adversarial.State(dict(batchSize=N), G, D)
    S.keep_outputs = True
    T = O.Trainer(Go, Do)
    np.testing.assert_array_equal(S.PARAMETERS_D.numpy(), T.pD)
    np.testing.assert_array_equal(S.PARAMETERS_G.numpy(), T.pG)
    rs = np.random.RandomState(9)
    P = 2 * N
    pool = rs.rand(P, C, size, size).astype(f32)
    data = cg.adversarial.TrainData(pool)
    steps = 3 if cfg == "c2" else 2
    slices = {"D": [], "G": []}
    for key, net in (("D", Do), ("G", Go)):
        off = 0
        for p_, _ in net.parameters():
            slices[key].append((off, p_.size, p_.shape)); off += p_.size
    bns_e = [m for m in G.listModules() if isinstance(m, cg.nn.SpatialBatchNormalization)]
    bns_o = [m for m in Go.modules() if isinstance(m, O.SBN)]
    for step in range(steps):
        if step > 0:
            _teacher_force(cg, S, T, G, Go)
        idx = rs.randint(0, P, size=N // 2)
        nd = (rs.rand(N // 2, 100) * 2 - 1).astype(f32); ng = (rs.rand(N, 100) * 2 - 1).astype(f32)
        cg.adversarial.iteration(S, data, N, real_idx=idx, noise_D=nd, noise_G=ng)
        r = T.step(pool[idx], nd, ng)
        for key, g_eng, g_orc, st_e, st_o in (("D", S._last["gD"].numpy(), r["gD"], S.OPTSTATE["adam"]["D"], T.stD),
                                              ("G", S._last["gG"].numpy(), r["gG"], S.OPTSTATE["adam"]["G"], T.stG)):
            q, mx = _grad_report(f"{cfg} N={N} step {step} g{key}", g_eng, g_orc)
            # bulk: half of the entries agree to fp32 rounding of a long sum, 99 % to 5e-5 of the largest entry (measured, rounds 3-4:
            # median <= 6e-7, p99 <= 8e-6, p99.9 <= 1.5e-5, max <= 1.3e-3; profiles/r03_step_gradients_vs_oracle.txt)
            assert q[0] <= 5e-6, f"g{key} step {step}: median rel diff {q[0]:.2e}"
            assert q[1] <= 5e-5, f"g{key} step {step}: p99 rel diff {q[1]:.2e}"
            assert q[2] <= 2e-4, f"g{key} step {step}: p99.9 rel diff {q[2]:.2e}"
            assert mx <= 1e-2, f"g{key} step {step}: max rel diff {mx:.2e}"
            assert _rel(g_eng, g_orc) <= 2e-3, f"g{key} step {step}: rel l2 {_rel(g_eng, g_orc):.2e}"
            # the outliers are COUNTED, not only bounded: kink flips (an activation within fp32 rounding of a PReLU / max-pool / clamp
            # corner) move isolated entries; a wrong tap or a mis-indexed plane would move a whole row of them
            bad = _outliers(f"{cfg} N={N} step {step} g{key}", g_eng, g_orc, slices[key])
            assert bad[1e-4] <= 1e-3 * g_orc.size, f"g{key} step {step}: {bad[1e-4]} entries beyond 1e-4 of the scale"
            assert bad[1e-3] <= 16, f"g{key} step {step}: {bad[1e-3]} entries beyond 1e-3 of the scale"
            # per parameter tensor (a wrong layer cannot hide behind the big ones)
            for off, n, shape in slices[key]:
                a, b = g_eng[off:off + n], g_orc[off:off + n]
                nb = float(np.linalg.norm(b))
                if nb < 1e-6 * max(float(np.linalg.norm(g_orc)), 1e-30) or n < 2:
                    continue   # e.g. the zero-initialised transformer classifiers' inputs, single PReLU slopes
                e = _rel(a, b)
                assert e <= 2e-2, f"g{key} step {step} tensor {shape} @{off}: rel l2 {e:.2e}"
            # Adam moments after the update: m and v are linear / quadratic in the gradients seen so far
            m_e, v_e = st_e["m"].numpy(), st_e["v"].numpy()
            assert st_e["t"] == st_o["t"] == step + 1
            em, ev = _rel(m_e, st_o["m"]), _rel(v_e, st_o["v"])
            print(f"[adam] {cfg} N={N} step {step} {key}: rel-l2 m {em:.2e} v {ev:.2e}")
            assert em <= 2e-3 and ev <= 4e-3, (key, step, em, ev)
        # single-weight tensors: PReLU slopes are one number each (a cancelling sum over every activation of the layer, so
        # fp32 summation-order noise is relative to the largest entries, not to the sum) - compare them directly
        for key, g_eng, g_orc in (("D", S._last["gD"].numpy(), r["gD"]), ("G", S._last["gG"].numpy(), r["gG"])):
            for off, n, shape in slices[key]:
                if n == 1:
                    a, b = float(g_eng[off]), float(g_orc[off])
                    assert abs(a - b) <= 5e-2 * abs(b) + 2e-5 * float(np.abs(g_orc).max()), (key, step, off, a, b)
        d_img = np.abs(S._last_fake.numpy() - r["fake"]).max()
        assert d_img <= 2e-4, f"step {step}: fake images differ by {d_img:.2e}"
        # BN running statistics (written by the step, read only by evaluate-mode sampling): two train-mode forwards per step
        for e, o in zip(bns_e, bns_o):
            close(e.running_mean.numpy(), o.running_mean, tol=2e-5, what=f"step {step} running_mean")
            close(e.running_var.numpy(), o.running_var, tol=2e-5, what=f"step {step} running_var")
