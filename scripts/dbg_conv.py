"""Debug: one SpatialConvolution (forward / backward) against the oracle at a chosen shape, printing the worst element."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
cg = importlib.import_module("cat-generator_amd")
from oracle import oracle as O
f32 = np.float32
def run(N, Cin, H, W, Cout, k, ups):
    rs = np.random.RandomState(1); pad = (k - 1) // 2
    m = cg.nn.SpatialConvolution(Cin, Cout, k, k, 1, 1, pad)
    w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(f32); b = rs.randn(Cout).astype(f32)
    m.weight.copy(w); m.bias.copy(b)
    x = rs.randn(N, Cin, H, W).astype(f32); xin = cg.Tensor.from_numpy(x); xl = x
    if ups:
        up = cg.nn.SpatialUpSamplingNearest(2); xin = up.forward(xin); xl = np.repeat(np.repeat(x, 2, 2), 2, 3)
    y = m.forward(xin).numpy(); yo = O.conv2d_forward(xl, w, b, pad)
    dy = rs.randn(*yo.shape).astype(f32)
    m.gradWeight.zero(); m.gradBias.zero()
    gi = m.backward(xin, cg.Tensor.from_numpy(dy))
    gio = O.conv2d_backward_data(dy, w, xl.shape, pad)
    if ups: gi = up.updateGradInput(None, gi).numpy(); gio = O.UpSample2().backward(gio)
    else: gi = gi.numpy()
    gw, gb = np.zeros_like(w), np.zeros_like(b); O.conv2d_backward_weight(xl, dy, gw, gb, pad)
    e = lambda a, b_: float(np.abs(a - b_).max() / max(1.0, np.abs(b_).max()))
    print((N, Cin, H, W, Cout, k, ups), "fwd %.2e dgrad %.2e wgrad %.2e bias %.2e" % (e(y, yo), e(gi, gio), e(m.gradWeight.numpy(), gw), e(m.gradBias.numpy(), gb)), flush=True)
for c in [(2,3,8,8,5,3,0),(3,8,5,7,12,3,0),(2,16,8,8,16,3,0),(2,64,16,16,64,3,0),(2,512,4,4,512,3,1),(1,256,16,16,128,5,1),(2,8,4,4,4,5,1),(3,16,3,5,8,3,1),(2,128,32,32,3,3,0)]:
    run(*c)
