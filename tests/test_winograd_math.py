"""CPU check of the transform matrices csrc/winograd.hip hard-codes (no GPU, no extension): F(2x2,3x3) on the phase
kernels of upsample2 -> conv5x5 reproduces the direct convolution of the materialised upsampled map, forward, data
gradient (flipped kernels over the four phase sub-lattices of dy) and weight gradient (A dY A^T, G^T . G)."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def phase_map(a, d, pad=2):
    return ((a + d - pad) >> 1) - ((a - pad) >> 1)


def phase_kernels(w):
    """w [5][5] -> g[p][3][3]: the taps of output phase p=(a,b) folded onto the low-res grid (gemm.hip pack_weight_ups2)."""
    g = np.zeros((4, 3, 3))
    for p in range(4):
        for dy in range(5):
            for dx in range(5):
                g[p, phase_map(p >> 1, dy), phase_map(p & 1, dx)] += w[dy, dx]
    return g


def direct(x, w):
    """conv5x5 pad 2 of the nearest-2x upsampled x (single plane)."""
    up = np.repeat(np.repeat(x, 2, 0), 2, 1)
    xp = np.pad(up, 2)
    H, W = up.shape
    return np.array([[np.sum(xp[i:i + 5, j:j + 5] * w) for j in range(W)] for i in range(H)])


def test_forward_and_gradients_match_direct_convolution():
    rs = np.random.RandomState(0)
    Hl = 6
    x, w = rs.randn(Hl, Hl), rs.randn(5, 5)
    y_ref = direct(x, w)
    g = phase_kernels(w)
    xp = np.pad(x, 1)                       # low-res rows -1 .. Hl
    dy = rs.randn(2 * Hl, 2 * Hl)
    y = np.zeros_like(y_ref)
    dg = np.zeros((4, 3, 3))
    for ti in range(Hl // 2):
        for tj in range(Hl // 2):
            d = xp[2 * ti:2 * ti + 4, 2 * tj:2 * tj + 4]      # rows 2ti-1 .. 2ti+2
            V = BT @ d @ BT.T
            for p in range(4):
                a, b = p >> 1, p & 1
                U = G @ g[p] @ G.T
                Y = AT @ (U * V) @ AT.T
                e = np.zeros((2, 2))
                for u in range(2):
                    for v in range(2):
                        y[2 * (2 * ti + u) + a, 2 * (2 * tj + v) + b] = Y[u, v]
                        e[u, v] = dy[2 * (2 * ti + u) + a, 2 * (2 * tj + v) + b]
                dM = AT.T @ e @ AT                          # A dY A^T
                dg[p] += G.T @ (V * dM) @ G                 # weight gradient of the phase kernel
    np.testing.assert_allclose(y, y_ref, rtol=0, atol=1e-12)
    # weight gradient: scatter the phase taps onto the canonical 5x5 taps (wino_wgrad_finish_kernel)
    dw = np.zeros((5, 5))
    for p in range(4):
        for ddy in range(5):
            for ddx in range(5):
                dw[ddy, ddx] += dg[p, phase_map(p >> 1, ddy), phase_map(p & 1, ddx)]
    up = np.pad(np.repeat(np.repeat(x, 2, 0), 2, 1), 2)
    dw_ref = np.array([[np.sum(up[i:i + 2 * Hl, j:j + 2 * Hl] * dy) for j in range(5)] for i in range(5)])
    np.testing.assert_allclose(dw, dw_ref, rtol=0, atol=1e-11)
    # data gradient w.r.t. the low-res input: 3x3 pad-1 correlation of each phase sub-lattice with the flipped phase kernel
    dx = np.zeros((Hl, Hl))
    for p in range(4):
        a, b = p >> 1, p & 1
        dp = np.pad(dy[a::2, b::2], 1)
        hflip = g[p][::-1, ::-1]
        for ti in range(Hl // 2):
            for tj in range(Hl // 2):
                V = BT @ dp[2 * ti:2 * ti + 4, 2 * tj:2 * tj + 4] @ BT.T
                dx[2 * ti:2 * ti + 2, 2 * tj:2 * tj + 2] += AT @ ((G @ hflip @ G.T) * V) @ AT.T
    # reference: adjoint of (upsample -> conv): correlate dy with the flipped 5x5 kernel, then sum 2x2 blocks
    dyp = np.pad(dy, 2)
    dup = np.array([[np.sum(dyp[i:i + 5, j:j + 5] * w[::-1, ::-1]) for j in range(2 * Hl)] for i in range(2 * Hl)])
    dx_ref = dup.reshape(Hl, 2, Hl, 2).sum(axis=(1, 3))
    np.testing.assert_allclose(dx, dx_ref, rtol=0, atol=1e-11)


# ---- F(2x2,2x2): upsample2 -> conv3x3 (pad 1); csrc/winograd.hip wino22_* ------------------------------------------------------
BT22 = np.array([[1, -1, 0], [0, 1, 0], [0, -1, 1]], dtype=np.float64)
G22 = np.array([[1, 0], [1, 1], [0, 1]], dtype=np.float64)
AT22 = np.array([[1, 1, 0], [0, 1, 1]], dtype=np.float64)


def phase_kernels_3x3(w):
    """w [3][3] (pad 1) -> g[p][2][2]: output phase (a, b) reads low-res rows -1 + a + {0, 1} (gemm.hip pack_weight_ups2)."""
    g = np.zeros((4, 2, 2))
    for p in range(4):
        for dy in range(3):
            for dx in range(3):
                g[p, phase_map(p >> 1, dy, 1), phase_map(p & 1, dx, 1)] += w[dy, dx]
    return g


def test_f22_on_the_phases_of_an_upsampled_3x3_convolution():
    """The minimal form (9 products per 2x2 tile and phase instead of 16) reproduces the direct convolution of the materialised
    upsampled map; the four phases' 3x3 windows are the corners of ONE 4x4 low-res patch (rows 2ti-1 .. 2ti+2), and the output
    transform's coefficients are the 0/1 pattern wino_gemm_g_kernel<.., 9> hard-codes: cy0 = (xi < 2), cy1 = (xi > 0)."""
    rs = np.random.RandomState(3)
    Hl = 6
    x, w = rs.randn(Hl, Hl), rs.randn(3, 3)
    up = np.pad(np.repeat(np.repeat(x, 2, 0), 2, 1), 1)
    y_ref = np.array([[np.sum(up[i:i + 3, j:j + 3] * w) for j in range(2 * Hl)] for i in range(2 * Hl)])
    g = phase_kernels_3x3(w)
    xp = np.pad(x, 1)
    y = np.zeros_like(y_ref)
    for i in range(3):
        assert AT22[0, i] == float(i < 2) and AT22[1, i] == float(i > 0)
    for ti in range(Hl // 2):
        for tj in range(Hl // 2):
            patch = xp[2 * ti:2 * ti + 4, 2 * tj:2 * tj + 4]
            for p in range(4):
                a, b = p >> 1, p & 1
                V = BT22 @ patch[a:a + 3, b:b + 3] @ BT22.T
                U = G22 @ g[p] @ G22.T
                Y = AT22 @ (U * V) @ AT22.T
                for u in range(2):
                    for v in range(2):
                        y[2 * (2 * ti + u) + a, 2 * (2 * tj + v) + b] = Y[u, v]
    np.testing.assert_allclose(y, y_ref, rtol=0, atol=1e-12)


def test_f22_data_gradient_and_weight_gradient_of_an_upsampled_3x3_convolution():
    """The backward forms csrc/winograd.hip runs for the same layer.  Data gradient (wino22_dy_input_transform_kernel + the 9-position GEMMs
    over K = 4 Cout): dx[u][v] = sum_p sum_r dy_p[u - a + ry][v - b + rx] . g_p[1-ry][1-rx] on the phase sub-lattices dy_p = dy[a::2, b::2] -
    per phase F(2x2,2x2) with the FLIPPED 2x2 kernel on the 3x3 window that starts at row 2ti - a, the four phases summed.  Weight gradient
    (the F(2x2,2x2)-domain form round 4 built and round 6 removed from the engine - no gain in the step; the identity stays): dU_p = sum_tiles V_p . (A dY_p A^T), A = [1 0; 1 1; 0 1];
    dg_p = G^T dU_p G; the phase taps scatter onto the canonical 3x3 taps through the same map that folded them."""
    rs = np.random.RandomState(4)
    Hl = 6
    x, w = rs.randn(Hl, Hl), rs.randn(3, 3)
    dy = rs.randn(2 * Hl, 2 * Hl)
    g = phase_kernels_3x3(w)
    # reference: adjoint of (upsample -> conv3x3 pad 1)
    dyp = np.pad(dy, 1)
    dup = np.array([[np.sum(dyp[i:i + 3, j:j + 3] * w[::-1, ::-1]) for j in range(2 * Hl)] for i in range(2 * Hl)])
    dx_ref = dup.reshape(Hl, 2, Hl, 2).sum(axis=(1, 3))
    up = np.pad(np.repeat(np.repeat(x, 2, 0), 2, 1), 1)
    dw_ref = np.array([[np.sum(up[i:i + 2 * Hl, j:j + 2 * Hl] * dy) for j in range(3)] for i in range(3)])
    dx = np.zeros((Hl, Hl))
    dg = np.zeros((4, 2, 2))
    xp = np.pad(x, 1)
    A = AT22.T
    for p in range(4):
        a, b = p >> 1, p & 1
        sub = dy[a::2, b::2]                                   # [Hl][Hl]
        subp = np.pad(sub, ((a, 2 - a), (b, 2 - b)))           # window of tile ti starts at sub-lattice row 2ti - a
        U = G22 @ g[p][::-1, ::-1] @ G22.T
        for ti in range(Hl // 2):
            for tj in range(Hl // 2):
                V = BT22 @ subp[2 * ti:2 * ti + 3, 2 * tj:2 * tj + 3] @ BT22.T
                dx[2 * ti:2 * ti + 2, 2 * tj:2 * tj + 2] += AT22 @ (U * V) @ AT22.T
                Vx = BT22 @ xp[2 * ti + a:2 * ti + a + 3, 2 * tj + b:2 * tj + b + 3] @ BT22.T     # the forward's V of this phase
                dM = A @ sub[2 * ti:2 * ti + 2, 2 * tj:2 * tj + 2] @ A.T
                dg[p] += G22.T @ (Vx * dM) @ G22
    np.testing.assert_allclose(dx, dx_ref, rtol=0, atol=1e-11)
    dw = np.zeros((3, 3))
    for p in range(4):
        for ddy in range(3):
            for ddx in range(3):
                dw[ddy, ddx] += dg[p, phase_map(p >> 1, ddy, 1), phase_map(p & 1, ddx, 1)]
    np.testing.assert_allclose(dw, dw_ref, rtol=0, atol=1e-11)
