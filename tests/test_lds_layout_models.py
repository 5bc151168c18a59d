"""Host-side models of the LDS tile layouts of the GEMM kernels (csrc/gemm.hip, csrc/winograd.hip).

The kernels place a K tile in LDS by formulas (which slot a thread stores, which slot a lane's MFMA fragment reads, which
XOR keeps a group of lanes on distinct banks).  These tests restate the formulas in numpy for every compiled tile shape and
check two properties that do not need a GPU: (1) with `v_mfma_f32_32x32x2_f32` operand semantics (lane l supplies row l % 32,
k index l / 32) the tile product comes out as A @ B, i.e. stores and fragment reads agree and the k permutation is the same
for both operands; (2) every LDS instruction is conflict-free under the bank model of MI355X_MICROARCH.md (ds_read_b128:
four groups of 16 lanes, bank = 16-byte slot mod 16; ds_write_b128: contiguous groups of 8 lanes, slot mod 8; ds_read_b32:
two half-waves, dword mod 32).  The GPU parity tests hold the kernels themselves to the oracle; this file documents the
layouts and catches an edit that breaks one formula but not the other.
"""
import numpy as np
import pytest

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def worst(slots, groups, mod):
    return max(int(np.bincount([slots[l] % mod for l in g]).max()) for g in groups)


def mfma_accumulate(C, wm0, wn0, i, j, a_col, b_col):
    """One v_mfma_f32_32x32x2_f32: lanes 0..31 carry k index 0, lanes 32..63 k index 1 of their row / column."""
    for h in range(2):
        C[wm0 + i * 32:wm0 + i * 32 + 32, wn0 + j * 32:wn0 + j * 32 + 32] += np.outer(a_col[h * 32:(h + 1) * 32], b_col[h * 32:(h + 1) * 32])


NN_TILES = [(128, 128, 2, 2), (64, 128, 2, 2), (128, 64, 2, 2), (64, 64, 2, 2), (128, 32, 4, 1)]


@pytest.mark.parametrize("BM,BN,WM,WN", NN_TILES)
@pytest.mark.parametrize("BK", [16, 32])
def test_quad_layout_of_igemm_nn_kernel(BM, BN, WM, WN, BK):
    """igemm_nn_kernel<..., QUAD>: As[k/4][m ^ swz(k/4)][4], Bs[k/4][n ^ ((n >> 3) & 3)][4]; the same B-side formulas serve both
    operands of igemm_tnq_kernel."""
    rng = np.random.default_rng(BM + BN + BK)
    KV, NVEC = BK // 4, BN // 4
    ARPP, AROWS, NBLK = 256 // KV, BM // (256 // KV), KV * NVEC
    MI, NI, SH = BM // WM // 32, BN // WN // 32, (0 if KV == 8 else 1)
    A, B = rng.standard_normal((BM, BK)), rng.standard_normal((BK, BN))
    As, Bs = np.full((KV * BM, 4), np.nan), np.full((KV * BN, 4), np.nan)
    wa, wb = [dict() for _ in range(AROWS)], [dict() for _ in range(4)]
    for tid in range(256):
        a_kv, a_r = tid % KV, tid // KV
        for p in range(AROWS):
            row = a_r + ARPP * p
            slot = a_kv * BM + (row ^ ((a_kv & 7) << SH))
            assert np.isnan(As[slot]).all()
            As[slot] = A[row, 4 * a_kv:4 * a_kv + 4]
            wa[p][tid] = slot
        b_nv, b_kq = tid % NVEC, tid // NVEC
        if tid < NBLK:
            blk = B[4 * b_kq:4 * b_kq + 4, 4 * b_nv:4 * b_nv + 4]
            for i in range(4):
                slot = b_kq * BN + 4 * b_nv + (i ^ ((b_nv >> 1) & 3))
                assert np.isnan(Bs[slot]).all()
                Bs[slot] = blk[:, i]
                wb[i][tid] = slot
    assert not np.isnan(As).any() and not np.isnan(Bs).any()
    for w in wa + wb:   # ds_write_b128: contiguous 8-lane groups
        tids = sorted(w)
        for g0 in range(0, len(tids), 8):
            assert worst(w, [tids[g0:g0 + 8]], 8) == 1
    C = np.zeros((BM, BN))
    for wave in range(4):
        wm0, wn0 = (wave // WN) * (BM // WM), (wave % WN) * (BN // WN)
        for gq in range(KV // 2):
            qa, qb = {}, {}
            for lane in range(64):
                l31, h = lane & 31, lane >> 5
                kq = 2 * gq + h
                qa[lane] = kq * BM + wm0 + (l31 ^ ((kq & 7) << SH))
                qb[lane] = h * BN + wn0 + (l31 ^ ((l31 >> 3) & 3)) + gq * 2 * BN
            for i in range(MI):
                assert worst({l: qa[l] + i * 32 for l in qa}, B128_GROUPS, 16) == 1
            for j in range(NI):
                assert worst({l: qb[l] + j * 32 for l in qb}, B128_GROUPS, 16) == 1
            for s in range(4):
                for i in range(MI):
                    for j in range(NI):
                        mfma_accumulate(C, wm0, wn0, i, j, np.array([As[qa[l] + i * 32][s] for l in range(64)]),
                                        np.array([Bs[qb[l] + j * 32][s] for l in range(64)]))
    np.testing.assert_allclose(C, A @ B, atol=1e-12)


def glds_tile_product(BM, BN, BK, nwaves, wave_tile, a_waves):
    """LDS-direct loads: a wave instruction writes lane l's 16 bytes at base + 16 l.  A: row-major [row][BK/4 quads], the lane
    at position j of its row loads quad j ^ swz(row); B: [k][n] rows in place.  Returns (C, A @ B)."""
    rng = np.random.default_rng(BM * 3 + BN + BK)
    KV = BK // 4
    RPW, NVEC = 64 // KV, BN // 4
    BRPW = 64 // NVEC
    swz = (lambda r: (r >> 1) & 7) if KV == 8 else (lambda r: (r >> 2) & 3)
    A, B = rng.standard_normal((BM, BK)), rng.standard_normal((BK, BN))
    As, Bs = np.full((BM * KV, 4), np.nan), np.full(BK * BN, np.nan)
    a_instr = [(w, p) for p in range(BM // (RPW * a_waves)) for w in range(a_waves)]
    for w, p in a_instr:
        row0 = (p * a_waves + w) * RPW
        for lane in range(64):
            row, j = row0 + lane // KV, lane % KV
            q = j ^ swz(row)
            assert np.isnan(As[row0 * KV + lane]).all()
            As[row0 * KV + lane] = A[row, 4 * q:4 * q + 4]
    rows_per_pass = nwaves * BRPW
    for q in range((BK + rows_per_pass - 1) // rows_per_pass):
        for w in range(nwaves):
            row0 = q * rows_per_pass + w * BRPW
            if row0 >= BK:
                continue
            for lane in range(64):
                kr, nv = row0 + lane // NVEC, lane % NVEC
                Bs[row0 * BN + 4 * lane:row0 * BN + 4 * lane + 4] = B[kr, 4 * nv:4 * nv + 4]
    assert not np.isnan(As).any() and not np.isnan(Bs).any()
    C = np.zeros((BM, BN))
    for wave in range(nwaves):
        wm0, wn0, MI, NI = wave_tile(wave)
        for gq in range(KV // 2):
            qa = {l: (wm0 + (l & 31)) * KV + ((2 * gq + (l >> 5)) ^ swz(l & 31)) for l in range(64)}
            for i in range(MI):
                assert worst({l: qa[l] + i * 32 * KV for l in qa}, B128_GROUPS, 16) == 1
            for s in range(4):
                qb = {l: (8 * gq + 4 * (l >> 5) + s) * BN + wn0 + (l & 31) for l in range(64)}
                for j in range(NI):
                    assert worst({l: qb[l] + j * 32 for l in qb}, [list(range(32)), list(range(32, 64))], 32) == 1
                for i in range(MI):
                    for j in range(NI):
                        mfma_accumulate(C, wm0, wn0, i, j, np.array([As[qa[l] + i * 32 * KV][s] for l in range(64)]),
                                        np.array([Bs[qb[l] + j * 32] for l in range(64)]))
    return C, A @ B


@pytest.mark.parametrize("BM,BN,WM,WN", NN_TILES)
@pytest.mark.parametrize("BK", [16, 32])
def test_lds_direct_layout_of_igemm_nng_kernel(BM, BN, WM, WN, BK):
    MI, NI = BM // WM // 32, BN // WN // 32
    C, ref = glds_tile_product(BM, BN, BK, 4, lambda w: ((w // WN) * (BM // WM), (w % WN) * (BN // WN), MI, NI), 4)
    np.testing.assert_allclose(C, ref, atol=1e-12)


@pytest.mark.parametrize("BK", [16, 32])
def test_lds_direct_layout_of_wino_gemm_g_kernel(BK):
    """8 waves of 32 x 32 on a 64 x 128 tile; at BK 16 only waves 0..3 bring rows of V."""
    C, ref = glds_tile_product(64, 128, BK, 8, lambda w: ((w & 1) * 32, (w >> 1) * 32, 1, 1), 64 // (64 // (BK // 4)))
    np.testing.assert_allclose(C, ref, atol=1e-12)
