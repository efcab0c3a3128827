"""Generic MFMA GEMM (csrc/gemm.hip) vs numpy float64: every operand layout, edge shapes, batching,
epilogues, split-K atomics, both compute types."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _run(M, N, K, a_k, b_k, bf16, flags_extra=0, nz=(1, 1), splitk=1, beta=0.0, alpha=1.0, seed=0, src_bf16=False):
    from ttmi import ops
    rng = np.random.default_rng(seed)
    nb = nz[0] * nz[1]
    A = rng.integers(-4, 5, size=(nb, M, K)).astype(np.float32) if seed == 99 else rng.normal(size=(nb, M, K)).astype(np.float32)
    B = rng.integers(-4, 5, size=(nb, N, K)).astype(np.float32) if seed == 99 else rng.normal(size=(nb, N, K)).astype(np.float32)
    C0 = rng.normal(size=(nb, M, N)).astype(np.float32)
    bias = rng.normal(size=(N,)).astype(np.float32)
    aux = rng.normal(size=(nb, M, N)).astype(np.float32)
    Am = A if a_k else np.ascontiguousarray(A.transpose(0, 2, 1))     # stored [M,K] or [K,M]
    Bm = B if b_k else np.ascontiguousarray(B.transpose(0, 2, 1))     # stored [N,K] or [K,N]
    dt = torch.bfloat16 if src_bf16 else torch.float32
    tA = torch.tensor(Am, device="cuda").to(dt)
    tB = torch.tensor(Bm, device="cuda").to(dt)
    if src_bf16:
        A = tA.float().cpu().numpy() if a_k else tA.float().cpu().numpy().transpose(0, 2, 1)
        B = tB.float().cpu().numpy() if b_k else tB.float().cpu().numpy().transpose(0, 2, 1)
    tC = torch.tensor(C0, device="cuda")
    flags = flags_extra | (ops.GEMM_A_KMAJOR if a_k else 0) | (ops.GEMM_B_KMAJOR if b_k else 0) | (ops.GEMM_BF16_MFMA if bf16 else 0)
    tb = torch.tensor(bias, device="cuda") if flags & ops.GEMM_BIAS else None
    ta = torch.tensor(aux, device="cuda") if flags & ops.GEMM_MASK_AUX else None
    lda = K if a_k else M
    ldb = K if b_k else N
    ops.gemm(tA, tB, tC, M, N, K, lda, ldb, N, flags, bias=tb, aux=ta, alpha=alpha, beta=beta, nz1=nz[0], nz2=nz[1],
             sA=(nz[1] * M * K, M * K), sB=(nz[1] * N * K, N * K), sC=(nz[1] * M * N, M * N), splitk=splitk)
    torch.cuda.synchronize()
    want = alpha * np.einsum("zmk,znk->zmn", A.astype(np.float64), B.astype(np.float64))
    if flags & ops.GEMM_BIAS:
        want = want + bias
    if flags & ops.GEMM_ATOMIC:
        want = want + C0
    else:
        want = want + beta * C0
        if flags & ops.GEMM_RELU:
            want = np.maximum(want, 0)
        if flags & ops.GEMM_MASK_AUX:
            want = np.where(aux > 0, want, 0)
    return tC.cpu().numpy(), want


LAYOUTS = [(True, True), (True, False), (False, False), (False, True)]


@pytest.mark.parametrize("a_k,b_k", LAYOUTS)
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 130, 51), (37, 300, 17), (1, 5, 3), (129, 257, 100), (64, 36, 7)])
def test_layouts_and_edges(a_k, b_k, bf16, M, N, K):
    got, want = _run(M, N, K, a_k, b_k, bf16, seed=M + N + K)
    assert rel_err(got, want) < (1e-2 if bf16 else 2e-6)


@pytest.mark.parametrize("a_k,b_k", LAYOUTS)
def test_exact_small_integers_bf16(a_k, b_k):
    """integer operands are exact in bf16: catches any fragment/lane-map permutation error bit-for-bit"""
    got, want = _run(160, 96, 80, a_k, b_k, True, seed=99)
    assert np.array_equal(got, want.astype(np.float32))


@pytest.mark.parametrize("bf16", [False, True])
def test_epilogues(bf16):
    from ttmi import ops
    tol = 1e-2 if bf16 else 2e-6
    got, want = _run(150, 70, 40, True, True, bf16, flags_extra=ops.GEMM_BIAS | ops.GEMM_RELU, seed=1)
    assert rel_err(got, want) < tol
    got, want = _run(150, 70, 40, True, False, bf16, flags_extra=ops.GEMM_MASK_AUX, seed=2)
    assert rel_err(got, want) < tol
    got, want = _run(150, 70, 40, True, True, bf16, beta=1.0, alpha=0.5, seed=3)
    assert rel_err(got, want) < tol
    got, want = _run(90, 70, 1000, False, False, bf16, flags_extra=ops.GEMM_ATOMIC, splitk=5, seed=4)
    assert rel_err(got, want) < tol
    got, want = _run(90, 70, 1000, False, False, bf16, flags_extra=ops.GEMM_ATOMIC | ops.GEMM_BIAS, splitk=3, seed=5)
    assert rel_err(got, want) < tol


@pytest.mark.parametrize("a_k,b_k", LAYOUTS[:3])
def test_batched(a_k, b_k):
    got, want = _run(70, 45, 36, a_k, b_k, False, nz=(3, 4), seed=7)
    assert rel_err(got, want) < 2e-6


@pytest.mark.parametrize("a_k,b_k", LAYOUTS[:3])
def test_bf16_sources(a_k, b_k):
    got, want = _run(130, 140, 72, a_k, b_k, True, seed=8, src_bf16=True)
    assert rel_err(got, want) < 1e-5      # operands exactly representable -> only f32 accumulation error


@pytest.mark.parametrize("a_k,b_k", LAYOUTS)
@pytest.mark.parametrize("M,N,K", [(200, 130, 51), (129, 257, 100), (256, 128, 64)])
def test_three_term_bf16_layouts_and_edges(a_k, b_k, M, N, K):
    """GEMM_BF16X3 (TTMI_PRECISION=bf16x3: the attention core's batched products): f32 operands split into bf16 hi + lo while they are staged,
    hi.hi + lo.hi + hi.lo on the bf16 MFMA.  32 batches so that the K-major shapes pass the skinny-kernel routing and reach the 128 x 128 kernel."""
    from ttmi import ops
    got, want = _run(M, N, K, a_k, b_k, False, flags_extra=ops.GEMM_BF16X3, nz=(4, 8), seed=M + N + K)
    assert rel_err(got, want) < 2e-5
    plain, _ = _run(M, N, K, a_k, b_k, True, nz=(4, 8), seed=M + N + K)
    assert rel_err(got, want) < 0.02 * rel_err(plain, want)      # ... and it is not the one-term bf16 product


@pytest.mark.parametrize("a_k,b_k", LAYOUTS)
def test_three_term_bf16_exact_small_integers_and_epilogues(a_k, b_k):
    from ttmi import ops
    got, want = _run(160, 96, 80, a_k, b_k, False, flags_extra=ops.GEMM_BF16X3, nz=(8, 8), seed=99)    # lo = 0: bit-exact, any lane-map slip shows
    assert np.array_equal(got, want.astype(np.float32))
    got, want = _run(150, 70, 40, a_k, b_k, False, flags_extra=ops.GEMM_BF16X3 | ops.GEMM_BIAS | ops.GEMM_RELU, nz=(8, 8), seed=1)
    assert rel_err(got, want) < 2e-5
    got, want = _run(150, 70, 40, a_k, b_k, False, flags_extra=ops.GEMM_BF16X3, nz=(8, 8), beta=1.0, alpha=0.5, seed=3)
    assert rel_err(got, want) < 2e-5
    if not a_k:
        got, want = _run(90, 70, 1000, a_k, b_k, False, flags_extra=ops.GEMM_BF16X3 | ops.GEMM_ATOMIC, splitk=5, seed=4)
        assert rel_err(got, want) < 2e-5


@pytest.mark.parametrize("a_k", [True, False])
@pytest.mark.parametrize("M,K", [(500, 500), (130, 37), (257, 64), (128, 96)])
def test_three_term_bf16_panel64_kernel(a_k, M, K):
    """GEMM_BF16X3 with 64 n-major columns (P V, dS K, P^T dO ... of the attention core) leaves the 128 x 128 kernel for x3_panel64_kernel: the [M, K] slab
    straight from global memory into MFMA fragments, B through LDS; edge rows, K that is no multiple of the 32-wide strip (or of 4), beta = 1, atomics"""
    from ttmi import ops
    got, want = _run(M, 64, K, a_k, False, False, flags_extra=ops.GEMM_BF16X3, nz=(3, 4), seed=M + K)
    assert rel_err(got, want) < 2e-5
    got, want = _run(M, 64, K, a_k, False, False, flags_extra=ops.GEMM_BF16X3, nz=(3, 4), seed=99)
    assert np.array_equal(got, want.astype(np.float32))
    got, want = _run(M, 64, K, a_k, False, False, flags_extra=ops.GEMM_BF16X3, nz=(2, 2), beta=1.0, seed=5)
    assert rel_err(got, want) < 2e-5
    got, want = _run(M, 64, K, a_k, False, False, flags_extra=ops.GEMM_BF16X3 | ops.GEMM_ATOMIC, nz=(2, 2), seed=6)
    assert rel_err(got, want) < 2e-5


@pytest.mark.parametrize("M,N", [(500, 500), (33, 70), (256, 31), (40, 128), (70, 1100), (64, 513)])
def test_three_term_bf16_rows_nt64_kernel(M, N):
    """GEMM_BF16X3, both operands k-major with K = 64 (q E^T, (q + u) k^T, dO V^T): x3_rows_nt64_kernel - a workgroup writes 32 complete rows"""
    from ttmi import ops
    got, want = _run(M, N, 64, True, True, False, flags_extra=ops.GEMM_BF16X3, nz=(3, 4), seed=M + N)
    assert rel_err(got, want) < 2e-5
    got, want = _run(M, N, 64, True, True, False, flags_extra=ops.GEMM_BF16X3, nz=(3, 4), seed=99)
    assert np.array_equal(got, want.astype(np.float32))
    got, want = _run(M, N, 64, True, True, False, flags_extra=ops.GEMM_BF16X3 | ops.GEMM_BIAS, nz=(2, 2), beta=1.0, seed=7)
    assert rel_err(got, want) < 2e-5


def test_three_term_bf16_panel64_unaligned_slab():
    """the pitch-(L+1) view of dG (attention backward, dq += dG E): a k-major slab one float off 16-byte alignment with an odd pitch"""
    from ttmi import ops
    L, D, Z = 77, 64, 6
    g = torch.Generator(device="cuda").manual_seed(5)
    slab = torch.randn(Z * L * (L + 1) + 8, device="cuda", generator=g)
    e = torch.randn(L, D, device="cuda", generator=g)
    out = torch.randn(Z, L, D, device="cuda", generator=g)
    want = out.double() + torch.as_strided(slab[1:], (Z, L, L), (L * (L + 1), L + 1, 1)).double() @ e.double()
    ops.gemm(slab[1:], e, out, L, D, L, L + 1, D, D, ops.GEMM_A_KMAJOR | ops.GEMM_BF16X3, beta=1.0, nz1=2, nz2=3,
             sA=(3 * L * (L + 1), L * (L + 1)), sB=(0, 0), sC=(3 * L * D, L * D))
    torch.cuda.synchronize()
    assert float((out.double() - want).norm() / want.norm()) < 2e-5


def test_linearity_large():
    """size-independent property at a joint-sized K panel: GEMM(A1 + A2) == GEMM(A1) + GEMM(A2) (f32 path)"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 4096, 4334, 1024
    A1 = torch.randn(M, K, device="cuda", generator=g)
    A2 = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g)
    f = ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR
    out = [torch.empty(M, N, device="cuda") for _ in range(3)]
    for X, C in zip((A1, A2, A1 + A2), out):
        ops.gemm(X, W, C, M, N, K, K, K, N, f)
    err = (out[0] + out[1] - out[2]).norm() / out[2].norm()
    assert float(err) < 1e-5
    ref = A1.double() @ W.double().t()      # torch fp64 matmul as an independent check of one product
    assert float((out[0].double() - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("b_k", [True, False])
@pytest.mark.parametrize("M,N,K", [(1, 1536, 512), (64, 4334, 1024), (7, 33, 9), (128, 512, 1030), (33, 64, 64), (50, 42, 3)])
def test_skinny_f32_exact_integers(M, N, K, b_k):
    """the decoder-sized exact-f32 kernel (M <= 128, 32x32 tile, reduction split over 4 waves, operands straight from global memory):
    integer operands are exact, so any lane-map / k-pairing / tail error shows bit-for-bit; same numbers with the kernel switched off"""
    from ttmi import ops
    got, want = _run(M, N, K, True, b_k, False, seed=99)
    assert np.array_equal(got, want.astype(np.float32))
    ops.set_option(7, 0)
    try:
        got2, _ = _run(M, N, K, True, b_k, False, seed=99)
    finally:
        ops.set_option(7, 128)
    assert np.array_equal(got2, got)


def test_skinny_f32_epilogues_and_batches():
    from ttmi import ops
    got, want = _run(100, 70, 40, True, True, False, flags_extra=ops.GEMM_BIAS | ops.GEMM_RELU, seed=11)
    assert rel_err(got, want) < 2e-6
    got, want = _run(100, 70, 41, True, False, False, flags_extra=ops.GEMM_MASK_AUX, seed=12)
    assert rel_err(got, want) < 2e-6
    got, want = _run(65, 70, 40, True, True, False, beta=1.0, alpha=0.5, seed=13)
    assert rel_err(got, want) < 2e-6
    got, want = _run(51, 51, 64, True, True, False, nz=(2, 8), seed=14)           # per-head score products of a 51-token history
    assert rel_err(got, want) < 2e-6
    got, want = _run(51, 64, 51, True, False, False, nz=(1, 8), seed=15)          # per-head P.V (B stored [K, N])
    assert rel_err(got, want) < 2e-6


@pytest.mark.parametrize("M,N,K,pad", [(2048, 4334, 1024, 0), (1792, 4334, 1024, 2), (3360, 1536, 512, 0), (1024, 128, 64, 0), (3001, 700, 96, 4),
                                       (16000, 512, 2048, 0), (1025, 1024, 512, 0)])
def test_large_f32_nt_on_the_persistent_kernel_exact_integers(M, N, K, pad):
    """exact-f32 NT products with >= 1024 rows run on the persistent 256x128 LDS-DMA kernel with f32 operands (gemm_nt_f32: v_mfma_f32_16x16x4_f32
    over ds_read_b128 fragments): integer operands are exact in f32 whatever the reduction order, so any staging / k-pairing / epilogue error
    shows bit-for-bit - plain, bias + ReLU, and beta = 1 (accumulate into C), ragged M / N, padded pitches; same numbers with option 17 = 0"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randint(-4, 5, (M, K + 4 * pad), device="cuda", generator=g).float()
    B = torch.randint(-4, 5, (N, K + 4 * pad), device="cuda", generator=g).float()
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    C0 = torch.randint(-9, 10, (M + 1, N + pad), device="cuda", generator=g).float()
    f = ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR
    ref = A[:, :K].double() @ B[:, :K].double().t()
    outs = []
    for fast in (1, 0):
        ops.set_option(17, fast)
        try:
            C1, C2, C3 = C0.clone(), C0.clone(), C0.clone()
            ops.gemm(A, B, C1, M, N, K, A.stride(0), B.stride(0), C1.stride(0), f)
            ops.gemm(A, B, C2, M, N, K, A.stride(0), B.stride(0), C2.stride(0), f | ops.GEMM_BIAS | ops.GEMM_RELU, bias=bias)
            ops.gemm(A, B, C3, M, N, K, A.stride(0), B.stride(0), C3.stride(0), f | ops.GEMM_BIAS, bias=bias, beta=1.0)
        finally:
            ops.set_option(17, 1)
        outs.append((C1, C2, C3))
    C1, C2, C3 = outs[0]
    assert torch.equal(C1[:M, :N].double(), ref)
    assert torch.equal(C2[:M, :N].double(), (ref + bias.double()).clamp_min(0))
    assert torch.equal(C3[:M, :N].double(), ref + bias.double() + C0[:M, :N].double())
    for C in (C1, C2, C3):                                   # nothing outside the problem is touched
        assert torch.equal(C[M], C0[M]) and torch.equal(C[:, N:], C0[:, N:])
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_large_f32_nt_random_against_float64():
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 2048, 1536, 512
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(N, K, device="cuda", generator=g)
    C = torch.empty(M, N, device="cuda")
    ops.gemm(A, B, C, M, N, K, K, K, N, ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR)
    ref = A.double() @ B.double().t()
    assert float((C.double() - ref).norm() / ref.norm()) < 5e-7


@pytest.mark.parametrize("M,N,K,pad", [(800, 1536, 512, 0), (192, 1536, 512, 0), (1, 64, 32, 0), (65, 130, 96, 3), (333, 1023, 1024, 1), (2048, 4334, 1024, 0),
                                       (3360, 512, 1024, 0), (640, 514, 64, 2)])
def test_mid_f32_nt_64x64_tiles_exact_integers(M, N, K, pad):
    """the 64 x 64-tile exact-f32 kernel of the label-encoder-sized products (gemm_nt_f32_mid_kernel: LDS-DMA stages, v_mfma_f32_16x16x4_f32), forced
    with option 17 = 2: integer operands are exact whatever the reduction order - plain, bias + ReLU, accumulate; ragged M / N down to one row, pitches
    that are / are not multiples of 4 floats (vector / element-wise stores); rows and columns outside the problem untouched; equal to option 17 = 0"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K + pad)
    A = torch.randint(-4, 5, (M, K + 4 * pad), device="cuda", generator=g).float()
    B = torch.randint(-4, 5, (N, K + 4 * pad), device="cuda", generator=g).float()
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    C0 = torch.randint(-9, 10, (M + 1, N + pad), device="cuda", generator=g).float()
    f = ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR
    ref = A[:, :K].double() @ B[:, :K].double().t()
    outs = []
    for mode in (2, 0):
        ops.set_option(17, mode)
        try:
            C1, C2, C3 = C0.clone(), C0.clone(), C0.clone()
            ops.gemm(A, B, C1, M, N, K, A.stride(0), B.stride(0), C1.stride(0), f)
            ops.gemm(A, B, C2, M, N, K, A.stride(0), B.stride(0), C2.stride(0), f | ops.GEMM_BIAS | ops.GEMM_RELU, bias=bias)
            ops.gemm(A, B, C3, M, N, K, A.stride(0), B.stride(0), C3.stride(0), f | ops.GEMM_BIAS, bias=bias, beta=1.0)
        finally:
            ops.set_option(17, 1)
        outs.append((C1, C2, C3))
    C1, C2, C3 = outs[0]
    assert torch.equal(C1[:M, :N].double(), ref)
    assert torch.equal(C2[:M, :N].double(), (ref + bias.double()).clamp_min(0))
    assert torch.equal(C3[:M, :N].double(), ref + bias.double() + C0[:M, :N].double())
    for C in (C1, C2, C3):
        assert torch.equal(C[M], C0[M]) and torch.equal(C[:, N:], C0[:, N:])
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_mid_f32_nt_random_against_float64():
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(6)
    M, N, K = 800, 1536, 512
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(N, K, device="cuda", generator=g)
    C = torch.empty(M, N, device="cuda")
    ops.gemm(A, B, C, M, N, K, K, K, N, ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR)        # default rule: 13 x 24 tiles of 64 x 64
    ref = A.double() @ B.double().t()
    assert float((C.double() - ref).norm() / ref.norm()) < 5e-7
