"""Throughput bf16 GEMMs (csrc/gemm_fast.hip: glds staging, XOR-swizzled LDS, ds_read_b64_tr_b16) vs exact references."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(shape, g):
    return torch.randint(-4, 5, shape, device="cuda", generator=g).to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 200, 192), (1000, 4334, 1024), (77, 130, 64),
                                   (2048, 300, 192), (1500, 4334, 1024), (1024, 128, 64), (3001, 257, 72)])   # last four: 256x128 3-stage kernel
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_nt_exact_integers(M, N, K, cdt):
    """small-integer operands: products and sums are exact in f32, so any lane-map / swizzle error shows bit-for-bit"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _ints((M, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    C = torch.full((M, N), 7.0, device="cuda", dtype=cdt)
    ops.gemm_nt_bf16(A, B, C, bias)
    want = (A.float() @ B.float().t() + bias).to(cdt)
    assert torch.equal(C, want)


def test_nt_padded_pitch_and_random():
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, K = 500, 333, 256
    Ab = torch.randn(M, K + 64, device="cuda", generator=g).to(torch.bfloat16)
    Bb = torch.randn(N, K + 8, device="cuda", generator=g).to(torch.bfloat16)
    Cb = torch.zeros(M, N + 19, device="cuda")
    ops.gemm_nt_bf16(Ab[:, :K], Bb[:, :K], Cb[:, :N])
    want = Ab[:, :K].double() @ Bb[:, :K].double().t()
    assert float((Cb[:, :N].double() - want).norm() / want.norm()) < 1e-6
    assert float(Cb[:, N:].abs().max()) == 0


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 128, 256), (4334, 1024, 4096), (200, 72, 640), (130, 1000, 128),
                                   (300, 130, 4100), (512, 512, 8192)])                                       # K >= 4096: 256x128 3-stage kernel
def test_tn_exact_integers(M, N, K):
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    lda, ldb = (M + 7) // 8 * 8 + 8, (N + 7) // 8 * 8
    Ab, Bb = _ints((K, lda), g), _ints((K, ldb), g)
    C = torch.zeros(M, N, device="cuda")
    ops.gemm_tn_bf16(Ab[:, :M], Bb[:, :N], C, accumulate=True)
    want = Ab[:, :M].float().t() @ Bb[:, :N].float()
    assert torch.equal(C, want)
    cs = torch.full((M,), 3.0, device="cuda")
    ops.gemm_tn_bf16(Ab[:, :M], Bb[:, :N], C, accumulate=True, colsum_a=cs)       # accumulates; fused column sums of A
    assert torch.equal(C, 2 * want)
    assert torch.equal(cs, 3.0 + Ab[:, :M].float().sum(0))


def test_rejects_unsupported_shapes():
    from ttmi import ops
    A = torch.zeros(64, 100, device="cuda", dtype=torch.bfloat16)     # K not a multiple of 64
    with pytest.raises(ValueError):
        ops.gemm_nt_bf16(A, A, torch.zeros(64, 64, device="cuda"))


def test_both_kernel_generations_agree():
    """ttmi_set_option(1, v): every kernel generation gives the identical (exact) result"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    A, B = _ints((4096, 512), g), _ints((640, 512), g)
    At, Bt = _ints((8192, 520), g), _ints((8192, 264), g)
    want_nt, want_tn = A.float() @ B.float().t(), At[:, :512].float().t() @ Bt[:, :260].float()
    for v in (1, 3, 4, 5, 8, 9):
        ops.set_option(1, v)
        C = torch.zeros(4096, 640, device="cuda")
        ops.gemm_nt_bf16(A, B, C)
        assert torch.equal(C, want_nt), v
        D = torch.zeros(512, 260, device="cuda")
        ops.gemm_tn_bf16(At[:, :512], Bt[:, :260], D, accumulate=True)
        assert torch.equal(D, want_tn), v
    ops.set_option(1, 4)


@pytest.mark.parametrize("M,N,K,pad", [(1024, 256, 64, 0), (1024, 256, 128, 0), (1024, 256, 192, 0), (1030, 700, 72, 3),
                                       (2048, 260, 320, 4), (5000, 4334, 1024, 18), (3000, 1024, 4352, 0), (1025, 257, 1000, 0),
                                       (70000, 1100, 512, 0)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_persistent_256_kernel_exact(M, N, K, pad, cdt):
    """v8 (persistent 256x256, staggered wave groups, counted vmcnt, LDS-transposed epilogue): exact on small integers for
    1..68 K-tiles, K tails, ragged M/N, padded output pitch (pad columns untouched), several rounds of tiles per CU"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _ints((M, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    Cfull = torch.full((M, N + pad), 5.0, device="cuda", dtype=cdt)
    ops.set_option(1, 8)
    try:
        ops.gemm_nt_bf16(A, B, Cfull[:, :N], bias)
    finally:
        ops.set_option(1, 4)
    assert torch.equal(Cfull[:, :N], (A.float() @ B.float().t() + bias).to(cdt))
    if pad:
        assert bool((Cfull[:, N:] == 5.0).all())


@pytest.mark.parametrize("M,N,K,pad", [(1024, 128, 64, 0), (1024, 256, 128, 0), (1024, 384, 192, 4), (16000, 512, 512, 0),
                                       (16000, 1536, 512, 0), (3001, 700, 320, 3), (16000, 2048, 2048, 0), (1025, 129, 1024, 0),
                                       (50000, 512, 256, 0), (3001, 700, 72, 3)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_persistent_256x128_kernel_exact(M, N, K, pad, cdt):
    """v9 (persistent 256x128, three LDS stages, one phase per K-tile): exact on small integers for 1..32 K-tiles, ragged M/N,
    padded pitch, 1..4 rounds of tiles per CU (K %% 64 != 0 falls back to the 128x128 kernel: same contract)"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _ints((M, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    Cfull = torch.full((M, N + pad), 5.0, device="cuda", dtype=cdt)
    ops.set_option(1, 9)
    try:
        ops.gemm_nt_bf16(A, B, Cfull[:, :N], bias)
    finally:
        ops.set_option(1, 4)
    assert torch.equal(Cfull[:, :N], (A.float() @ B.float().t() + bias).to(cdt))
    if pad:
        assert bool((Cfull[:, N:] == 5.0).all())


@pytest.mark.parametrize("reserve", [1, 4, 5, 13, 32, 100])
@pytest.mark.parametrize("gen,M,N,K", [(9, 16000, 512, 512), (9, 16000, 1024, 512), (9, 3001, 700, 320), (9, 1024, 128, 64), (8, 16000, 2048, 512), (8, 3001, 700, 320)])
def test_persistent_grids_of_any_size_exact(reserve, gen, M, N, K):
    """data-parallel backward leaves `reserve` CUs to the collective's kernels: the persistent NT kernels then run on EXACTLY 256 - reserve workgroups (round 6; rounded
    down to a multiple of 8 before, a reservation of 4 turned one round of the encoder's 252-tile GEMMs into two).  Any grid size, same bits: the XCDs' shares of a
    round differ by one workgroup"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K + reserve)
    A, B = _ints((M, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    want = (A.float() @ B.float().t() + bias).to(torch.bfloat16)
    C = torch.full((M, N), 5.0, device="cuda", dtype=torch.bfloat16)
    ops.set_option(1, gen)
    ops.reserve_cus(reserve)
    try:
        ops.gemm_nt_bf16(A, B, C, bias)
    finally:
        ops.reserve_cus(0)
        ops.set_option(1, 4)
    assert torch.equal(C, want)


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 32768), (1124, 1024, 40000), (2048, 1024, 33001), (4334, 1024, 65600),
                                   (1349, 2048, 34048), (1024, 1536, 32768)])
def test_persistent_256_wgrad_kernel_exact(M, N, K):
    """TN v8 (huge-reduction wgrad: persistent 256x256, transposed LDS reads, K-ranges per XCD, atomics, fused column sums via an
    all-ones MFMA) + the 128x128 kernel on the M % 256 strip: exact on small integers, K tail, with and without column sums"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    lda = (M + 7) // 8 * 8 + 8
    A, B = _ints((K, lda), g), _ints((K, N), g)
    want, wcs = A[:, :M].float().t() @ B.float(), A[:, :M].float().sum(0)
    C = torch.ones(M, N, device="cuda")
    cs = torch.full((M,), 3.0, device="cuda")
    ops.gemm_tn_bf16(A[:, :M], B, C, accumulate=True, colsum_a=cs)
    assert torch.equal(C, want + 1) and torch.equal(cs, wcs + 3)
    C.fill_(1.0)
    ops.gemm_tn_bf16(A[:, :M], B, C, accumulate=True)
    assert torch.equal(C, want + 1)


@pytest.mark.parametrize("M,N,K,cs", [(512, 512, 16000, False), (1536, 512, 16000, False), (2048, 512, 16000, True), (512, 2048, 16000, False),
                                      (256, 1024, 4096, False), (1024, 512, 8192, True), (512, 1024, 8192, True), (768, 2048, 6400, True)])
def test_persistent_256x128_wgrad_kernel_exact(M, N, K, cs):
    """TN v9 (encoder wgrads: persistent 256x128, transposed LDS reads, three stages, K-range items per XCD, atomics, all-ones-MFMA column
    sums from column tiles 0..3 when there are at least 4): exact on small integers, accumulating into a non-zero C"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A, B = _ints((K, M), g), _ints((K, N), g)
    want, wcs = A.float().t() @ B.float(), A.float().sum(0)
    C = torch.ones(M, N, device="cuda")
    csum = torch.full((M,), 3.0, device="cuda")
    ops.gemm_tn_bf16(A, B, C, accumulate=True, colsum_a=csum if cs else None)
    assert torch.equal(C, want + 1)
    if cs:
        assert torch.equal(csum, wcs + 3)
    ops.set_option(1, 5)                               # the 128x128 kernel gives the same numbers
    try:
        C2 = torch.ones(M, N, device="cuda")
        ops.gemm_tn_bf16(A, B, C2, accumulate=True)
    finally:
        ops.set_option(1, 4)
    assert torch.equal(C2, want + 1)


@pytest.mark.parametrize("M,N,K", [(300000, 512, 128), (350000, 384, 128)])
def test_streaming_store_path_exact(M, N, K):
    """bf16 outputs of 256 MB and more leave the persistent kernels (v8 / v9, lean bias epilogue) as streaming stores: exact on small
    integers, and the same bits with the streaming stores switched off (option 1 = 14)"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A, B = _ints((M, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    C = torch.full((M, N), 5.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt_bf16(A, B, C, bias)
    want = (A[:50000].float() @ B.float().t() + bias).to(torch.bfloat16)
    assert torch.equal(C[:50000], want) and torch.equal(C[-50000:], (A[-50000:].float() @ B.float().t() + bias).to(torch.bfloat16))
    ops.set_option(1, 14)
    try:
        C2 = torch.full((M, N), 5.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm_nt_bf16(A, B, C2, bias)
    finally:
        ops.set_option(1, 15)
    assert torch.equal(C2, C)


@pytest.mark.parametrize("M,N", [(4334, 1024), (1024, 1024), (6485, 2048)])       # C2 (strip inside the kernel), no strip, C4 (strip on the 128x128 kernel)
def test_tn_weighted_column_sums_exact(M, N):
    """persistent 256x256 TN kernel with column sums weighted per reduction row (the bias gradient of the exp-domain loss form): integer data,
    so every order of summation gives the same f32; M = 4334 has the ragged strip riding inside the kernel; 3 launches accumulate"""
    import ctypes
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(21)
    K = 65536 + 64 * 5
    lda = (M + 63) // 64 * 64                               # the joint's padded pitch (4352 for V = 4334)
    A = torch.randint(-1, 2, (K, lda), device="cuda", generator=g).to(torch.bfloat16)[:, :M]
    B = torch.randint(-1, 2, (K, N), device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randint(-2, 3, (K,), device="cuda", generator=g).to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda")
    col = torch.zeros(M, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for _ in range(3):
        ops.check(ops.lib().ttmi_gemm_tn_bf16_wsum(p(A), p(B), p(C), M, N, K, ctypes.c_long(lda), ctypes.c_long(N), ctypes.c_long(N), p(col), p(w),
                                                   ops._stream()), "ttmi_gemm_tn_bf16_wsum")
    want_c = 3 * (A.float().t() @ B.float())
    want_col = 3 * (A.float() * w.float()[:, None]).sum(0)
    assert torch.equal(col, want_col)
    assert torch.equal(C, want_c)


@pytest.mark.parametrize("M,N,K,gen", [(6400, 1536, 512, 9), (6400, 512, 512, 9), (6400, 512, 2048, 9), (6400, 2048, 512, 8), (3001, 700, 128, 9),
                                       (3001, 700, 128, 8), (6400, 2048, 512, 4), (1025, 129, 128, 9), (1000, 1536, 512, 4), (816, 2048, 512, 4), (77, 130, 72, 4)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_two_term_weight_one_launch_exact(M, N, K, gen, cdt):
    """NtEpilogue::B_lo (option 13): the persistent kernels walk A's K-tiles twice, the second time against the weight's second bf16 term -
    exact on small integers against A (B + B_lo)^T with bias and ReLU, on both generations and on the default route; rows beyond the problem untouched"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K + gen)
    A, B, Bl = _ints((M, K), g), _ints((N, K), g), _ints((N, K), g)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    for relu in (False, True):
        C = torch.full((M + 1, N), 5.0, device="cuda", dtype=cdt)
        ops.set_option(1, gen)
        try:
            ops.gemm_nt_bf16_two_term(A, B, Bl, C[:M], bias, relu)
        finally:
            ops.set_option(1, 4)
        want = A.float() @ (B.float() + Bl.float()).t() + bias
        if relu:
            want = want.clamp_min(0)
        assert torch.equal(C[:M], want.to(cdt))
        assert bool((C[M] == 5.0).all())


@pytest.mark.parametrize("M,N,Kh,gen", [(6400, 1536, 512, 9), (6400, 512, 2048, 8), (3001, 700, 128, 9), (3001, 700, 128, 8), (6400, 2048, 512, 4),
                                        (1000, 1536, 512, 4), (77, 130, 64, 4)])
def test_two_term_weight_second_walk_over_a_prefix_exact(M, N, Kh, gen):
    """NtEpilogue::K_lo (the bf16x3 products' layout): A = [A1 | A2] (K = 2 Kh), B = [B1 | B2] with pitch 3 Kh and B_lo = the third block of B's rows:
    C = A1 B1^T + A2 B2^T + A1 B_lo^T in one launch - exact on small integers on both persistent generations, the default route and the 128 x 128 kernel"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + Kh + gen)
    A, B3 = _ints((M, 2 * Kh), g), _ints((N, 3 * Kh), g)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    C = torch.full((M + 1, N), 5.0, device="cuda")
    ops.set_option(1, gen)
    try:
        ops.gemm_nt_bf16_two_term(A, B3[:, :2 * Kh], B3[:, 2 * Kh:], C[:M], bias, False, k_lo=Kh)
    finally:
        ops.set_option(1, 4)
    want = A.float() @ B3[:, :2 * Kh].float().t() + A[:, :Kh].float() @ B3[:, 2 * Kh:].float().t() + bias
    assert torch.equal(C[:M], want)
    assert bool((C[M] == 5.0).all())


def _exp_call(A, B, P, bias, rs, M, N, K, ld, shift=None):
    import ctypes
    from ttmi import ops
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    ops.check(ops.lib().ttmi_gemm_nt_bf16_exp(p(A), p(B), p(P), p(bias), p(rs), rs.shape[0], p(shift), M, N, K, ctypes.c_long(K), ctypes.c_long(K),
                                              ctypes.c_long(ld), ops._stream()), "ttmi_gemm_nt_bf16_exp")


@pytest.mark.parametrize("M,N,K,ld", [(5000, 4334, 1024, 4352), (33600, 1000, 256, 1024), (1041, 300, 128, 304), (2048, 256, 192, 256)])
def test_exp_store_instance_vs_float64(M, N, K, ld):
    """the exp-store instance of the persistent 256x256 kernel on its own (round 6; so far it was only exercised through the fused loss): C = bf16(exp(A.B^T + b - shift))
    with exact zeros in the pad columns [N, ld) and row-sum partials that add up to the row's sum of exponentials, against a float64 product: ragged M / N
    (the element-wise tail of the last 8-column group), 2 to 16 K-tiles, several rounds of tiles per CU"""
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).tanh().to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    shift = torch.full((1,), 0.25, device="cuda")
    nparts = 4 * ((N + 255) // 256)
    P = torch.full((M, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    rs = torch.zeros(nparts, M, device="cuda")
    ops.set_option(1, 8)                                  # the persistent kernel at every size it can run
    try:
        _exp_call(A, B, P, bias, rs, M, N, K, ld, shift)
    finally:
        ops.set_option(1, 4)
    assert bool((P[:, N:] == 0).all())
    rows = torch.randint(0, M, (512,), device="cuda", generator=g)
    rows[:2] = torch.tensor([0, M - 1], device="cuda")
    want = torch.exp(A[rows].double() @ B.double().t() + bias.double() - 0.25)
    got = P[rows, :N].double()
    assert float(((got - want).abs() / want).max()) < 4.2e-3                      # bf16 rounding of the stored value
    assert float(((rs.sum(0)[rows].double() - want.sum(1)).abs() / want.sum(1)).max()) < 1e-5
    # every 64-column partial on its own (part-major [part][row])
    for part in (0, nparts // 2, (N - 1) // 64):
        w = want[:, part * 64:min(part * 64 + 64, N)].sum(1)
        assert float(((rs[part, rows].double() - w).abs() / w.clamp_min(1e-30)).max()) < 1e-5, part
    if (N - 1) // 64 + 1 < nparts:
        assert bool((rs[(N - 1) // 64 + 1:] == 0).all())                          # strips entirely beyond N


@pytest.mark.parametrize("M,N,K", [(5000, 1024, 4352), (33600, 256, 1024), (1041, 1024, 128), (3000, 328, 256)])
def test_row_factor_instance_vs_float64(M, N, K):
    """the row-factor instance (joint dgrad of the exp-domain loss form) on its own through ttmi_gemm_nt_bf16_rowscale: dH = (A.B^T) * (1 - h^2) * s_r
    with h <- s_r h in place, against float64 (N = 328: the ragged last 8-column group goes element-wise)"""
    import ctypes
    from ttmi import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + 4)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    H0 = (torch.randn(M, N, device="cuda", generator=g)).tanh().to(torch.bfloat16)
    sr = torch.rand(M, device="cuda", generator=g) + 0.5
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    H = H0.clone()
    C = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.set_option(1, 8)
    try:
        ops.check(ops.lib().ttmi_gemm_nt_bf16_rowscale(p(A), p(B), p(C), p(H), p(sr), M, N, K, ctypes.c_long(K), ctypes.c_long(K), ctypes.c_long(N),
                                                       ops._stream()), "ttmi_gemm_nt_bf16_rowscale")
    finally:
        ops.set_option(1, 4)
    want = (A.double() @ B.double().t()) * (1 - H0.double() ** 2) * sr.double()[:, None]
    err = (C.double() - want).abs().max() / want.abs().max()
    assert float(err) < 5e-3
    assert torch.equal(H, (H0.float() * sr[:, None]).to(torch.bfloat16))
