"""CPU oracle for the Transformer-Transducer hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement (forward AND hand-derived backward) of the
reference's algorithm.  It is the checker for the HIP path; nothing in the
product package may import it.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use it.

Parity status
  * model half (encoders, joint): PINNED against the imported reference
    (tools/gen_golden.py ran /root/reference/tt on CPU in the build container;
    fixtures under tests/golden/, checked by tests/test_oracle_golden.py).
  * RNN-T loss half: the reference takes it from the un-vendored, un-pinned
    third-party package `warprnnt_pytorch` (requirements.txt:7), absent here.
    The restatement follows Graves-2012 / the warp-transducer call contract
    (train.py:53,231) and is pinned by the public warp-transducer known-answer
    vector, a float64 autograd lattice, brute-force alignment enumeration and
    finite differences (tests/test_oracle_rnnt.py).  The reference itself
    holds no test vector for it: "parity unpinned" at that boundary.

Reference lines followed (relative to /root/reference):
  layer_norm_*      torch.nn.LayerNorm as used at tt/transformer.py:52,76
  rel_attn_*        tt/transformer.py:106-177 (+ _rel_shift :82-89)
  ffn_*             tt/transformer.py:54-58
  layer_*           tt/transformer.py:192-197, tt/encoder.py:24-29
  encoder_*         tt/encoder.py:44-50
  decoder_*         tt/decoder.py:38-45
  joint_*           tt/model.py:20-39
  transducer_*      tt/model.py:58-68
  rnnt_*            contract at train.py:53 (external warprnnt_pytorch)
  greedy_decode     tt/model.py:70-108

All tensors are batch-major [B, L, ...] internally (the reference transposes to
[L, B, ...]; every op on the path is independent per batch element so the
layout is immaterial to the numbers).  dtype follows the inputs (float32 or
float64).  Dropout: the layer functions take optional multiplier arrays (0 or
1/(1-p)) under the keys drop_attn / drop_ff_in / drop_ff_out / drop_layer, so a
test can feed them the exact masks the HIP kernels drew; absent = no dropout.
"""
import numpy as np

LN_EPS = 1e-5


# ----------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------
def layer_norm_fwd(x, g, b, eps=LN_EPS):
    mean = x.mean(-1, keepdims=True)
    xc = x - mean
    var = (xc * xc).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = xc * rstd
    return xh * g + b, (xh, rstd)


def layer_norm_bwd(dy, cache, g):
    xh, rstd = cache
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xh * (dxh * xh).mean(-1, keepdims=True))
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xh).sum(red), dy.sum(red)


# ----------------------------------------------------------------------------
# masks (tt/utils.py:233-251) and the reference's mask-broadcast rules
# (tt/transformer.py:154-159)
# ----------------------------------------------------------------------------
def look_ahead_mask(L):
    return np.triu(np.ones((L, L), dtype=bool), 1)


def context_mask(L, left=10, right=2):
    up = np.triu(np.ones((L, L)), right + 1)
    down = np.tril(np.ones((L, L)), -left - 1)
    return (up + down) != 0


def chunk_mask(L, chunk, left):
    """Block-streaming mask: frame i (in block i//chunk) sees its own block and
    `left` frames before the block start.  No counterpart in the reference (it
    accepts any [L,L,1] mask tensor); used for BASELINE config 4."""
    i = np.arange(L)[:, None]
    j = np.arange(L)[None, :]
    lo = (i // chunk) * chunk - left
    hi = (i // chunk + 1) * chunk - 1
    return (j < lo) | (j > hi)


def normalize_mask(mask, B, L):
    """-> bool [B or 1, L(i), L(j)] or None.  A 2-D mask is (klen, bsz) and is
    broadcast over the query index; a 3-D mask is (qlen, klen, bsz|1)."""
    if mask is None:
        return None
    m = np.asarray(mask) != 0
    if m.ndim == 2:                      # [j, b] -> [b, 1, j]
        return np.ascontiguousarray(m.T[:, None, :])
    if m.ndim == 3:                      # [i, j, b|1] -> [b|1, i, j]
        return np.ascontiguousarray(np.transpose(m, (2, 0, 1)))
    raise ValueError("mask must be 2-D or 3-D")


# ----------------------------------------------------------------------------
# relative-position attention
# ----------------------------------------------------------------------------
def rel_table_index(L, K):
    """e(p) = max(0, p + K - L): which row of the K-row learnable table serves
    effective position p in [0, L).  L <= K: the last L rows (:133-135);
    L > K: row 0 repeated L-K times in front (:128-132)."""
    return np.maximum(0, np.arange(L) + K - L)


def rel_shift_index(L):
    """Closed form of _rel_shift (:82-89) applied to X[i, p]:
         BD[i, j] = X[i,   L-1-i+j]   j <= i
                  = 0                 j == i+1
                  = X[i+1, j-i-2]     j >= i+2
    Returns (row, col, valid) index arrays of shape [L, L]."""
    i = np.arange(L)[:, None]
    j = np.arange(L)[None, :]
    low = j <= i
    up = j >= i + 2
    row = np.where(low, i, np.minimum(i + 1, L - 1))
    col = np.where(low, L - 1 - i + j, np.maximum(j - i - 2, 0))
    return row, col, (low | up)


def rel_shift_flat(G):
    """The same shift as rel_shift_index, as ONE reinterpretation of memory: prepend a zero column to
    G [.., L, L] -> [.., L, L+1]; the flat buffer read from offset L with pitch L is BD.  (This is what the
    reference's pad/view/drop computes and what the HIP path does with GEMM pitches.)  Equality with the
    index form is asserted in tests/test_oracle_golden.py."""
    L = G.shape[-1]
    lead = G.shape[:-2]
    Gp = np.concatenate([np.zeros(lead + (L, 1), dtype=G.dtype), G], -1).reshape(lead + (L * (L + 1),))
    return Gp[..., L:].reshape(lead + (L, L))


def rel_shift_flat_bwd(dBD):
    """adjoint of rel_shift_flat: dG [.., L, L] from dBD [.., L, L]"""
    L = dBD.shape[-1]
    lead = dBD.shape[:-2]
    flat = np.concatenate([np.zeros(lead + (L,), dtype=dBD.dtype), dBD.reshape(lead + (L * L,))], -1)
    return flat.reshape(lead + (L, L + 1))[..., 1:]


def rel_attn_fwd(w, p, mask=None):
    """w [B,L,d]; p: dict with qkv_w [3HD,d], o_w [d,HD], ln_g, ln_b,
    r_emb [K,H,D], r_w_bias [H,D], r_bias [K,H]."""
    B, L, d = w.shape
    K, H, D = p["r_emb"].shape
    heads = w @ p["qkv_w"].T                                  # :115
    q, k, v = [heads[..., s * H * D:(s + 1) * H * D].reshape(B, L, H, D) for s in range(3)]
    e = rel_table_index(L, K)
    E = p["r_emb"][e]                                         # [L,H,D]
    c = p["r_bias"][e]                                        # [L,H]
    qh, kh, vh = (t.transpose(0, 2, 1, 3) for t in (q, k, v))            # [B,H,L,D]
    AC = (qh + p["r_w_bias"][None, :, None, :]) @ kh.transpose(0, 1, 3, 2)   # :140-142
    G = qh @ E.transpose(1, 2, 0)[None] + c.T[None, :, None, :]            # :143-144  [B,H,L,L(p)]
    BD = rel_shift_flat(G)                                    # :145
    scale = 1.0 / np.sqrt(D)
    S = (AC + BD) * scale                                     # :148-149
    m = normalize_mask(mask, B, L)
    if m is not None:
        S = np.where(m[:, None], -np.inf, S)                  # :154-159
    Smax = S.max(-1, keepdims=True)
    Pn = np.exp(S - Smax)
    P = Pn / Pn.sum(-1, keepdims=True)                        # :164
    O = (P @ vh).transpose(0, 2, 1, 3).reshape(B, L, H * D)   # :167-170
    a = O @ p["o_w"].T                                        # :172
    dm = p.get("drop_attn")                                   # optional dropout multipliers (0 or 1/(1-p)), [B,L,d]; :173
    if dm is not None:
        a = a * dm
    y, lnc = layer_norm_fwd(w + a, p["ln_g"], p["ln_b"])      # :175
    cache = dict(w=w, q=q, k=k, v=v, E=E, e=e, P=P, O=O, lnc=lnc, m=m)
    return y, cache


def rel_attn_bwd(dy, cache, p):
    w, q, k, v, E, e, P, O = (cache[n] for n in ("w", "q", "k", "v", "E", "e", "P", "O"))
    B, L, d = w.shape
    K, H, D = p["r_emb"].shape
    scale = 1.0 / np.sqrt(D)
    g = {}
    dres, g["ln_g"], g["ln_b"] = layer_norm_bwd(dy, cache["lnc"], p["ln_g"])
    dw = dres.copy()
    da = dres if p.get("drop_attn") is None else dres * p["drop_attn"]
    g["o_w"] = da.reshape(-1, d).T @ O.reshape(-1, H * D)
    dO = (da @ p["o_w"]).reshape(B, L, H, D).transpose(0, 2, 1, 3)      # [B,H,L,D]
    qh, kh, vh = (t.transpose(0, 2, 1, 3) for t in (q, k, v))
    dP = dO @ vh.transpose(0, 1, 3, 2)
    dv = (P.transpose(0, 1, 3, 2) @ dO).transpose(0, 2, 1, 3)
    dS = P * (dP - (dP * P).sum(-1, keepdims=True)) * scale   # zero where masked (P = 0)
    dqh = dS @ kh                                             # AC wrt (q+u)
    g["r_w_bias"] = dqh.sum((0, 2))
    dk = (dS.transpose(0, 1, 3, 2) @ (qh + p["r_w_bias"][None, :, None, :])).transpose(0, 2, 1, 3)
    dG = rel_shift_flat_bwd(dS)
    dqh = dqh + dG @ E.transpose(1, 0, 2)[None]               # [B,H,L,L(p)] @ [H,L(p),D]
    dq = dqh.transpose(0, 2, 1, 3)
    dE = (dG.transpose(0, 1, 3, 2) @ qh).sum(0).transpose(1, 0, 2)        # [L(p),H,D]
    dc = dG.sum((0, 2)).T                                     # [L(p), H]
    g["r_emb"] = np.zeros_like(p["r_emb"])
    g["r_bias"] = np.zeros_like(p["r_bias"])
    np.add.at(g["r_emb"], e, dE)
    np.add.at(g["r_bias"], e, dc)
    dheads = np.concatenate([dq.reshape(B, L, -1), dk.reshape(B, L, -1), dv.reshape(B, L, -1)], -1)
    g["qkv_w"] = dheads.reshape(-1, 3 * H * D).T @ w.reshape(-1, d)
    dw = dw + dheads @ p["qkv_w"]
    return dw, g


# ----------------------------------------------------------------------------
# position-wise FFN: y = LN(x + W2 relu(W1 LN(x) + b1) + b2), ONE LayerNorm
# module used twice (tt/transformer.py:52,55-56)
# ----------------------------------------------------------------------------
def ffn_fwd(x, p):
    h, c1 = layer_norm_fwd(x, p["ff_ln_g"], p["ff_ln_b"])
    z1 = h @ p["ff_w1"].T + p["ff_b1"]
    # ReLU is discontinuous in its derivative: a unit whose pre-activation is within rounding noise of 0 may be on in one arithmetic and
    # off in another, which changes gradients by a whole rank-1 term.  Like the dropout multipliers below, a test may therefore feed the
    # ON/OFF decisions the checked implementation actually took ("relu_active", bool [B,L,Di]); absent = decide here (tt/transformer.py:45).
    active = p.get("relu_active")
    a = np.maximum(z1, 0) if active is None else np.where(active, z1, 0)
    relu_on = (a > 0) if active is None else active
    one = np.ones((), dtype=x.dtype)                          # optional dropout multipliers: CoreNet.2, CoreNet.4, layer (:47,49,196)
    m_in, m_out, m_layer = (p.get(k, one) if p.get(k) is not None else one for k in ("drop_ff_in", "drop_ff_out", "drop_layer"))
    a = a * m_in
    f = (a @ p["ff_w2"].T + p["ff_b2"]) * m_out
    y, c2 = layer_norm_fwd(x + f, p["ff_ln_g"], p["ff_ln_b"])
    return y * m_layer, dict(x=x, h=h, a=a, c1=c1, c2=c2, m_in=m_in, m_out=m_out, m_layer=m_layer, relu_on=relu_on, z1_abs_min=float(np.abs(z1).min()))


def ffn_bwd(dy, cache, p):
    x, h, a = cache["x"], cache["h"], cache["a"]
    d = x.shape[-1]
    g = {}
    dres, dg2, db2 = layer_norm_bwd(dy * cache["m_layer"], cache["c2"], p["ff_ln_g"])
    dx = dres.copy()
    df = dres * cache["m_out"]
    g["ff_b2"] = df.reshape(-1, d).sum(0)
    g["ff_w2"] = df.reshape(-1, d).T @ a.reshape(-1, a.shape[-1])
    da = (df @ p["ff_w2"]) * cache["m_in"] * cache["relu_on"]
    g["ff_b1"] = da.reshape(-1, a.shape[-1]).sum(0)
    g["ff_w1"] = da.reshape(-1, a.shape[-1]).T @ h.reshape(-1, d)
    dh = da @ p["ff_w1"]
    dx1, dg1, db1 = layer_norm_bwd(dh, cache["c1"], p["ff_ln_g"])
    g["ff_ln_g"] = dg1 + dg2
    g["ff_ln_b"] = db1 + db2
    return dx + dx1, g


def layer_fwd(x, p, mask=None):
    y, ca = rel_attn_fwd(x, p, mask)
    z, cf = ffn_fwd(y, p)
    return z, (ca, cf)


def layer_bwd(dz, cache, p):
    ca, cf = cache
    dy, gf = ffn_bwd(dz, cf, p)
    dx, ga = rel_attn_bwd(dy, ca, p)
    ga.update(gf)
    return dx, ga


# ----------------------------------------------------------------------------
# state_dict <-> oracle parameter names
# ----------------------------------------------------------------------------
_LAYER_KEYS = {
    "r_emb": "r_emb", "r_w_bias": "r_w_bias", "r_bias": "r_bias",
    "qkv_w": "MultiHeadAttention.dec_attn.qkv_net.weight",
    "o_w": "MultiHeadAttention.dec_attn.o_net.weight",
    "ln_g": "MultiHeadAttention.dec_attn.layer_norm.weight",
    "ln_b": "MultiHeadAttention.dec_attn.layer_norm.bias",
    "ff_w1": "MultiHeadAttention.pos_ff.CoreNet.0.weight",
    "ff_b1": "MultiHeadAttention.pos_ff.CoreNet.0.bias",
    "ff_w2": "MultiHeadAttention.pos_ff.CoreNet.3.weight",
    "ff_b2": "MultiHeadAttention.pos_ff.CoreNet.3.bias",
    "ff_ln_g": "MultiHeadAttention.pos_ff.layer_norm.weight",
    "ff_ln_b": "MultiHeadAttention.pos_ff.layer_norm.bias",
}


_LAYER_EXTRAS = ("relu_active", "drop_attn", "drop_ff_in", "drop_ff_out", "drop_layer")     # optional per-layer test inputs, same key scheme


def layer_params(sd, prefix, i):
    p = {k: sd["%slayers.%d.%s" % (prefix, i, v)] for k, v in _LAYER_KEYS.items()}
    for k in _LAYER_EXTRAS:
        v = sd.get("%slayers.%d.%s" % (prefix, i, k))
        if v is not None:
            p[k] = v
    return p


def n_layers(sd, prefix):
    n = 0
    while "%slayers.%d.r_emb" % (prefix, n) in sd:
        n += 1
    return n


def layer_grads_to_sd(g, prefix, i, out):
    for k, v in _LAYER_KEYS.items():
        out["%slayers.%d.%s" % (prefix, i, v)] = g[k]


# ----------------------------------------------------------------------------
# stacks, joint, full model.  `sd` is a flat dict keyed like the reference's
# state_dicts with prefixes "encoder." / "decoder." / "joint.".
# ----------------------------------------------------------------------------
def stack_fwd(x, sd, prefix, mask=None):
    caches = []
    for i in range(n_layers(sd, prefix)):
        x, c = layer_fwd(x, layer_params(sd, prefix, i), mask)
        caches.append(c)
    return x, caches


def stack_bwd(dx, caches, sd, prefix, grads):
    for i in reversed(range(len(caches))):
        dx, g = layer_bwd(dx, caches[i], layer_params(sd, prefix, i))
        layer_grads_to_sd(g, prefix, i, grads)
    return dx


def encoder_fwd(x, sd, mask=None):
    return stack_fwd(x, sd, "encoder.", mask)


def decoder_fwd(tokens, sd, mask=None):
    emb = sd["decoder.dec_embedding.weight"][tokens]          # tt/decoder.py:39
    y, caches = stack_fwd(emb, sd, "decoder.", mask)
    return y, (tokens, caches)


def decoder_bwd(dy, cache, sd, grads):
    tokens, caches = cache
    demb = stack_bwd(dy, caches, sd, "decoder.", grads)
    gw = np.zeros_like(sd["decoder.dec_embedding.weight"])
    np.add.at(gw, tokens.reshape(-1), demb.reshape(-1, demb.shape[-1]))
    gw[0] = 0                                                 # padding_idx=0 (:26)
    grads["decoder.dec_embedding.weight"] = gw


def joint_fwd(enc, dec, sd):
    """z[b,t,u,:] = Wp tanh(We enc[b,t] + Wd dec[b,u] + bf) + bp with
    forward_layer.weight = [We | Wd] (encoder half first: cat((enc,dec)), :33)."""
    Wf, bf = sd["joint.forward_layer.weight"], sd["joint.forward_layer.bias"]
    Wp, bp = sd["joint.project_layer.weight"], sd["joint.project_layer.bias"]
    de = enc.shape[-1]
    pe = enc @ Wf[:, :de].T
    pd = dec @ Wf[:, de:].T
    if enc.ndim == 1:                                         # decode path (:30-33)
        h = np.tanh(pe + pd + bf)
        return h @ Wp.T + bp, None
    h = np.tanh(pe[:, :, None, :] + pd[:, None, :, :] + bf)
    z = (h.reshape(-1, h.shape[-1]) @ Wp.T + bp).reshape(h.shape[:-1] + (Wp.shape[0],))
    return z, dict(enc=enc, dec=dec, h=h)


def joint_bwd(dz, cache, sd, grads):
    Wf = sd["joint.forward_layer.weight"]
    Wp = sd["joint.project_layer.weight"]
    enc, dec, h = cache["enc"], cache["dec"], cache["h"]
    de = enc.shape[-1]
    J = h.shape[-1]
    V = dz.shape[-1]
    grads["joint.project_layer.bias"] = dz.reshape(-1, V).sum(0)
    grads["joint.project_layer.weight"] = dz.reshape(-1, V).T @ h.reshape(-1, J)
    dpre = (dz.reshape(-1, V) @ Wp).reshape(h.shape) * (1 - h * h)
    dpe = dpre.sum(2)                                         # [B,T,J]
    dpd = dpre.sum(1)                                         # [B,U1,J]
    grads["joint.forward_layer.bias"] = dpe.reshape(-1, J).sum(0)
    gWe = dpe.reshape(-1, J).T @ enc.reshape(-1, de)
    gWd = dpd.reshape(-1, J).T @ dec.reshape(-1, dec.shape[-1])
    grads["joint.forward_layer.weight"] = np.concatenate([gWe, gWd], 1)
    return dpe @ Wf[:, :de], dpd @ Wf[:, de:]


def transducer_fwd(inputs, targets, sd, audio_mask=None):
    """tt/model.py:58-68.  targets [B,U] int; returns logits [B,T,U+1,V]."""
    B, U = targets.shape
    tg = np.concatenate([np.zeros((B, 1), dtype=targets.dtype), targets], 1)   # :59
    enc, ce = encoder_fwd(inputs, sd, audio_mask)                              # :63
    lm = look_ahead_mask(U + 1)[:, :, None]                                    # :62
    dec, cd = decoder_fwd(tg, sd, lm)                                          # :64
    z, cj = joint_fwd(enc, dec, sd)                                            # :66
    return z, (ce, cd, cj)


def transducer_bwd(dz, cache, sd):
    ce, cd, cj = cache
    grads = {}
    denc, ddec = joint_bwd(dz, cj, sd, grads)
    decoder_bwd(ddec, cd, sd, grads)
    dinp = stack_bwd(denc, ce, sd, "encoder.", grads)
    return grads, dinp


# ----------------------------------------------------------------------------
# RNN-T loss (contract: train.py:53,231; blank = 0, reduction = 'mean')
# ----------------------------------------------------------------------------
def _logaddexp(a, b):
    return np.logaddexp(a, b)


def rnnt_lattice(lp_blank, lp_label):
    """lp_blank [T,U1], lp_label [T,U1] (column U1-1 of lp_label unused).
    Returns alpha, beta [T,U1] and ll = log P(y|x)."""
    T, U1 = lp_blank.shape
    NEG = -np.inf
    alpha = np.full((T, U1), NEG, dtype=lp_blank.dtype)
    beta = np.full((T, U1), NEG, dtype=lp_blank.dtype)
    alpha[0, 0] = 0
    for t in range(T):
        for u in range(U1):
            if t == 0 and u == 0:
                continue
            a = alpha[t - 1, u] + lp_blank[t - 1, u] if t > 0 else NEG
            b = alpha[t, u - 1] + lp_label[t, u - 1] if u > 0 else NEG
            alpha[t, u] = _logaddexp(a, b)
    beta[T - 1, U1 - 1] = lp_blank[T - 1, U1 - 1]
    for t in range(T - 1, -1, -1):
        for u in range(U1 - 1, -1, -1):
            if t == T - 1 and u == U1 - 1:
                continue
            a = beta[t + 1, u] + lp_blank[t, u] if t < T - 1 else NEG
            b = beta[t, u + 1] + lp_label[t, u] if u < U1 - 1 else NEG
            beta[t, u] = _logaddexp(a, b)
    return alpha, beta, beta[0, 0]


def rnnt_lattice_diag(lp_blank, lp_label):
    """Same recursion, vectorised over anti-diagonals (used for larger cases)."""
    T, U1 = lp_blank.shape
    NEG = -np.inf
    alpha = np.full((T + 1, U1 + 1), NEG, dtype=lp_blank.dtype)   # 1-padded at the low side
    A = alpha[1:, 1:]
    A[0, 0] = 0
    for dg in range(1, T + U1 - 1):
        u = np.arange(max(0, dg - T + 1), min(U1 - 1, dg) + 1)
        t = dg - u
        a = np.where(t > 0, alpha[t, u + 1] + lp_blank[np.maximum(t - 1, 0), u], NEG)
        b = np.where(u > 0, alpha[t + 1, u] + lp_label[t, np.maximum(u - 1, 0)], NEG)
        A[t, u] = np.logaddexp(a, b)
    beta = np.full((T + 1, U1 + 1), NEG, dtype=lp_blank.dtype)    # 1-padded at the high side
    beta[T - 1, U1 - 1] = lp_blank[T - 1, U1 - 1]
    for dg in range(T + U1 - 3, -1, -1):
        u = np.arange(max(0, dg - T + 1), min(U1 - 1, dg) + 1)
        t = dg - u
        a = beta[t + 1, u] + lp_blank[t, u]
        b = beta[t, u + 1] + lp_label[t, u]
        beta[t, u] = np.logaddexp(a, b)
    return np.ascontiguousarray(A), np.ascontiguousarray(beta[:T, :U1]), beta[0, 0]


def rnnt_loss(logits, labels, act_lens, label_lens, blank=0, reduction="mean", lattice=None):
    """logits [B,T,U1,V] (un-normalised), labels [B,U] int, act_lens [B],
    label_lens [B].  Returns (loss, costs[B], grad wrt logits [B,T,U1,V]) where
    grad already includes the 1/B of reduction='mean'."""
    lattice = lattice or rnnt_lattice_diag
    B, T, U1, V = logits.shape
    costs = np.zeros(B, dtype=logits.dtype)
    grad = np.zeros_like(logits)
    for b in range(B):
        Tb, Ub = int(act_lens[b]), int(label_lens[b])
        x = logits[b, :Tb, :Ub + 1]
        mx = x.max(-1, keepdims=True)
        lse = mx[..., 0] + np.log(np.exp(x - mx).sum(-1))
        lp = x - lse[..., None]
        y = np.asarray(labels[b, :Ub]).astype(np.int64)
        lpb = lp[..., blank]
        lpl = np.zeros_like(lpb)
        if Ub > 0:
            lpl[:, :Ub] = np.take_along_axis(lp[:, :Ub], np.broadcast_to(y[None, :, None], (Tb, Ub, 1)), -1)[..., 0]
        alpha, beta, ll = lattice(lpb, lpl)
        costs[b] = -ll
        ab = alpha + beta - ll
        g = np.exp(ab[..., None] + lp)
        # blank emissions
        bnext = np.full_like(beta, -np.inf)
        bnext[:-1] = beta[1:]
        bnext[Tb - 1, Ub] = 0.0
        g[..., blank] -= np.exp(alpha + lpb + bnext - ll)
        # label emissions
        if Ub > 0:
            t_idx = np.arange(Tb)[:, None]
            u_idx = np.arange(Ub)[None, :]
            sub = np.exp(alpha[:, :Ub] + lpl[:, :Ub] + beta[:, 1:] - ll)
            np.subtract.at(g, (t_idx, u_idx, y[None, :]), sub)
        grad[b, :Tb, :Ub + 1] = g
    if reduction == "mean":
        return costs.sum() / B, costs, grad / B
    if reduction == "sum":
        return costs.sum(), costs, grad
    return costs, costs, grad


def rnnt_brute_force(logits, labels, blank=0):
    """-log sum over ALL alignments, by explicit enumeration (tiny T,U only)."""
    T, U1, V = logits.shape
    U = U1 - 1
    lp = logits - np.log(np.exp(logits).sum(-1, keepdims=True))
    total = []

    def walk(t, u, acc):
        if t == T - 1 and u == U:
            total.append(acc + lp[t, u, blank])
            return
        if t < T - 1:
            walk(t + 1, u, acc + lp[t, u, blank])
        if u < U:
            walk(t, u + 1, acc + lp[t, u, labels[u]])

    walk(0, 0, 0.0)
    return -np.log(np.exp(np.array(total)).sum())


# ----------------------------------------------------------------------------
# whole training-step maths: logits -> loss -> every gradient
# ----------------------------------------------------------------------------
def transducer_loss_and_grads(inputs, targets, act_lens, label_lens, sd, audio_mask=None, lattice=None):
    z, cache = transducer_fwd(inputs, targets, sd, audio_mask)
    loss, costs, dz = rnnt_loss(z, targets, act_lens, label_lens, lattice=lattice)
    grads, dinp = transducer_bwd(dz, cache, sd)
    return dict(logits=z, loss=loss, costs=costs, dlogits=dz, grads=grads, dinputs=dinp)


# ----------------------------------------------------------------------------
# greedy decode (tt/model.py:70-108): <=1 symbol per frame, label encoder rerun
# on the full history WITHOUT look-ahead mask, start token 0.
# ----------------------------------------------------------------------------
def greedy_decode(enc_state, length, sd):
    tokens = [0]
    dec, _ = decoder_fwd(np.array([tokens]), sd, None)
    dstate = dec[0, -1]
    for t in range(int(length)):
        z, _ = joint_fwd(enc_state[t], dstate, sd)
        pred = int(np.argmax(z))
        if pred != 0:
            tokens.append(pred)
            dec, _ = decoder_fwd(np.array([tokens]), sd, None)
            dstate = dec[0, -1]
    return tokens[1:]


def beam_search(enc_state, length, sd, beam_width=5):
    """tt/model.py:110-179 with its quirks: the frame loop follows the currently most probable hypothesis (a frame on which it predicts blank
    expands nothing); on an expanding frame every hypothesis contributes its top `beam_width` non-blank symbols (top beam_width + 1, the blank
    or else the last one dropped); the child token lists are appended to and never re-seeded from their parents; the first expansion fills
    the children column-wise with the log-probabilities of the last hypothesis scored."""
    import copy
    import heapq

    def posterior(tokens, t):
        dec, _ = decoder_fwd(np.array([tokens]), sd, None)
        z, _ = joint_fwd(enc_state[t], dec[0, -1], sd)
        e = np.exp(z - z.max())
        return e / e.sum()

    def topk(p, k):
        idx = np.argsort(-p, kind="stable")[:k]                 # torch.topk returns descending values; ties are not expected on real logits
        return p[idx].tolist(), idx.tolist()

    hyps = [[0] for _ in range(beam_width)]
    score = np.zeros((beam_width,), dtype=float)
    child = [[[0] for _ in range(beam_width)] for _ in range(beam_width)]
    child_score = np.zeros((beam_width, beam_width), dtype=float)
    first = True
    for t in range(int(length)):
        lead = int(score.argmax())
        if int(np.argmax(posterior(hyps[lead], t))) == 0:
            continue
        for k in range(beam_width):
            values, indices = topk(posterior(hyps[k], t), beam_width + 1)
            drop = indices.index(0) if 0 in indices else len(indices) - 1
            indices.pop(drop)
            values.pop(drop)
            for i, tok in enumerate(indices):
                if first:
                    child[i][k].append(tok)
                else:
                    child[k][i].append(tok)
            if first:
                child_score[:, k] = np.log(values)
            else:
                child_score[k] = score[k] + np.log(values)
        if first:
            first = False
            for i in range(beam_width):
                hyps[i] = copy.deepcopy(child[i][0])
                score[i] = child_score[i, 0]
        else:
            best = heapq.nlargest(beam_width, range(beam_width ** 2), child_score.take)
            for i, idx in enumerate(best):
                score[i] = child_score[idx // beam_width, idx % beam_width]
                hyps[i] = copy.deepcopy(child[idx // beam_width][idx % beam_width])
    return hyps[int(score.argmax())][1:]


def recognize_beam_search(inputs, lengths, sd, audio_mask=None):
    enc, _ = encoder_fwd(inputs, sd, audio_mask)
    return [beam_search(enc[b], lengths[b], sd) for b in range(inputs.shape[0])]


def recognize(inputs, lengths, sd, audio_mask=None):
    enc, _ = encoder_fwd(inputs, sd, audio_mask)
    return [greedy_decode(enc[b], lengths[b], sd) for b in range(inputs.shape[0])]
