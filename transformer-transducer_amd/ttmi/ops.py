"""Thin typed wrappers: torch tensors -> raw device pointers -> libttmi C ABI.

torch is plumbing only (device memory, current stream).  Every function here
launches hand-written HIP kernels; none has a PyTorch/CPU fallback.
"""
import ctypes
import os
import sys

import torch

from . import check, lib

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p


_TRACE = bool(os.environ.get("TTMI_TRACE_CALLS"))       # debugging: name and shape of every layer-level call on stderr before it is issued


def _trace(name, *dims):
    if _TRACE:
        print("[ttmi] %s %s stream %x" % (name, dims, torch.cuda.current_stream().cuda_stream), file=sys.stderr, flush=True)


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def weights_fresh():
    """called at the module-level entry points (BuildEncoder / BuildDecoder / JointNet forward, Transducer.loss) before weights are handed
    to the library: if a ttmi.train.FlatModel keeps bf16 weight shadows and a parameter was changed through torch since their last
    refresh, they are rebuilt first.  Free when no shadows exist."""
    import sys
    tr = sys.modules.get("ttmi.train")
    if tr is not None and tr._shadowed:
        tr.ensure_fresh()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("ttmi ops need device tensors (no CPU fallback); got a %s tensor" % t.device)


# ----------------------------------------------------------------------------- row-padded views
def row_pitch(t):
    """t [..., V] -> pitch (elements between consecutive rows) if t is a dense tensor or a [..., :V] view of a dense
    [..., pitch] buffer (what the bf16 joint produces: pitch = V rounded up to 64); else None."""
    if t.dim() < 2 or t.stride(-1) != 1:
        return None
    ld = t.stride(-2)
    if ld < t.shape[-1]:
        return None
    for i in range(t.dim() - 2):
        if t.shape[i] > 1 and t.stride(i) != t.stride(i + 1) * t.shape[i + 1]:
            return None
    return ld


def padded_empty(shape, dtype, device, multiple=64):
    """dense [..., roundup(V, multiple)] buffer and its [..., :V] view"""
    V = shape[-1]
    Vp = (V + multiple - 1) // multiple * multiple
    buf = torch.empty(*shape[:-1], Vp, dtype=dtype, device=device)
    return buf, buf[..., :V]


# gradients written by rnnt_loss_bwd carry exact zeros in their pad columns; the joint's fast dgrad relies on that.  The guarantee is
# attached to the gradient's TENSOR OBJECT (attribute `_ttmi_zero_pad` = its pitch), never to a raw address: a foreign gradient that the
# caching allocator happens to place where an earlier one lived must not inherit it.


# ----------------------------------------------------------------------------- RNN-T loss
def rnnt_workspace(B, T, U1, device):
    n = lib().ttmi_rnnt_workspace_bytes(c_int(B), c_int(T), c_int(U1))
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=device)


def rnnt_loss_fwd(logits, labels, act_lens, label_lens, blank, workspace):
    """logits: f32/bf16 [B,T,U1,V], dense or row-padded view"""
    _need_cuda(logits, labels, act_lens, label_lens, workspace)
    B, T, U1, V = logits.shape
    ld = row_pitch(logits)
    assert ld is not None
    costs = torch.empty(B, dtype=torch.float32, device=logits.device)
    check(lib().ttmi_rnnt_loss_fwd(_p(logits), c_int(_DT[logits.dtype]), c_long(ld), _p(labels), _p(act_lens), _p(label_lens),
                                   c_int(B), c_int(T), c_int(U1), c_int(V), c_int(blank), _p(workspace), _p(costs), _stream()),
          "ttmi_rnnt_loss_fwd")
    return costs


def rnnt_loss_bwd(logits, labels, act_lens, label_lens, blank, workspace, grad_out, grad_out_stride, scale, inplace=False):
    """inplace: the gradient overwrites the logits (same dtype and pitch; the fused joint + loss path needs them only once)"""
    _need_cuda(logits, labels, act_lens, label_lens, workspace, grad_out)
    B, T, U1, V = logits.shape
    ld = row_pitch(logits)
    if inplace:
        buf = grad = logits
        ldg = ld
    elif ld == V:
        buf = grad = torch.empty_like(logits)
        ldg = V
    else:
        buf = torch.empty(B, T, U1, ld, dtype=logits.dtype, device=logits.device)
        grad, ldg = buf[..., :V], ld
    check(lib().ttmi_rnnt_loss_bwd(_p(logits), c_int(_DT[logits.dtype]), c_long(ld), _p(labels), _p(act_lens), _p(label_lens),
                                   c_int(B), c_int(T), c_int(U1), c_int(V), c_int(blank), _p(workspace), _p(grad_out),
                                   c_int(grad_out_stride), c_float(scale), _p(grad), c_long(ldg), _stream()),
          "ttmi_rnnt_loss_bwd")
    if ldg != V:
        grad._ttmi_zero_pad = ldg
    return grad


# ----------------------------------------------------------------------------- generic GEMM (tests / bring-up)
GEMM_BIAS, GEMM_RELU, GEMM_ATOMIC, GEMM_MASK_AUX, GEMM_A_KMAJOR, GEMM_B_KMAJOR, GEMM_BF16_MFMA, GEMM_BF16X3 = 1, 2, 4, 8, 16, 32, 64, 128
_DT = {torch.float32: 0, torch.bfloat16: 1}


def gemm(A, B, C, M, N, K, lda, ldb, ldc, flags, bias=None, aux=None, alpha=1.0, beta=0.0, nz1=1, nz2=1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), splitk=1):
    _need_cuda(A, B, C, bias, aux)
    check(lib().ttmi_gemm(_p(A), _p(B), _p(C), _p(bias), _p(aux), c_int(_DT[A.dtype]), c_int(_DT[B.dtype]),
                          c_int(_DT[C.dtype]), c_int(M), c_int(N), c_int(K), c_long(lda), c_long(ldb), c_long(ldc),
                          c_int(nz1), c_int(nz2), c_long(sA[0]), c_long(sA[1]), c_long(sB[0]), c_long(sB[1]),
                          c_long(sC[0]), c_long(sC[1]), c_float(alpha), c_float(beta), c_int(flags), c_int(splitk),
                          _stream()), "ttmi_gemm")
    return C


# ----------------------------------------------------------------------------- grouped weight gradients
class WgradDesc(ctypes.Structure):
    """ttmi_wgrad_desc (include/ttmi.h)"""
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("colsum", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int),
                ("lda", c_long), ("ldb", c_long), ("ldc", c_long)]


class _WgradLane:
    __slots__ = ("stream", "descs", "alive", "after")

    def __init__(self, stream):
        self.stream, self.descs, self.alive, self.after = stream, [], [], []


class WgradQueue:
    """Weight-gradient GEMMs of audio-sized encoder layers, held back until `group` layers' worth (4 problems each) are waiting and
    then launched together: 4 layers x 64 tiles = one 256 x 128 tile per CU over the whole reduction, no atomics (ttmi_wgrad_group).
    Holds the operand buffers alive until then; `after` callbacks (gradient-ready hooks of the deferred parameters) run after the launch.
    One lane per stream: a layer's problems are launched on the stream that produced their operands (the label encoder's backward pass
    runs on a side stream beside the audio encoder's; their entries never share a launch)."""

    def __init__(self, group=4, immediate_first_layer=False):
        self.limit = 4 * group
        # data-parallel runs: the first layer's gradients are the last of the step - whatever is reduced after them overlaps nothing.  Left
        # out of the groups they are ready as early as before; the layers behind them are launched when the first layer's backward STARTS.
        self.immediate_first_layer = immediate_first_layer
        self.lanes = {}

    def _lane(self):
        st = torch.cuda.current_stream()
        key = (st.device.index, st.cuda_stream)
        lane = self.lanes.get(key)
        if lane is None:
            lane = self.lanes[key] = _WgradLane(st)
        return lane

    @property
    def descs(self):
        return self._lane().descs

    def push(self, descs, tensors):
        lane = self._lane()
        lane.descs.extend(WgradDesc.from_buffer_copy(d) for d in descs)
        lane.alive.append(tensors)

    def add_callbacks(self, cbs):
        self._lane().after.extend(cbs)

    def maybe_flush(self, force=False):
        """the current stream's lane: launch if `group` layers wait (or anything at all with force)"""
        lane = self._lane()
        if lane.descs and (force or len(lane.descs) >= self.limit):
            arr = (WgradDesc * len(lane.descs))(*lane.descs)
            check(lib().ttmi_wgrad_group(arr, c_int(len(lane.descs)), _stream()), "ttmi_wgrad_group")
            cbs = lane.after
            lane.descs, lane.alive, lane.after = [], [], []
            for cb in cbs:
                cb()

    def flush_all(self):
        for lane in list(self.lanes.values()):
            if lane.descs:
                with torch.cuda.stream(lane.stream):
                    self.maybe_flush(force=True)

    def discard(self):
        """an aborted backward pass leaves entries behind: drop them (their gradients are lost with the step)"""
        self.lanes = {}


wgrad_queue = None          # set by ttmi.train.FlatModel.enable_grouped_wgrads(); read by the sub-layer backward passes


def wgrad_defer_supported(rows, d, H, Dh, Di, prec):
    return bool(lib().ttmi_wgrad_defer_supported(c_long(rows), c_int(d), c_int(H), c_int(Dh), c_int(Di), c_int(prec)))


def wgrad_flush(every_stream=False):
    """launch the weight gradients still queued on the current stream (end of an encoder's backward pass), or on every stream"""
    if wgrad_queue is not None:
        if every_stream:
            wgrad_queue.flush_all()
        else:
            wgrad_queue.maybe_flush(force=True)


# ----------------------------------------------------------------------------- sub-layers
class MaskSpec:
    """Attention mask handed to the kernels as parameters: kind 0 none, 1 causal
    (look_ahead_mask), 2 band (context_mask left/right), 3 arbitrary uint8 tensor [B|1, L, L], 4 per-row key intervals: int32 tensor
    [B|1, L, 2] of (lo, hi), key j of query i is masked iff j < lo or j > hi."""
    __slots__ = ("kind", "left", "right", "tensor")

    def __init__(self, kind=0, left=None, right=None, tensor=None):
        unknown = -1 if kind == 4 else 0
        self.kind, self.tensor = kind, tensor
        self.left = unknown if left is None else left
        self.right = unknown if right is None else right

    def args(self):
        t = self.tensor
        if self.kind == 4:
            # left / right: bounds on how far the intervals reach from the diagonal (-1 = not known)
            return c_int(4), c_int(self.left), c_int(self.right), _p(t), c_long(t.stride(0) if t.shape[0] > 1 else 0), c_long(2)
        if self.kind != 3:
            return c_int(self.kind), c_int(self.left), c_int(self.right), c_void_p(0), c_long(0), c_long(0)
        sb = t.stride(0) if t.shape[0] > 1 else 0
        si = t.stride(1) if t.shape[1] > 1 else 0
        return c_int(3), c_int(0), c_int(0), _p(t), c_long(sb), c_long(si)


def _f32(n, device):
    return torch.empty(int(n), dtype=torch.float32, device=device)


_ws_cache = {}
_side_streams = {}


def scratch(n, device):
    """One grow-only scratch arena per (device, stream): sub-layer drivers run back to back on a stream, so the
    scratch of one call is dead when the next call on that stream starts; concurrent streams get their own arena."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() < n:
        t = _ws_cache[key] = _f32(max(int(n), 1 << 20), device)
    return t


def scratch_generation(device, stream):
    """identity of `stream`'s scratch arena (its base address, None before first use): captured graphs hold this pointer, so whoever
    replays them must notice a re-allocation (the arena only ever grows)"""
    t = _ws_cache.get((device.type, device.index, stream.cuda_stream))
    return None if t is None else t.data_ptr()


def attn_fwd(x, p, mask, prec, p_drop=0.0, seed=0):
    """p: dict of parameter tensors (qkv_w, o_w, ln_g, ln_b, r_emb, r_w_bias, r_bias)."""
    _trace("attn_fwd", tuple(x.shape))
    _need_cuda(x)
    B, L, d = x.shape
    K, H, Dh = p["r_emb"].shape
    L_ = lib()
    L_.ttmi_attn_ctx_floats.restype = ctypes.c_size_t
    L_.ttmi_attn_ws_floats.restype = ctypes.c_size_t
    ctx = _f32(L_.ttmi_attn_ctx_floats(c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(prec)), x.device)
    ws = scratch(L_.ttmi_attn_ws_floats(c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(prec)), x.device)
    y = torch.empty_like(x)
    check(L_.ttmi_attn_fwd(_p(x), _p(p["qkv_w"]), _p(p["o_w"]), _p(p["ln_g"]), _p(p["ln_b"]), _p(p["r_emb"]),
                           _p(p["r_w_bias"]), _p(p["r_bias"]), c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(K),
                           *mask.args(), c_int(prec), c_float(p_drop), ctypes.c_uint(seed), _p(ctx), _p(ws), _p(y), _stream()),
          "ttmi_attn_fwd")
    return y, ctx


def attn_bwd(dy, x, p, ctx, prec, grads, p_drop=0.0, seed=0, mask=None, defer=None):
    """grads: dict of ZERO-INITIALISED (or running) f32 buffers, accumulated into.  defer: a WgradQueue - the two weight-gradient GEMMs
    are queued for a grouped launch instead of run here (their gradients appear when the queue is flushed)."""
    _trace("attn_bwd", tuple(x.shape))
    B, L, d = x.shape
    K, H, Dh = p["r_emb"].shape
    L_ = lib()
    L_.ttmi_attn_ws_floats.restype = ctypes.c_size_t
    ws = scratch(L_.ttmi_attn_ws_floats(c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(prec)), x.device)
    dx = torch.empty_like(x)
    if defer is not None:
        L_.ttmi_attn_bwd_keep_bytes.restype = ctypes.c_size_t
        keep = torch.empty(L_.ttmi_attn_bwd_keep_bytes(c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh)), dtype=torch.uint8, device=x.device)
        out = (WgradDesc * 2)()
        check(L_.ttmi_attn_bwd_defer(_p(dy), _p(x), _p(p["qkv_w"]), _p(p["o_w"]), _p(p["ln_g"]), _p(p["r_emb"]), _p(p["r_w_bias"]), _p(p["r_bias"]),
                                     c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(K), *(mask or MaskSpec()).args(), c_int(prec),
                                     c_float(p_drop), ctypes.c_uint(seed), _p(ctx), _p(ws), _p(dx),
                                     _p(grads["qkv_w"]), _p(grads["o_w"]), _p(grads["ln_g"]), _p(grads["ln_b"]), _p(grads["r_emb"]),
                                     _p(grads["r_w_bias"]), _p(grads["r_bias"]), _p(keep), out, _stream()), "ttmi_attn_bwd_defer")
        defer.push(out, (keep, ctx, grads["qkv_w"], grads["o_w"]))
        return dx
    check(L_.ttmi_attn_bwd(_p(dy), _p(x), _p(p["qkv_w"]), _p(p["o_w"]), _p(p["ln_g"]), _p(p["r_emb"]), _p(p["r_w_bias"]), _p(p["r_bias"]),
                           c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(K), *(mask or MaskSpec()).args(), c_int(prec),
                           c_float(p_drop), ctypes.c_uint(seed), _p(ctx), _p(ws), _p(dx),
                           _p(grads["qkv_w"]), _p(grads["o_w"]), _p(grads["ln_g"]), _p(grads["ln_b"]), _p(grads["r_emb"]),
                           _p(grads["r_w_bias"]), _p(grads["r_bias"]), _stream()), "ttmi_attn_bwd")
    return dx


def ffn_fwd(y, p, prec, p_drop=0.0, p_layer=0.0, seed=0):
    _trace("ffn_fwd", tuple(y.shape))
    rows, d = y.numel() // y.shape[-1], y.shape[-1]
    Di = p["ff_w1"].shape[0]
    L_ = lib()
    L_.ttmi_ffn_ctx_floats.restype = ctypes.c_size_t
    L_.ttmi_ffn_ws_floats.restype = ctypes.c_size_t
    ctx = _f32(L_.ttmi_ffn_ctx_floats(c_long(rows), c_int(d), c_int(Di), c_int(prec)), y.device)
    ws = scratch(L_.ttmi_ffn_ws_floats(c_long(rows), c_int(d), c_int(Di), c_int(prec)), y.device)
    z = torch.empty_like(y)
    check(L_.ttmi_ffn_fwd(_p(y), _p(p["ff_w1"]), _p(p["ff_b1"]), _p(p["ff_w2"]), _p(p["ff_b2"]), _p(p["ff_ln_g"]),
                          _p(p["ff_ln_b"]), c_long(rows), c_int(d), c_int(Di), c_int(prec), c_float(p_drop), c_float(p_layer),
                          ctypes.c_uint(seed), _p(ctx), _p(ws), _p(z), _stream()),
          "ttmi_ffn_fwd")
    return z, ctx


def ffn_bwd(dz, y, p, ctx, prec, grads, p_drop=0.0, p_layer=0.0, seed=0, defer=None):
    _trace("ffn_bwd", tuple(y.shape))
    rows, d = y.numel() // y.shape[-1], y.shape[-1]
    Di = p["ff_w1"].shape[0]
    L_ = lib()
    L_.ttmi_ffn_ws_floats.restype = ctypes.c_size_t
    ws = scratch(L_.ttmi_ffn_ws_floats(c_long(rows), c_int(d), c_int(Di), c_int(prec)), y.device)
    dy = torch.empty_like(y)
    if defer is not None:
        L_.ttmi_ffn_bwd_keep_bytes.restype = ctypes.c_size_t
        keep = torch.empty(L_.ttmi_ffn_bwd_keep_bytes(c_long(rows), c_int(d), c_int(Di)), dtype=torch.uint8, device=y.device)
        out = (WgradDesc * 2)()
        check(L_.ttmi_ffn_bwd_defer(_p(dz), _p(y), _p(p["ff_w1"]), _p(p["ff_w2"]), _p(p["ff_ln_g"]), c_long(rows), c_int(d), c_int(Di),
                                    c_int(prec), c_float(p_drop), c_float(p_layer), ctypes.c_uint(seed), _p(ctx), _p(ws), _p(dy),
                                    _p(grads["ff_w1"]), _p(grads["ff_b1"]), _p(grads["ff_w2"]), _p(grads["ff_b2"]), _p(grads["ff_ln_g"]),
                                    _p(grads["ff_ln_b"]), _p(keep), out, _stream()), "ttmi_ffn_bwd_defer")
        defer.push(out, (keep, ctx, grads["ff_w1"], grads["ff_b1"], grads["ff_w2"]))
        return dy
    check(L_.ttmi_ffn_bwd(_p(dz), _p(y), _p(p["ff_w1"]), _p(p["ff_w2"]), _p(p["ff_ln_g"]), c_long(rows), c_int(d), c_int(Di),
                          c_int(prec), c_float(p_drop), c_float(p_layer), ctypes.c_uint(seed), _p(ctx), _p(ws), _p(dy), _p(grads["ff_w1"]), _p(grads["ff_b1"]), _p(grads["ff_w2"]),
                          _p(grads["ff_b2"]), _p(grads["ff_ln_g"]), _p(grads["ff_ln_b"]), _stream()), "ttmi_ffn_bwd")
    return dy


def layer_fused(d, H, Dh, Di, prec):
    """do the layer-level calls share passes between the sub-layers (bf16 pipeline at these widths) - i.e. may x16 / z16 be passed?"""
    return bool(lib().ttmi_layer_fused(c_int(d), c_int(H), c_int(Dh), c_int(Di), c_int(prec)))


def _layer_ws(L_, B, L, d, H, Dh, Di, prec, device):
    L_.ttmi_layer_ws_floats.restype = ctypes.c_size_t
    return scratch(L_.ttmi_layer_ws_floats(c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(Di), c_int(prec)), device)


def layer_fwd(x, x16, pa, pf, mask, prec, p_attn=0.0, seed_attn=0, p_ffn=0.0, p_layer=0.0, seed_ffn=0, want16=False):
    """one encoder layer (attention sub-layer + FFN) in one call: -> (y, z, z16 or None, ctx_attn, ctx_ffn).  pa / pf: the parameter
    dicts of attn_fwd / ffn_fwd; x16: bf16 copy of x from the previous layer's z16 (or None); want16: also return bf16(z)."""
    _trace("layer_fwd", tuple(x.shape))
    _need_cuda(x)
    B, L, d = x.shape
    K, H, Dh = pa["r_emb"].shape
    Di = pf["ff_w1"].shape[0]
    L_ = lib()
    L_.ttmi_attn_ctx_floats.restype = ctypes.c_size_t
    L_.ttmi_ffn_ctx_floats.restype = ctypes.c_size_t
    ctx_a = _f32(L_.ttmi_attn_ctx_floats(c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(prec)), x.device)
    ctx_f = _f32(L_.ttmi_ffn_ctx_floats(c_long(B * L), c_int(d), c_int(Di), c_int(prec)), x.device)
    ws = _layer_ws(L_, B, L, d, H, Dh, Di, prec, x.device)
    y, z = torch.empty_like(x), torch.empty_like(x)
    z16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want16 else None
    check(L_.ttmi_layer_fwd(_p(x), _p(x16), _p(pa["qkv_w"]), _p(pa["o_w"]), _p(pa["ln_g"]), _p(pa["ln_b"]), _p(pa["r_emb"]), _p(pa["r_w_bias"]),
                            _p(pa["r_bias"]), _p(pf["ff_w1"]), _p(pf["ff_b1"]), _p(pf["ff_w2"]), _p(pf["ff_b2"]), _p(pf["ff_ln_g"]),
                            _p(pf["ff_ln_b"]), c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh), c_int(K), c_int(Di), *mask.args(), c_int(prec),
                            c_float(p_attn), ctypes.c_uint(seed_attn), c_float(p_ffn), c_float(p_layer), ctypes.c_uint(seed_ffn), _p(ctx_a),
                            _p(ctx_f), _p(ws), _p(y), _p(z), _p(z16), _stream()), "ttmi_layer_fwd")
    return y, z, z16, ctx_a, ctx_f


def layer_bwd(dz, x, x16, y, pa, pf, ctx_a, ctx_f, prec, grads, mask=None, p_attn=0.0, seed_attn=0, p_ffn=0.0, p_layer=0.0, seed_ffn=0, defer=None):
    """backward of layer_fwd -> dx; grads: running f32 buffers by parameter name (both sub-layers'), accumulated into.  defer: a WgradQueue
    that takes the layer's four weight-gradient GEMMs."""
    _trace("layer_bwd", tuple(x.shape))
    B, L, d = x.shape
    K, H, Dh = pa["r_emb"].shape
    Di = pf["ff_w1"].shape[0]
    L_ = lib()
    ws = _layer_ws(L_, B, L, d, H, Dh, Di, prec, x.device)
    dx = torch.empty_like(x)
    keep_a = keep_f = out = None
    if defer is not None:
        L_.ttmi_attn_bwd_keep_bytes.restype = ctypes.c_size_t
        L_.ttmi_ffn_bwd_keep_bytes.restype = ctypes.c_size_t
        keep_a = torch.empty(L_.ttmi_attn_bwd_keep_bytes(c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh)), dtype=torch.uint8, device=x.device)
        keep_f = torch.empty(L_.ttmi_ffn_bwd_keep_bytes(c_long(B * L), c_int(d), c_int(Di)), dtype=torch.uint8, device=x.device)
        out = (WgradDesc * 4)()
    for nm, t in (("dz", dz), ("x", x), ("y", y), ("ctx_attn", ctx_a), ("ctx_ffn", ctx_f), ("ws", ws), ("dx", dx)):
        if t is None or t.data_ptr() == 0:      # (round 6: one bench run in ~40 died here with the library's anonymous "null pointer"; name the tensor next time)
            raise ValueError("layer_bwd: %s is %s (x %s, prec %d, stream %#x)" % (nm, "None" if t is None else "an empty tensor of shape %s" % (tuple(t.shape),),
                                                                                 tuple(x.shape), prec, torch.cuda.current_stream(x.device).cuda_stream))
    check(L_.ttmi_layer_bwd(_p(dz), _p(x), _p(x16), _p(y), _p(pa["qkv_w"]), _p(pa["o_w"]), _p(pa["ln_g"]), _p(pa["r_emb"]), _p(pa["r_w_bias"]),
                            _p(pa["r_bias"]), _p(pf["ff_w1"]), _p(pf["ff_w2"]), _p(pf["ff_ln_g"]), c_int(B), c_int(L), c_int(d), c_int(H), c_int(Dh),
                            c_int(K), c_int(Di), *(mask or MaskSpec()).args(), c_int(prec), c_float(p_attn), ctypes.c_uint(seed_attn),
                            c_float(p_ffn), c_float(p_layer), ctypes.c_uint(seed_ffn), _p(ctx_a), _p(ctx_f), _p(ws), _p(dx),
                            _p(grads["qkv_w"]), _p(grads["o_w"]), _p(grads["ln_g"]), _p(grads["ln_b"]), _p(grads["r_emb"]), _p(grads["r_w_bias"]),
                            _p(grads["r_bias"]), _p(grads["ff_w1"]), _p(grads["ff_b1"]), _p(grads["ff_w2"]), _p(grads["ff_b2"]),
                            _p(grads["ff_ln_g"]), _p(grads["ff_ln_b"]), _p(keep_a), _p(keep_f), out, _stream()), "ttmi_layer_bwd")
    if defer is not None:
        defer.push(out, (keep_a, keep_f, ctx_a, ctx_f, x16, grads["qkv_w"], grads["o_w"], grads["ff_w1"], grads["ff_b1"], grads["ff_w2"]))
    return dx


def joint_logits_dtype(prec, J):
    return torch.bfloat16 if lib().ttmi_joint_logits_dtype(c_int(prec), c_int(J)) == 1 else torch.float32


def joint_fwd(enc, dec, wf, bf, wp, bp, prec):
    """-> logits [B,T,U1,V] (f32: prec 0 / 2, bf16: prec 1), a [..., :V] view of a pitch-roundup(V,64) buffer"""
    B, T, de = enc.shape
    U1, dd = dec.shape[1], dec.shape[2]
    J, V = wf.shape[0], wp.shape[0]
    L_ = lib()
    L_.ttmi_joint_ctx_floats_prec.restype = ctypes.c_size_t
    L_.ttmi_joint_ws_floats_prec.restype = ctypes.c_size_t
    ctx = _f32(L_.ttmi_joint_ctx_floats_prec(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(prec)), enc.device)
    ws = scratch(L_.ttmi_joint_ws_floats_prec(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(V), c_int(prec)), enc.device)
    dt = joint_logits_dtype(prec, J)
    # (f32 logits too: rows of V = 4334 floats start 8 bytes off every second time - the projection's epilogue then stores float by float, 12 of
    # the bf16x3 mode's 21 ms forward GEMM at C2; on a pitch of roundup(V, 64) every row piece is whole 16-byte stores.  The loss and its
    # gradient take the pitch as they do for bf16 logits)
    buf, logits = padded_empty((B, T, U1, V), dt, enc.device)
    ldv = buf.shape[-1]
    check(L_.ttmi_joint_fwd(_p(enc), _p(dec), _p(wf), _p(bf), _p(wp), _p(bp), c_int(B), c_int(T), c_int(U1), c_int(de),
                            c_int(dd), c_int(J), c_int(V), c_int(prec), _p(ctx), _p(ws), _p(logits), c_long(ldv), _stream()),
          "ttmi_joint_fwd")
    return logits, ctx


def joint_bwd(dlogits, enc, dec, wf, wp, ctx, prec, grads, out=None):
    """out: optional (denc, ddec) buffers to write the input gradients into"""
    B, T, de = enc.shape
    U1, dd = dec.shape[1], dec.shape[2]
    J, V = wf.shape[0], wp.shape[0]
    L_ = lib()
    L_.ttmi_joint_ws_floats_prec.restype = ctypes.c_size_t
    ws = scratch(L_.ttmi_joint_ws_floats_prec(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(V), c_int(prec)), enc.device)
    dt = joint_logits_dtype(prec, J)
    ldg = row_pitch(dlogits)
    if dt is torch.bfloat16:
        ok = (dlogits.dtype is dt and ldg is not None and ldg % 8 == 0 and dlogits.data_ptr() % 16 == 0 and
              (ldg == V or getattr(dlogits, "_ttmi_zero_pad", None) == ldg))
        if not ok:                                  # foreign gradient: repack into a zero-padded bf16 buffer
            buf, view = padded_empty((B, T, U1, V), dt, enc.device)
            buf.zero_()
            view.copy_(dlogits)
            dlogits, ldg = view, buf.shape[-1]
    else:
        if dlogits.dtype is not dt or ldg is None:
            dlogits = dlogits.to(dt).contiguous()
            ldg = V
    denc, ddec = (torch.empty_like(enc), torch.empty_like(dec)) if out is None else out
    assert denc.is_contiguous() and ddec.is_contiguous() and denc.shape == enc.shape and ddec.shape == dec.shape
    check(L_.ttmi_joint_bwd(_p(dlogits), c_long(ldg), _p(enc), _p(dec), _p(wf), _p(wp), c_int(B), c_int(T), c_int(U1), c_int(de),
                            c_int(dd), c_int(J), c_int(V), c_int(prec), _p(ctx), _p(ws), _p(denc), _p(ddec), _p(grads["wf"]),
                            _p(grads["bf"]), _p(grads["wp"]), _p(grads["bp"]), _stream()), "ttmi_joint_bwd")
    return denc, ddec


def joint_loss_split_supported(logits, J, prec):
    """bf16x3 mode: may the loss gradient be written over the f32 logits as the two bf16 planes the joint's backward multiplies (no split pass over d logits)?"""
    if prec != 2 or logits.dtype is not torch.float32 or os.environ.get("TTMI_X3_SPLIT_GRAD", "1") == "0":
        return False
    B, T, U1, V = logits.shape
    ld = row_pitch(logits)
    return (ld is not None and bool(lib().ttmi_rnnt_loss_bwd_split_ok(c_long(ld), _p(logits))) and
            bool(lib().ttmi_joint_bwd_split_ok(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(V), c_int(prec), c_long(ld))))


def rnnt_loss_bwd_split(logits, labels, act_lens, label_lens, blank, workspace, grad_out, grad_out_stride, scale):
    """the f32 logits are REPLACED by their gradient's bf16 planes [hi | lo] per row (ttmi_rnnt_loss_bwd_split); -> the same tensor, to be handed to
    joint_bwd_split only"""
    _need_cuda(logits, labels, act_lens, label_lens, workspace, grad_out)
    B, T, U1, V = logits.shape
    check(lib().ttmi_rnnt_loss_bwd_split(_p(logits), c_long(row_pitch(logits)), _p(labels), _p(act_lens), _p(label_lens), c_int(B), c_int(T), c_int(U1),
                                         c_int(V), c_int(blank), _p(workspace), _p(grad_out), c_int(grad_out_stride), c_float(scale), _stream()),
          "ttmi_rnnt_loss_bwd_split")
    return logits


def joint_bwd_split(planes, enc, dec, wf, wp, ctx, prec, grads, out=None):
    """joint_bwd for d logits that rnnt_loss_bwd_split left as bf16 planes in the logits' own buffer"""
    B, T, de = enc.shape
    U1, dd = dec.shape[1], dec.shape[2]
    J, V = wf.shape[0], wp.shape[0]
    L_ = lib()
    L_.ttmi_joint_ws_floats_prec.restype = ctypes.c_size_t
    ws = scratch(L_.ttmi_joint_ws_floats_prec(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(V), c_int(prec)), enc.device)
    denc, ddec = (torch.empty_like(enc), torch.empty_like(dec)) if out is None else out
    assert denc.is_contiguous() and ddec.is_contiguous() and denc.shape == enc.shape and ddec.shape == dec.shape
    check(L_.ttmi_joint_bwd_split(_p(planes), c_long(row_pitch(planes)), _p(enc), _p(dec), _p(wf), _p(wp), c_int(B), c_int(T), c_int(U1), c_int(de),
                                  c_int(dd), c_int(J), c_int(V), c_int(prec), _p(ctx), _p(ws), _p(denc), _p(ddec), _p(grads["wf"]),
                                  _p(grads["bf"]), _p(grads["wp"]), _p(grads["bp"]), _stream()), "ttmi_joint_bwd_split")
    return denc, ddec


# ---- fused joint + loss fast path (exp-domain forms, include/ttmi.h)
def joint_exp_supported(B, T, U1, J, V, prec, fwd_only=False):
    ldv = (V + 63) // 64 * 64
    fn = lib().ttmi_joint_exp_fwd_supported if fwd_only else lib().ttmi_joint_exp_supported
    return bool(fn(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(V), c_int(prec), c_long(ldv)))


def exp_padded_rows(B, T, U1):
    f = lib().ttmi_joint_exp_padded_rows
    f.restype = c_long
    return int(f(c_int(B), c_int(T), c_int(U1)))


def joint_fwd_exp(enc, dec, wf, bf, wp, bp, prec, shift=None, labels=None, blank=0):
    """-> (P bf16 [B,T,U1,V] view of a pitch-roundup(V,64) buffer = exp(logits - shift), rowsum f32 [nparts, B*T*U1], ctx, emis).
    labels (int32 [B, U1-1]) given: emis f32 [B*T*U1, 4] = the blank's and the next label's logit of every lattice row (from the GEMM's bf16
    operands, then from f32 operands), else None"""
    B, T, de = enc.shape
    U1, dd = dec.shape[1], dec.shape[2]
    J, V = wf.shape[0], wp.shape[0]
    L_ = lib()
    L_.ttmi_joint_ctx_floats_prec.restype = ctypes.c_size_t
    L_.ttmi_joint_ws_floats_prec.restype = ctypes.c_size_t
    ctx = _f32(L_.ttmi_joint_ctx_floats_prec(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(prec)), enc.device)
    ws = scratch(L_.ttmi_joint_ws_floats_prec(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(V), c_int(prec)), enc.device)
    # room for the lattice rows padded to the wgrad's 64-row reduction tile (ttmi_joint_bwd_exp zero-fills the pad rows: include/ttmi.h)
    rows, ldv = B * T * U1, (V + 63) // 64 * 64
    rows_p = exp_padded_rows(B, T, U1)
    buf = torch.empty(rows_p * ldv, dtype=torch.bfloat16, device=enc.device)[:rows * ldv].view(B, T, U1, ldv)
    P = buf[..., :V]
    nparts = L_.ttmi_joint_exp_nparts(c_int(V))
    rowsum = torch.empty(nparts, B * T * U1, dtype=torch.float32, device=enc.device)
    emis = None
    if labels is not None:
        _need_cuda(labels)
        assert labels.dtype is torch.int32 and labels.is_contiguous() and tuple(labels.shape) == (B, U1 - 1)
        emis = torch.empty(B * T * U1, 4, dtype=torch.float32, device=enc.device)
    check(L_.ttmi_joint_fwd_exp(_p(enc), _p(dec), _p(wf), _p(bf), _p(wp), _p(bp), c_int(B), c_int(T), c_int(U1), c_int(de),
                                c_int(dd), c_int(J), c_int(V), c_int(prec), _p(ctx), _p(ws), _p(P), c_long(buf.shape[-1]),
                                _p(rowsum), c_int(nparts), _p(shift), _p(labels), c_int(blank), _p(emis), _stream()), "ttmi_joint_fwd_exp")
    return P, rowsum, ctx, emis


def rnnt_loss_fwd_exp(P, rowsum, labels, act_lens, label_lens, blank, workspace, shift_cur=None, shift_next=None, emis=None, flag=None):
    """emis: f32 [rows, 4] from joint_fwd_exp (the emission log-probs then come from f32 logits, not from bf16 entries of P); flag: device
    int32 [1], bit 0 set when a row sum under- / overflowed (that step's costs and gradients are NaN)"""
    _need_cuda(P, rowsum, labels, act_lens, label_lens, workspace, emis, flag)
    B, T, U1, V = P.shape
    costs = torch.empty(B, dtype=torch.float32, device=P.device)
    check(lib().ttmi_rnnt_loss_fwd_exp(_p(P), c_long(row_pitch(P)), _p(rowsum), c_int(rowsum.shape[0]), _p(labels), _p(act_lens),
                                       _p(label_lens), c_int(B), c_int(T), c_int(U1), c_int(V), c_int(blank), _p(workspace),
                                       _p(costs), _p(shift_cur), _p(shift_next), _p(emis), _p(flag), _stream()), "ttmi_rnnt_loss_fwd_exp")
    return costs


def rnnt_shift_seed(workspace, act_lens, label_lens, B, T, U1, shift_next):
    """shift_next = max(shift_next, max log-sum-exp - 40) over the lattice rows of a PLAIN rnnt_loss_fwd's workspace (device only)"""
    check(lib().ttmi_rnnt_shift_seed(_p(workspace), _p(act_lens), _p(label_lens), c_int(B), c_int(T), c_int(U1), _p(shift_next), _stream()),
          "ttmi_rnnt_shift_seed")


def rnnt_loss_bwd_exp(P, labels, act_lens, label_lens, blank, workspace, grad_out, grad_out_stride, scale):
    """patches P in place -> (srow f32 [rows], srow16 bf16 [rows]): d logits = srow[r] * P[r, :]"""
    B, T, U1, V = P.shape
    rows = B * T * U1
    srow = torch.empty(rows, dtype=torch.float32, device=P.device)
    srow16 = torch.empty(exp_padded_rows(B, T, U1), dtype=torch.bfloat16, device=P.device)[:rows]      # (pad rows: zero-filled by ttmi_joint_bwd_exp)
    check(lib().ttmi_rnnt_loss_bwd_exp(_p(P), c_long(row_pitch(P)), _p(labels), _p(act_lens), _p(label_lens), c_int(B), c_int(T),
                                       c_int(U1), c_int(V), c_int(blank), _p(workspace), _p(grad_out), c_int(grad_out_stride),
                                       c_float(scale), _p(srow), _p(srow16), _stream()), "ttmi_rnnt_loss_bwd_exp")
    return srow, srow16


def joint_bwd_exp(P, srow, srow16, enc, dec, wf, wp, ctx, prec, grads, out=None):
    B, T, de = enc.shape
    U1, dd = dec.shape[1], dec.shape[2]
    J, V = wf.shape[0], wp.shape[0]
    L_ = lib()
    L_.ttmi_joint_ws_floats.restype = ctypes.c_size_t
    ws = scratch(L_.ttmi_joint_ws_floats(c_int(B), c_int(T), c_int(U1), c_int(J), c_int(V)), enc.device)
    denc, ddec = (torch.empty_like(enc), torch.empty_like(dec)) if out is None else out
    assert denc.is_contiguous() and ddec.is_contiguous() and denc.shape == enc.shape and ddec.shape == dec.shape
    check(L_.ttmi_joint_bwd_exp(_p(P), c_long(row_pitch(P)), _p(srow), _p(srow16), _p(enc), _p(dec), _p(wf), _p(wp), c_int(B),
                                c_int(T), c_int(U1), c_int(de), c_int(dd), c_int(J), c_int(V), c_int(prec), _p(ctx), _p(ws),
                                _p(denc), _p(ddec), _p(grads["wf"]), _p(grads["bf"]), _p(grads["wp"]), _p(grads["bp"]), _stream()),
          "ttmi_joint_bwd_exp")
    return denc, ddec


def embed_fwd(tokens, W):
    _need_cuda(tokens, W)
    if tokens.dtype is not torch.long:          # the kernel reads int64 ids; nn.Embedding (tt/decoder.py:26) also takes int32
        if tokens.dtype is not torch.int32:
            raise TypeError("embedding indices must be int64 or int32, got %s" % tokens.dtype)
        tokens = tokens.long()
    tokens = tokens.contiguous()
    n, (V, d) = tokens.numel(), W.shape
    out = torch.empty(*tokens.shape, d, dtype=torch.float32, device=W.device)
    check(lib().ttmi_embed_fwd(_p(tokens), _p(W), c_long(n), c_int(d), c_int(V), _p(out), _stream()), "ttmi_embed_fwd")
    return out


def embed_bwd(tokens, dout, V, padding_idx, gW):
    n, d = tokens.numel(), dout.shape[-1]
    check(lib().ttmi_embed_bwd(_p(tokens), _p(dout), c_long(n), c_int(d), c_int(V), c_int(padding_idx), _p(gW), _stream()),
          "ttmi_embed_bwd")
    return gW


# ----------------------------------------------------------------------------- optimiser tail / probes
def sumsq(x, out):
    check(lib().ttmi_sumsq(_p(x), c_long(x.numel()), _p(out), _stream()), "ttmi_sumsq")


def sgd_step(p, g, mom, lr, momentum, weight_decay, nesterov, max_norm, normsq, grad_scale, hyper=None):
    """hyper: device f32 [2] = (learning rate, steps taken), read (and the count advanced) by the kernels at run time - see include/ttmi.h"""
    check(lib().ttmi_sgd_step(_p(p), _p(g), _p(mom), c_long(p.numel()), c_float(lr), c_float(momentum), c_float(weight_decay),
                              c_int(1 if nesterov else 0), c_float(max_norm), _p(normsq), c_float(grad_scale), _p(hyper), _stream()),
          "ttmi_sgd_step")


def adam_step(p, g, m, v, lr, betas, eps, weight_decay, step, max_norm, normsq, grad_scale, hyper=None):
    check(lib().ttmi_adam_step(_p(p), _p(g), _p(m), _p(v), c_long(p.numel()), c_float(lr), c_float(betas[0]), c_float(betas[1]),
                               c_float(eps), c_float(weight_decay), c_int(step), c_float(max_norm), _p(normsq),
                               c_float(grad_scale), _p(hyper), _stream()), "ttmi_adam_step")


def adadelta_step(p, g, sq, acc, lr, rho, eps, weight_decay, max_norm, normsq, grad_scale, hyper=None):
    check(lib().ttmi_adadelta_step(_p(p), _p(g), _p(sq), _p(acc), c_long(p.numel()), c_float(lr), c_float(rho), c_float(eps),
                                   c_float(weight_decay), c_float(max_norm), _p(normsq), c_float(grad_scale), _p(hyper), _stream()), "ttmi_adadelta_step")


def probe_arm(slot=0):
    check(lib().ttmi_probe_arm(c_int(slot)), "ttmi_probe_arm")


def probe_read_ms(slot=0, point=0):
    """duration of probe `point` (0 joint projection, 1 RNN-T loss forward, 2 RNN-T loss backward) armed in `slot`; < 0 = never fired"""
    f = lib().ttmi_probe_point_read_ms
    f.restype = ctypes.c_float
    return float(f(c_int(point), c_int(slot)))


# ----------------------------------------------------------------------------- throughput bf16 GEMMs (tests / tools)
def gemm_nt_bf16(A, B, C, bias=None):
    """C[M,N] = A[M,K] @ B[N,K]^T (+bias).  A, B bf16 row-major (row pitch = stride(0)); C f32 or bf16."""
    M, K = A.shape
    N = B.shape[0]
    check(lib().ttmi_gemm_nt_bf16(_p(A), _p(B), _p(C), c_int(_DT[C.dtype]), _p(bias), c_int(M), c_int(N), c_int(K),
                                  c_long(A.stride(0)), c_long(B.stride(0)), c_long(C.stride(0)), _stream()), "ttmi_gemm_nt_bf16")
    return C


def gemm_nt_bf16_two_term(A, B, B_lo, C, bias=None, relu=False, k_lo=0):
    """C[M,N] = act(A[M,K] @ (B + B_lo)[N,K]^T + bias) in one launch; k_lo != 0: B_lo meets A's first k_lo columns only."""
    M, K = A.shape
    N = B.shape[0]
    assert B_lo.shape[0] == B.shape[0] and B_lo.stride(0) == B.stride(0)
    if k_lo:
        check(lib().ttmi_gemm_nt_bf16_two_term_klo(_p(A), _p(B), _p(B_lo), _p(C), c_int(_DT[C.dtype]), _p(bias), c_int(1 if relu else 0), c_int(M),
                                                   c_int(N), c_int(K), c_int(k_lo), c_long(A.stride(0)), c_long(B.stride(0)), c_long(C.stride(0)), _stream()),
              "ttmi_gemm_nt_bf16_two_term_klo")
        return C
    check(lib().ttmi_gemm_nt_bf16_two_term(_p(A), _p(B), _p(B_lo), _p(C), c_int(_DT[C.dtype]), _p(bias), c_int(1 if relu else 0), c_int(M),
                                           c_int(N), c_int(K), c_long(A.stride(0)), c_long(B.stride(0)), c_long(C.stride(0)), _stream()),
          "ttmi_gemm_nt_bf16_two_term")
    return C


def gemm_tn_bf16(A, B, C, accumulate=False, colsum_a=None):
    """C[M,N] (f32) (+)= A[K,M]^T @ B[K,N].  A, B bf16 row-major.  colsum_a (f32 [M]) += column sums of A."""
    K, M = A.shape
    N = B.shape[1]
    check(lib().ttmi_gemm_tn_bf16(_p(A), _p(B), _p(C), c_int(M), c_int(N), c_int(K), c_long(A.stride(0)), c_long(B.stride(0)),
                                  c_long(C.stride(0)), c_int(1 if accumulate else 0), _p(colsum_a), _stream()), "ttmi_gemm_tn_bf16")
    return C


def dropout_multipliers(n, p, seed, device):
    """the exact 0 / 1/(1-p) multipliers the fused sub-layers use for dropout site `seed` (tests, debugging)"""
    ones = torch.ones(n, dtype=torch.float32, device=device)
    out = torch.empty_like(ones)
    check(lib().ttmi_dropout_apply(_p(ones), c_long(n), c_float(p), ctypes.c_uint(seed), _p(out), _stream()), "ttmi_dropout_apply")
    return out


_salt_keep = [None]


def set_dropout_salt(t):
    """t: int32 / uint32 device tensor [1] (or None): a word every dropout site mixes into its seed at kernel start; bump it on the device
    between steps when the step is replayed from a captured graph (ttmi.train.GraphedStep)"""
    if t is not None:
        _need_cuda(t)
        assert t.numel() == 1 and t.element_size() == 4
    _salt_keep[0] = t                       # the library holds the raw pointer: keep the tensor alive
    check(lib().ttmi_set_dropout_salt(_p(t)), "ttmi_set_dropout_salt")


def set_option(key, value):
    """process-wide A/B switches (key 0: 1 disables the fused attention kernels of the bf16 pipeline)"""
    check(lib().ttmi_set_option(c_int(key), c_int(value)), "ttmi_set_option")


def reserve_cus(n, device=None):
    """CUs the encoder-sized persistent GEMMs leave free on the CURRENT stream and on the label encoder's side stream (data-parallel
    backward: RCCL's kernels run beside it).  Per-stream state in the library, read at launch time; 0 = whole chip."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    check(lib().ttmi_stream_reserve_cus(c_void_p(torch.cuda.current_stream(device).cuda_stream), c_int(n)), "ttmi_stream_reserve_cus")
    st = _side_streams.get((device.type, device.index))
    if st is not None:
        check(lib().ttmi_stream_reserve_cus(c_void_p(st.cuda_stream), c_int(n)), "ttmi_stream_reserve_cus")


def side_stream(device):
    """the per-device side stream on which the label encoder runs concurrently with the audio encoder"""
    key = (device.type, device.index)
    st = _side_streams.get(key)
    if st is None:
        # TTMI_SIDE_STREAM_PRIORITY (A/B knob, round 6): torch clamps it to the device's range; lower number = served first
        st = _side_streams[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("TTMI_SIDE_STREAM_PRIORITY", "0")))
        # a second-level stream (forked from the caller's main stream): library calls on it never fork again inside a stream capture
        check(lib().ttmi_stream_set_nofork(c_void_p(st.cuda_stream), c_int(1)), "ttmi_stream_set_nofork")
    return st


def join_side_streams():
    """make the current stream wait for everything queued on the side streams (call before consuming gradients that
    backward nodes running on a side stream wrote in place, e.g. before the optimiser step / gradient all-reduce)"""
    for st in _side_streams.values():
        torch.cuda.current_stream(st.device).wait_stream(st)


def greedy_scan(logits2d, blank=0):
    """logits2d [n, V] (f32/bf16, row pitch = stride(0)) -> (first row whose argmax != blank, symbol) or (n, None); one 8-byte
    D2H read."""
    n, V = logits2d.shape
    out = torch.empty(1, dtype=torch.int64, device=logits2d.device)
    check(lib().ttmi_greedy_scan(_p(logits2d), c_int(_DT[logits2d.dtype]), c_long(logits2d.stride(0)), c_int(n), c_int(V),
                                 c_int(blank), _p(out), _stream()), "ttmi_greedy_scan")
    key = int(out.item())
    row, tok = key >> 32, key & 0xffffffff
    return (row, tok) if row < n else (n, None)


def greedy_scan_batch(logits, t, T_len, need, key, blank=0):
    """logits [B, n, V] (f32 / bf16, row pitch = stride(-2)): per utterance the first existing frame of the block whose argmax is not blank
    -> key[b] (include/ttmi.h: ttmi_greedy_scan_batch); device only, no synchronisation"""
    B, n, V = logits.shape
    _need_cuda(logits, t, T_len, need, key)
    check(lib().ttmi_greedy_scan_batch(_p(logits), c_int(_DT[logits.dtype]), c_long(logits.stride(-2)), c_int(B), c_int(n), c_int(V), c_int(blank),
                                       _p(t), _p(T_len), _p(need), _p(key), _stream()), "ttmi_greedy_scan_batch")


def greedy_advance(key, n, n_hist, hist, t, T_len, need, done, count, flags):
    """consume key (ttmi_greedy_advance): histories, frame positions and the need / done flags move on the device; flags[0] = utterances
    that still need a symbol in this step, flags[1] = utterances not finished"""
    B = key.shape[0]
    assert hist.dtype is torch.long and hist.stride(1) == 1
    check(lib().ttmi_greedy_advance(_p(key), c_int(B), c_int(n), c_int(n_hist), _p(hist), c_long(hist.stride(0)), _p(t), _p(T_len), _p(need),
                                    _p(done), _p(count), _p(flags), _stream()), "ttmi_greedy_advance")
