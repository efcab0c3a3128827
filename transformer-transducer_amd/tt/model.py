"""Transducer / JointNet with the reference's API surface (tt/model.py): .encoder/.decoder/.joint,
forward(inputs[B,T,d], targets[B,U]) -> logits[B,T,U+1,V], decode/recognize (greedy), beam search."""
import copy
import heapq
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ttmi import ops
from ttmi.ops import MaskSpec
from tt.decoder import BuildDecoder
from tt.encoder import BuildEncoder
from tt.transformer import label_value_precision, default_precision, grad_targets
from tt.utils import context_mask, look_ahead_mask  # noqa: F401  (re-exported like the reference module)


class _JointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, dec, wf, bf, wp, bp, prec):
        enc, dec = enc.contiguous(), dec.contiguous()
        params = (wf, bf, wp, bp)
        wf, bf, wp, bp = (t.detach() for t in (wf, bf, wp, bp))
        logits, saved = ops.joint_fwd(enc, dec, wf, bf, wp, bp, prec)
        ctx.save_for_backward(enc, dec, wf, wp, saved)
        ctx.prec = prec
        ctx.params = params
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        enc, dec, wf, wp, saved = ctx.saved_tensors
        g, rets, after = grad_targets(ctx.params, ("wf", "bf", "wp", "bp"))
        denc, ddec = ops.joint_bwd(dlogits, enc, dec, wf, wp, saved, ctx.prec, g)
        for cb in after:
            cb()
        return (denc, ddec, *rets, None)


class _ExpShift:
    """Range control of the exp-domain loss form, per JointNet module and device (never shared between models): the device scalars `cur`
    (shift subtracted before exp in this step) and `nxt` (gathered for the next one), and a sticky device `flag` the loss kernel raises when
    a lattice row's sum under- or overflowed.  No host synchronisation anywhere:

    * not `valid` (first use, after JointNet.load_state_dict - a hook invalidates it -, or after a flagged step): the step runs the plain
      fused form, whose log-sum-exp pass also seeds `nxt` (ttmi_rnnt_shift_seed); from the next step on the exp-domain kernels run.
      Weights overwritten behind the module's back (p.data.copy_) are caught by the flag if their logits left the 40 e-fold margin.
    * every exp step leaves max(log-sum-exp) - 40 in `nxt`; it becomes `cur` only if an exp (or seeding) chunk actually ran.
    * the flag is copied to pinned host memory on a side stream and looked at (event query, non-blocking) at the next calls; a flagged
      step's costs and gradients are NaN by construction (never finite and wrong) and the fused optimiser drops such a step."""

    def __init__(self, device):
        self.cur = torch.zeros(1, dtype=torch.float32, device=device)
        self.nxt = torch.zeros(1, dtype=torch.float32, device=device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.copy_stream = torch.cuda.Stream(device)
        self.pending = []               # (event, generation) of flag copies in flight
        self.valid, self.gen, self.flagged_steps = False, 0, 0

    def set(self, shift):
        """start from a known shift (tests, callers that know their logits' scale)"""
        self.cur.fill_(float(shift))
        self.nxt.zero_()
        self.valid = True

    def poll(self):
        """look at completed flag copies; a raised flag invalidates the shift (the next step re-seeds it in the plain form)"""
        while self.pending and self.pending[0][0].query():
            _, gen = self.pending.pop(0)
            if gen == self.gen and int(self.host[0]) != 0:
                import warnings
                self.gen += 1
                self.flagged_steps += 1
                self.valid = False
                self.flag.zero_()
                warnings.warn("exp-domain RNN-T loss: a lattice row's sum of exponentials under/overflowed (the logits moved by more than "
                              "the 40 e-fold margin within one step); that step's loss and gradients were NaN and FusedOptimizer dropped "
                              "it; the next step runs the plain fused form and re-seeds the shift")

    def watch(self):
        """queue an asynchronous copy of the flag behind the work issued so far"""
        main = torch.cuda.current_stream(self.cur.device)
        self.copy_stream.wait_stream(main)
        with torch.cuda.stream(self.copy_stream):
            self.host.copy_(self.flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.pending.append((ev, self.gen))
        if len(self.pending) > 8:       # bounded: the oldest copy finished long ago
            self.pending[0][0].synchronize()
            self.poll()


_warned_no_exp = set()


def _warn_no_exp(Bc, T, U1, J, V):
    """one warning per shape class: the exp-domain form was asked for and cannot run (the plain fused form runs instead - same loss and
    gradients up to bf16 rounding, about 4 ms slower per C2-sized step)"""
    key = (J, V, Bc * T * U1 < 32768)
    if key in _warned_no_exp:
        return
    _warned_no_exp.add(key)
    import warnings
    warnings.warn("exp-domain RNN-T loss form: a chunk of %d x %d x %d lattice rows with joint sizes J=%d, V=%d is outside the persistent "
                  "kernels' sizes (they need >= 32768 lattice rows per chunk, J a multiple of 64 and >= 256, V >= 1024); the plain fused "
                  "joint + loss form runs instead" % (Bc, T, U1, J, V))


class _JointLossFn(torch.autograd.Function):
    """joint network + RNN-T loss as ONE op that never holds the [B, T, U+1, V] logits: the batch is cut into chunks
    of utterances; per chunk the logits are produced (ttmi_joint_fwd), reduced to the lattice (ttmi_rnnt_loss_fwd), overwritten IN PLACE
    by their own gradient (ttmi_rnnt_loss_bwd, the same kernels as RNNTLoss) and consumed by the joint's backward (ttmi_joint_bwd) - in
    the forward pass, as warp-transducer itself forms its gradients - so one chunk's buffer is the whole footprint (C2: 14.2 GB of
    logits + gradient -> 0.44 GB per chunk of 2 utterances; C5: 55.8 GB -> 3.5 GB per utterance).  backward() only scales by the
    incoming gradient."""

    @staticmethod
    def forward(ctx, enc, dec, wf, bf, wp, bp, labels, act_lens, label_lens, prec, chunk, reduction, exp_state, grad_mode=True, blank=0):
        """exp_state: None (plain fused form) or the JointNet's _ExpShift for this device (exp-domain form)"""
        enc, dec = enc.contiguous(), dec.contiguous()
        params = (wf, bf, wp, bp)
        wf_, bf_, wp_, bp_ = (t.detach() for t in params)
        B, T = enc.shape[0], enc.shape[1]
        U1 = dec.shape[1]
        need = grad_mode and any(ctx.needs_input_grad[:6])      # (needs_input_grad ignores torch.no_grad(); forward() itself always runs without grad mode)
        costs = torch.empty(B, dtype=torch.float32, device=enc.device)
        one = torch.ones(1, dtype=torch.float32, device=enc.device)
        scale = 1.0 / B if reduction == "mean" else 1.0
        if need:
            denc, ddec = torch.empty_like(enc), torch.empty_like(dec)
            g = {n: torch.zeros_like(t) for n, t in zip(("wf", "bf", "wp", "bp"), params)}
        J, V = wf.shape[0], wp.shape[0]
        st = exp_state
        exp_ran = seeded = False
        capturing = enc.is_cuda and torch.cuda.is_current_stream_capturing()      # (ttmi.train.GraphedStep looks at the flag between replays)
        if st is not None and not capturing:
            st.poll()
        for c0 in range(0, B, chunk):
            c1 = min(B, c0 + chunk)
            ws = ops.rnnt_workspace(c1 - c0, T, U1, enc.device)
            lab, al, ll = labels[c0:c1], act_lens[c0:c1], label_lens[c0:c1]
            exp_ok = st is not None and ops.joint_exp_supported(c1 - c0, T, U1, J, V, prec, fwd_only=not need)
            if st is not None and not exp_ok:
                _warn_no_exp(c1 - c0, T, U1, J, V)
            if exp_ok and st.valid:
                # the projection stores exp(logit - shift) and row sums; the loss reads the sums and two f32 logits per row, its gradient
                # stays factored as (row factor) x P and is consumed in that form (include/ttmi.h, "fused joint + loss fast path")
                P, rowsum, saved, emis = ops.joint_fwd_exp(enc[c0:c1], dec[c0:c1], wf_, bf_, wp_, bp_, prec, st.cur, lab.contiguous(), blank)
                costs[c0:c1] = ops.rnnt_loss_fwd_exp(P, rowsum, lab, al, ll, blank, ws, st.cur, st.nxt, emis, st.flag)
                if need:
                    srow, srow16 = ops.rnnt_loss_bwd_exp(P, lab, al, ll, blank, ws, one, 0, scale)
                    ops.joint_bwd_exp(P, srow, srow16, enc[c0:c1], dec[c0:c1], wf_, wp_, saved, prec, g, out=(denc[c0:c1], ddec[c0:c1]))
                del P, rowsum, saved, emis
                exp_ran = True
                continue
            logits, saved = ops.joint_fwd(enc[c0:c1], dec[c0:c1], wf_, bf_, wp_, bp_, prec)
            costs[c0:c1] = ops.rnnt_loss_fwd(logits, lab, al, ll, blank, ws)
            if st is not None and not st.valid:                 # the plain form's log-sum-exp pass seeds the shift for the next step
                ops.rnnt_shift_seed(ws, al, ll, c1 - c0, T, U1, st.nxt)
                seeded = True
            if need and ops.joint_loss_split_supported(logits, wf_.shape[0], prec):
                # bf16x3: the gradient leaves the loss kernel as the two bf16 planes the joint's three-term backward multiplies (round 6: no split pass over d logits)
                planes = ops.rnnt_loss_bwd_split(logits, lab, al, ll, blank, ws, one, 0, scale)
                ops.joint_bwd_split(planes, enc[c0:c1], dec[c0:c1], wf_, wp_, saved, prec, g, out=(denc[c0:c1], ddec[c0:c1]))
            elif need:
                grad = ops.rnnt_loss_bwd(logits, lab, al, ll, blank, ws, one, 0, scale, inplace=True)
                ops.joint_bwd(grad, enc[c0:c1], dec[c0:c1], wf_, wp_, saved, prec, g, out=(denc[c0:c1], ddec[c0:c1]))
            del logits, saved
        if st is not None and (exp_ran or seeded):
            st.cur.copy_(st.nxt)
            st.nxt.zero_()
            st.valid = True
            if exp_ran and not capturing:
                st.watch()
        if need:
            ctx.save_for_backward(denc, ddec, *g.values())
        ctx.params = params
        if reduction == "none":
            return costs
        return costs.sum().reshape(1) * scale

    @staticmethod
    def backward(ctx, gout):
        denc, ddec, *gs = ctx.saved_tensors
        gout = gout.float()
        if gout.numel() != 1:
            raise NotImplementedError("fused joint + loss: per-utterance upstream gradients (reduction='none') are not supported; "
                                      "use model(inputs, targets) + RNNTLoss(reduction='none')")
        rets = []
        for prm, gp in zip(ctx.params, gs):
            if getattr(prm, "_ttmi_direct", False) and prm.grad is not None:       # FlatModel: accumulate into the flat gradient buffer
                prm.grad.addcmul_(gp, gout)
                rets.append(None)
                cb = getattr(prm, "_ttmi_on_grad", None)
                if cb is not None:
                    cb()
            else:
                rets.append(gp * gout)
        return (denc * gout, ddec * gout, *rets, None, None, None, None, None, None, None, None, None)


def deferred_logits_enabled(config, prec):
    """does Transducer.forward hand out a DeferredLogits handle?  TTMI_DEFERRED_LOGITS=0 / 1 (read per call, like TTMI_PRECISION) or
    config.deferred_logits (True / False) decide; unset: on in the bf16 pipeline (the throughput mode - its logits are 7 GB of bf16 per
    C2 step that train.py:51-53 only ever hands to the loss) and, since round 6, in the bf16x3 mode (14 GB of f32 logits and 14 GB of gradient:
    fused, the loss gradient leaves its kernel as the bf16 planes the joint's three-term backward multiplies - 98.5 -> 93.4 ms per C2 step), off in
    the fp32 parity mode."""
    env = os.environ.get("TTMI_DEFERRED_LOGITS")
    if env is not None and env != "":
        return env != "0"
    if config is not None and config.deferred_logits is not None:
        return bool(config.deferred_logits)
    return prec in (1, 2)


def _meta_functions():
    """Tensor methods / property getters a DeferredLogits answers from its own metadata, without producing the logits"""
    T = torch.Tensor
    fns = {T.size, T.dim, T.ndimension, T.numel, T.nelement, T.element_size, T.is_floating_point, T.is_complex, T.get_device, T.__len__}
    # (not stride / is_contiguous: the eager logits are a [..., :V] view of a pitched buffer - those come from the real tensor)
    for name in ("shape", "dtype", "device", "ndim", "is_cuda", "is_cpu", "layout", "is_sparse", "is_quantized", "is_meta", "is_nested",
                 "requires_grad", "itemsize", "names"):
        # (grad_fn / is_leaf / grad are NOT answered here: they are the real logits' autograd state - a caller that checks `logits.grad_fn`
        # must see the joint's node - so they materialise, with the one-time warning below when that costs gigabytes)
        fns.add(getattr(T, name).__get__)
    return frozenset(fns)


class DeferredLogits(torch.Tensor):
    """What `logits = model(inputs, targets)` (train.py:51, tt/model.py:58-68) returns in the bf16 pipeline: a tensor-shaped HANDLE on the
    two encoder states.  train.py:53 hands it to `RNNTLoss` unchanged, and the `warprnnt_pytorch` shim then runs joint + loss as ONE fused
    op on those states (`_JointLossFn`: the exp-domain form when its kernels take the sizes, else the chunked memory form) - the
    [B, T, U+1, V] logits (7.1 GB of bf16 per C2 step, written once and walked twice by the loss) are never formed, which is what
    `Transducer.loss(..., exp_domain=True)` does for callers that changed their code.

    Any OTHER use - arithmetic, indexing, `.float()`, `softmax`, printing, `.data_ptr()`, `.stride()`, `torch.save`, a different loss -
    materialises exactly the tensor `model.joint(enc_state, dec_state)` returns today (same autograd graph into the encoders, same bits;
    produced once and kept) and carries on with it, and a handle that has been materialised is consumed by `RNNTLoss` as an ordinary
    logits tensor.  Metadata (`shape`, `size()`, `dim()`, `dtype`, `device`, `is_cuda`, `numel()`, `len()`, `requires_grad`) is answered
    from the handle.  Implementation: a storage-less wrapper subclass (`Tensor._make_wrapper_subclass`) whose `__torch_function__`
    swaps the handle for the real logits before any function that is not a metadata query runs."""

    _warned_big = False

    @staticmethod
    def __new__(cls, joint, enc_state, dec_state, prec):
        B, T = enc_state.shape[0], enc_state.shape[1]
        U1 = dec_state.shape[1]
        J, V = joint.forward_layer.out_features, joint.project_layer.out_features
        grad_mode = torch.is_grad_enabled()
        needs = grad_mode and (enc_state.requires_grad or dec_state.requires_grad or any(p.requires_grad for p in joint.parameters()))
        self = torch.Tensor._make_wrapper_subclass(cls, (B, T, U1, V), dtype=ops.joint_logits_dtype(prec, J), device=enc_state.device,
                                                   requires_grad=bool(needs))
        self._joint, self._enc, self._dec, self._prec, self._grad_mode, self._real = joint, enc_state, dec_state, prec, grad_mode, None
        if needs:
            # a gradient that arrives at the HANDLE (a caller below the Python API recorded it as an autograd input, e.g. a foreign
            # autograd.Function applied to it directly) belongs to the real logits: pass it on into their graph
            with torch._C.DisableTorchFunctionSubclass():
                torch.Tensor.register_hook(self, self._forward_gradient)
        return self

    def __init__(self, *args, **kwargs):
        pass

    def _forward_gradient(self, grad):
        if self._real is not None and self._real.requires_grad:
            torch.autograd.backward(self._real, grad)
        return None

    @property
    def is_materialized(self):
        return self._real is not None

    def _produce(self):
        j = self._joint
        ops.weights_fresh()
        return _JointFn.apply(self._enc, self._dec, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight,
                              j.project_layer.bias, self._prec)

    def materialize(self):
        """-> the real logits (produced on first use under the grad mode of the forward call that made the handle, then kept)"""
        if self._real is None:
            nbytes = self.numel() * self.element_size()
            if nbytes >= (1 << 30) and not DeferredLogits._warned_big:
                import warnings
                DeferredLogits._warned_big = True
                warnings.warn("DeferredLogits: a use other than RNNTLoss / shape queries is forming the real [B, T, U+1, V] logits (%.1f GB, kept "
                              "for the life of the handle); pass the handle straight to RNNTLoss to stay on the fused path, or set "
                              "TTMI_DEFERRED_LOGITS=0 if the logits are wanted anyway" % (nbytes / 2.0 ** 30))
            with torch.set_grad_enabled(self._grad_mode):
                self._real = self._produce()
        return self._real

    def rnnt_loss(self, labels, act_lens, label_lens, blank=0, reduction="mean"):
        """the loss of train.py:53 on this handle's logits without forming them (called by warprnnt_pytorch.rnnt_loss; arguments already
        certified there).  None when this case has to go through the real logits (per-utterance costs that need gradients)."""
        if self._real is not None:
            return None
        grad = self._grad_mode and torch.is_grad_enabled()
        if reduction == "none" and grad and self.requires_grad:
            return None                     # per-utterance upstream gradients cannot be folded into weight gradients formed in the forward pass
        j = self._joint
        B, T, U1 = self.shape[0], self.shape[1], self.shape[2]
        ops.weights_fresh()
        exp = self._prec == 1 and os.environ.get("TTMI_DEFERRED_EXP", "1") != "0"
        chunk = j.default_loss_chunk(B, T, U1, exp, self._prec)
        if exp and not ops.joint_exp_supported(chunk, T, U1, j.forward_layer.out_features, j.project_layer.out_features, self._prec,
                                               fwd_only=not (grad and self.requires_grad)):
            # the exp-domain kernels do not take this problem: the plain fused form runs, and ITS chunk is the memory form's (about 2 GB of
            # logits per chunk, not the 32 GB budget of the form that never holds them)
            chunk = j.default_loss_chunk(B, T, U1, False, self._prec)
        return _JointLossFn.apply(self._enc, self._dec, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight,
                                  j.project_layer.bias, labels, act_lens, label_lens, self._prec, int(chunk), reduction,
                                  j.exp_shift_state(self._enc.device) if exp else None, grad, int(blank))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _META_FUNCTIONS:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)

        def real(a):
            if isinstance(a, DeferredLogits):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(x) for x in a)
            return a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*real(args), **{k: real(v) for k, v in kwargs.items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # reached only by callers below the Python API (the handle has no storage): give them the real values.  Those values are
        # DETACHED - a foreign op applied at this level to a handle that needs gradients would silently cut the encoders off the graph,
        # so that case raises (grad mode on and the handle requires grad); inference / no_grad callers get the values
        import torch.utils._pytree as pytree

        def real(a):
            if a.requires_grad and torch.is_grad_enabled():
                raise RuntimeError("DeferredLogits reached %s below the Python API while it requires grad: the op would run on detached "
                                   "logits and lose the gradient into the encoders.  Call logits.materialize() first (or set "
                                   "TTMI_DEFERRED_LOGITS=0)" % (func,))
            return a.materialize().detach()
        args, kwargs = pytree.tree_map_only(DeferredLogits, real, (args, kwargs or {}))
        return func(*args, **kwargs)


_META_FUNCTIONS = _meta_functions()


class JointNet(nn.Module):
    """logits = project_layer(tanh(forward_layer(cat(enc, dec)))) evaluated in split-weight form
    (forward_layer.weight = [W_enc | W_dec]); accepts [B,T,de]/[B,U,dd] (lattice) or two 1-D vectors (decode)."""

    def __init__(self, input_size, inner_dim, vocab_size):
        super().__init__()
        self.forward_layer = nn.Linear(input_size, inner_dim, bias=True)
        self.tanh = nn.Tanh()
        self.project_layer = nn.Linear(inner_dim, vocab_size, bias=True)
        self._register_load_state_dict_pre_hook(self._new_weights)

    def _new_weights(self, *args):
        """load_state_dict: the logits' scale is unknown again - the exp-domain loss form re-seeds its shift at the next step"""
        for st in self.__dict__.get("_exp_shift", {}).values():
            st.valid = False

    def forward(self, enc_state, dec_state):
        ops.weights_fresh()
        if enc_state.dim() == 3 and dec_state.dim() == 3:
            squeeze = None
        else:
            assert enc_state.dim() == dec_state.dim()
            squeeze = enc_state.shape[:-1]
            enc_state = enc_state.reshape(-1, 1, enc_state.shape[-1])     # N vectors -> N lattices of 1 x 1
            dec_state = dec_state.reshape(-1, 1, dec_state.shape[-1])
        out = _JointFn.apply(enc_state, dec_state, self.forward_layer.weight, self.forward_layer.bias,
                             self.project_layer.weight, self.project_layer.bias, default_precision())
        return out if squeeze is None else out.reshape(*squeeze, out.shape[-1])

    def exp_shift_state(self, device):
        """this module's range-control state of the exp-domain loss form on `device` (_ExpShift; created on first use, not part of
        state_dict)"""
        states = self.__dict__.setdefault("_exp_shift", {})
        st = states.get(device)
        if st is None:
            st = states[device] = _ExpShift(device)
        return st

    def default_loss_chunk(self, B, T, U1, exp_domain=False, prec=None):
        """utterances per chunk of the fused joint + loss: about 2 GB of logits (the memory-saving form) or 32 GB (exp_domain: the speed
        form - every chunk boundary costs a pipeline fill of the three big GEMMs and one more lattice launch: C2 whole batch 35.0 ms per
        step, two halves 36.2; C5's 28 GB in one chunk 102.0 ms, in two 104.2 - on 288 GB of HBM the budget is not the constraint).  The
        chunks are balanced (B = 33 at a budget of 32 gives 17 + 16, not 32 + 1: every chunk stays above the persistent kernels' minimum
        row count).  Any row count runs the exp-domain kernels: the library pads the wgrad's reduction to its 64-row tile (round 3 needed
        chunk * T * U1 % 64 == 0 and fell back to the plain form otherwise - half of all real batches at B = 32)."""
        prec = default_precision() if prec is None else prec
        es = 2 if ops.joint_logits_dtype(prec, self.forward_layer.out_features) is torch.bfloat16 else 4
        budget = (32 << 30) if (exp_domain or prec == 2) else (2 << 30)     # (bf16x3: the speed form as well - C2 in eight chunks 108 ms, in one 93.4)
        chunk = max(1, min(B, int(budget // (es * T * U1 * self.project_layer.out_features))))
        n = -(-B // chunk)
        return -(-B // n)


class _LabelStateGraphs:
    """Greedy decoding re-runs the label encoder on the whole token history after every emitted symbol (tt/model.py:75,88 of the
    reference): ~100 tiny launches whose cost is the host issuing them.  One captured graph per history length L replays them with a
    single launch; the token histories live on the device (`master` [B, MAX_L]), each graph reads its own static copy and writes a static
    output.  B = 1: `Transducer.decode` (one utterance); B > 1: `Transducer.decode_batch` (every utterance of a batch in lockstep: all
    histories have the same length).  Graphs are captured lazily, share one memory pool and are dropped when the label encoder's
    parameter storage or the precision mode changes."""
    MAX_L = 128

    def __init__(self, model, device, batch=1):
        self.decoder, self.device, self.batch = model.decoder, device, batch
        self.master = torch.zeros(batch, self.MAX_L, dtype=torch.long, device=device)
        self.stream = torch.cuda.Stream(device)
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs = {}
        self.key = self.weights_key(model)
        # every captured graph bakes in the address of this stream's scratch arena (ttmi.ops.scratch), which is re-allocated when a
        # longer history needs more: size it for MAX_L before the first capture, and drop the graphs should it ever move all the same
        cur = torch.cuda.current_stream(device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.decoder(torch.zeros(batch, self.MAX_L, dtype=torch.long, device=device))
        cur.wait_stream(self.stream)
        self.arena = ops.scratch_generation(device, self.stream)

    @staticmethod
    def weights_key(model):
        return tuple(p.data_ptr() for p in model.decoder.parameters()) + (default_precision(),)

    def set_token(self, pos, tok):
        self.master[0, pos] = tok                                   # a device fill, no synchronisation

    def state(self, L):
        """label-encoder output at the last position of master[:, :L] -> [B, 1, d] (static buffer of graph L)"""
        if ops.scratch_generation(self.device, self.stream) != self.arena:      # the arena moved: every graph points at freed memory
            self.graphs.clear()
            self.arena = ops.scratch_generation(self.device, self.stream)
        entry = self.graphs.get(L)
        if entry is None:
            tok = self.master[:, :L].clone()
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                for _ in range(2):                                  # warm-up outside the capture: scratch arenas, kernel attributes
                    self.decoder(tok)
            cur.wait_stream(self.stream)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=self.pool, stream=self.stream):
                out = self.decoder(tok)[:, -1:, :].clone()       # (keeps [B, 1, d] alive per graph, not the stack's whole [B, L, d] output)
            assert ops.scratch_generation(self.device, self.stream) == self.arena, "scratch arena grew during a capture"
            entry = self.graphs[L] = (graph, tok, out)
        graph, tok, out = entry
        tok.copy_(self.master[:, :L])
        graph.replay()
        return out


class Transducer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder = BuildEncoder(config)
        self.decoder = BuildDecoder(config)
        self.joint = JointNet(input_size=config.joint.input_size, inner_dim=config.joint.inner_size,
                              vocab_size=config.vocab_size)
        if config.share_embedding:
            # the reference's branch dereferences self.decoder.embedding, which does not exist (tt/model.py:53-56)
            raise AttributeError("'BuildDecoder' object has no attribute 'embedding' (share_embedding is broken upstream)")

    def forward(self, inputs, targets):
        """-> logits [B, T, U+1, V] (tt/model.py:58-68).  In the bf16 pipeline the return value is a DeferredLogits handle on the two
        encoder states: `RNNTLoss` (train.py:53) consumes it through the fused joint + loss kernels, anything else makes it produce
        the real logits first - train.py:51-53 runs unchanged and on the fast path (deferred_logits_enabled for the switches)."""
        enc_state, dec_state = self._encode(inputs, targets)
        prec = default_precision()
        if enc_state.is_cuda and deferred_logits_enabled(self.config, prec):
            return DeferredLogits(self.joint, enc_state, dec_state, prec)
        return self.joint(enc_state, dec_state)

    def loss(self, inputs, inputs_length, targets, targets_length, reduction="mean", chunk=None, check_lengths=True, exp_domain=False):
        """Opt-in fused form of train.py:51-53 (`logits = model(inputs, targets); loss = criterion(logits, targets.int(),
        inputs_length.int(), targets_length.int())`) that never materialises the logits (API precedent: tt_espnet/model.py:35-81 returns
        the loss from forward).  Same numbers as the two-call form: the same kernels run, one chunk of `chunk` utterances at a time
        (default: about 2 GB of logits per chunk, adjusted to a row count the persistent wgrad kernel takes), and the chunk's buffer is overwritten by its gradient and consumed by the
        joint's backward before the next chunk starts.  Returns the loss ([1] for 'mean' / 'sum', [B] for 'none').

        exp_domain=True (bf16 mode, training-sized chunks; silently the form above otherwise): the projection stores exp(logit - shift) and
        per-row sums, so the loss never walks the lattice's rows and its gradient is consumed in factored form (include/ttmi.h).  Same
        loss and gradients up to bf16 rounding of different intermediates (tests/test_fused_loss_gpu.py states the tolerance).  The first
        call on a module (and the first after its joint weights were replaced) runs the form above and seeds the shift on the device
        (_ExpShift): no host synchronisation."""
        from warprnnt_pytorch import check_lengths as certify
        enc_state, dec_state = self._encode(inputs, targets)
        B, T, U1 = enc_state.shape[0], enc_state.shape[1], dec_state.shape[1]
        labels, al, ll = (t.to(device=enc_state.device, dtype=torch.int32).contiguous() for t in (targets, inputs_length, targets_length))
        certify(labels, al, ll, B, T, U1, check_lengths)
        ops.weights_fresh()
        prec = default_precision()
        if chunk is None:
            chunk = self.default_loss_chunk(B, T, U1, exp_domain)
        j = self.joint
        return _JointLossFn.apply(enc_state, dec_state, j.forward_layer.weight, j.forward_layer.bias, j.project_layer.weight,
                                  j.project_layer.bias, labels, al, ll, prec, int(chunk), reduction,
                                  j.exp_shift_state(enc_state.device) if exp_domain and prec == 1 else None, torch.is_grad_enabled())

    def default_loss_chunk(self, B, T, U1, exp_domain=False):
        """utterances per chunk of `loss()` (JointNet.default_loss_chunk)"""
        return self.joint.default_loss_chunk(B, T, U1, exp_domain)

    def _encode(self, inputs, targets):
        """both encoders of forward() (tt/model.py:58-65): -> (enc_state [B,T,d], dec_state [B,U+1,d])"""
        targets = F.pad(targets, pad=[1, 0, 0, 0], value=0)                 # leading blank / SOS
        audio_mask = self._audio_mask(inputs)
        if self.config.overlap_label_encoder and inputs.is_cuda:
            # the label encoder (tiny, launch-latency-bound kernels) is independent of the audio encoder until the joint:
            # run it on a side stream so both fill the chip together; autograd replays the same streams in backward
            main, side = torch.cuda.current_stream(inputs.device), ops.side_stream(inputs.device)
            side.wait_stream(main)                                          # the side stream needs `targets`, nothing else
            # host order: the audio encoder's long kernels are queued FIRST, the label encoder's ~400 tiny launches are issued while the
            # GPU is busy with them (issued first, they left the chip nearly idle for the ~1 ms the host needs to enqueue them)
            enc_state = self.encoder(inputs, audio_mask)
            # `targets` lives in the main stream's pool but is read by side-stream kernels - in backward as late as the embedding
            # gradient, the label encoder's last launch: without this its block could be handed to a main-stream allocation (and
            # overwritten) as soon as autograd drops the graph, while that launch is still queued
            capturing = torch.cuda.is_current_stream_capturing()    # (a captured step's tensors live in the graph's own pool for its lifetime)
            if not capturing:
                targets.record_stream(side)
            with torch.cuda.stream(side):
                dec_state = self._label_states(targets)
            main.wait_stream(side)
            if not capturing:
                dec_state.record_stream(main)
        else:
            enc_state = self.encoder(inputs, audio_mask)
            dec_state = self._label_states(targets)
        return enc_state, dec_state

    def _label_states(self, targets):
        """the label encoder on the padded targets (mask == look_ahead_mask(targets)[:, :, None]).  With TTMI_LABEL_VALUE_PRECISION (tt.transformer.label_value_precision) the
        states' VALUE comes from a second, gradient-free pass in a parity mode, on the same dropout masks (the CPU generator the seeds are drawn from is rewound for it), and
        the gradient flows through the bf16 pass: one label state meets all T frames of its utterance in the joint, so the label encoder's bf16 rounding is one pattern in T
        lattice rows - and, while its outputs still resemble each other (the first steps of training), in every utterance of the batch (round 6)."""
        vp = label_value_precision()
        if vp is None or not targets.is_cuda:
            return self.decoder(targets, MaskSpec(1))
        rng = torch.get_rng_state()
        dec = self.decoder(targets, MaskSpec(1))
        after = torch.get_rng_state()
        torch.set_rng_state(rng)
        with torch.no_grad():
            hi = self.decoder(targets, MaskSpec(1), prec=vp)
        torch.set_rng_state(after)
        return dec + (hi - dec.detach())

    def _audio_mask(self, inputs):
        """Reference behaviour is audio_mask=None (tt/model.py:60-61).  Opt-in `config.streaming` (absent in the
        reference YAMLs => None => off): {'left': l, 'right': r} band mask or {'chunk': c, 'left': l} block mask."""
        s = self.config.streaming
        if not s:
            return None
        if "chunk" in s:
            # per-row key intervals of the block mask (tt.utils.chunk_mask), built once per (T, chunk, left, device) and handed to the kernels
            # as they are: no [T, T] tensor per forward and none of as_mask_spec's host synchronisations (they made the C4-chunk run host-bound)
            chunk, left, T = int(s["chunk"]), int(s.get("left", 0)), inputs.size(1)
            key = (T, chunk, left, str(inputs.device))
            cache = self.__dict__.setdefault("_chunk_specs", {})
            spec = cache.get(key)
            if spec is None:
                i = torch.arange(T, device=inputs.device)
                lo = ((i // chunk) * chunk - left).clamp_(min=0)
                hi = ((i // chunk + 1) * chunk - 1).clamp_(max=T - 1)
                spec = cache[key] = MaskSpec(4, left=left + chunk - 1, right=chunk - 1,
                                             tensor=torch.stack([lo, hi], -1).to(torch.int32)[None].contiguous())
            return spec
        return MaskSpec(2, left=int(s.get("left", 0)), right=int(s.get("right", 0)))

    @torch.no_grad()
    def decode(self, enc_state, lengths, block=64):
        """Greedy: <= 1 symbol per frame, label encoder re-run on the whole history WITHOUT look-ahead mask (tt/model.py:70-90).
        Same token sequence as the reference's per-frame loop, but frames are scored `block` at a time against the current
        label state and the first non-blank frame is found on the device (ttmi_greedy_scan): one host sync per emitted
        symbol instead of one per frame.  The label-encoder re-runs replay captured graphs (one per history length, see
        _LabelStateGraphs)."""
        token_list = [0]
        dev = enc_state.device
        T = int(lengths)
        graphs = self._label_state_graphs(dev)

        def label_state():
            """label-encoder output for the current history -> [1, 1, d]"""
            L = len(token_list)
            if graphs is not None and L <= graphs.MAX_L:
                graphs.set_token(L - 1, token_list[-1])
                return graphs.state(L)
            return self.decoder(torch.tensor([token_list], dtype=torch.long, device=dev))[:, -1:, :]

        dec_state = label_state()
        t = 0
        while t < T:
            n = min(block, T - t)
            logits = self.joint(enc_state[t:t + n].unsqueeze(0), dec_state)                                  # [1, n, 1, V]
            row, tok = ops.greedy_scan(logits[0, :, 0, :])
            if tok is None:
                t += n
                continue
            if tok >= self.config.vocab_size:   # NaN logits (bad weights / inputs) come back as an impossible symbol: fail here, not later
                raise RuntimeError("greedy decode: the joint produced no finite maximum at frame %d (NaN logits?)" % (t + row))
            token_list.append(tok)
            dec_state = label_state()
            t += row + 1                                                    # the emitting frame is consumed
        return token_list[1:]

    def _label_state_graphs(self, dev, batch=1):
        """the per-model graph caches of `decode` / `decode_batch` (None: CPU tensors, or switched off with config.decode_graphs = False)"""
        if dev.type != "cuda" or self.config.decode_graphs is False:
            return None
        cache = self.__dict__.setdefault("_decode_graphs", {})
        g = cache.get(batch)
        if g is None or g.device != dev or g.key != _LabelStateGraphs.weights_key(self):
            if len(cache) > 4:
                cache.clear()                                       # (batch sizes come and go: keep a handful of graph sets)
            g = cache[batch] = _LabelStateGraphs(self, dev, batch)
        return g

    @torch.no_grad()
    def decode_batch(self, enc_states, lengths, block=None):
        """Greedy decoding of EVERY utterance of a batch at once: the token lists `decode(enc_states[b], lengths[b])` returns, for all b
        (tt/model.py:92-108 loops over the utterances, one host round trip per frame each).  The batch advances in lockstep over SYMBOL
        steps: in step s every utterance still decoding looks for its next non-blank frame (blocks of `block` frames from its own position,
        scored against its own label state: one joint call for the whole batch, ttmi_greedy_scan_batch / ttmi_greedy_advance keep positions,
        histories and flags on the device), then ONE label-encoder call of length s + 1 re-computes all label states (every history has
        exactly s + 1 tokens - the relative-position term depends on the sequence length, so utterances of different history lengths
        could not share a call).  Host synchronisations: one 8-byte read per scanned block of the whole batch (about one per symbol step)
        instead of one per symbol and utterance; label-encoder launches: one call per symbol step instead of one per symbol and
        utterance.  Utterances that have run out of frames LEAVE the batch (no extra host round trip: the count of the living comes with the
        flags, the rows from a stable sort on the device), so the long tail of a batch - the longest hypothesis of 32 synthetic utterances
        has 105 symbols, the mean 50 - runs on a handful of rows instead of 32.  Same arithmetic per utterance as `decode`: same tokens.
        block = frames scored per joint call (default 64)."""
        dev = enc_states.device
        B, T = enc_states.shape[0], enc_states.shape[1]
        block = 64 if block is None else block                       # frames scored per joint call
        T_len = torch.as_tensor(lengths, dtype=torch.int32).to(dev).clamp(max=T).contiguous()
        # Label-encoder graphs serve the symbol steps in which the batch is still COMPLETE (one replay instead of ~150 launches; a graph is
        # captured for a fixed row count); as soon as an utterance has finished the rows shrink and the calls are eager launches on the rows
        # still decoding - which is worth more than the replay (96 utt/s with a fixed batch of 32, 212 with shrinking rows).
        # config.decode_batch_graphs = False: eager launches throughout.  Round 4 had made the graphs opt-in because a third of the processes
        # came back from their first pure-replay pass with other tokens; round 5 found the cause - a hipMemset2DAsync NODE in the captured
        # label encoder (column 0 of the fp32 path's position slab) that ROCm 7.2's graph launch does not order against the kernels around
        # it - and removed every memset node from the library (csrc/rowops.hip fill_zero*, DESIGN.md section 4j,
        # tests/test_decode_graphs_gpu.py).
        graphs = self._label_state_graphs(dev, B) if self.config.decode_batch_graphs is not False else None
        final_hist = torch.zeros(B, T + 2, dtype=torch.long, device=dev)   # column 0 = the start symbol (blank); at most one symbol per frame
        final_count = torch.zeros(B, dtype=torch.int32, device=dev)
        hist = final_hist.clone()
        orig = torch.arange(B, device=dev)                           # the utterance each row of the (shrinking) batch belongs to
        t = torch.zeros(B, dtype=torch.int32, device=dev)
        need = torch.ones(B, dtype=torch.int32, device=dev)
        done = torch.zeros(B, dtype=torch.int32, device=dev)
        count = torch.zeros(B, dtype=torch.int32, device=dev)
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        n_frames = block
        key = torch.full((B,), n_frames << 32, dtype=torch.int64, device=dev)
        rows = torch.arange(n_frames, device=dev, dtype=torch.long)[None, :]

        def label_states(n_hist):
            """label-encoder outputs at the last position of every history (all of length n_hist) -> [rows of the batch, 1, d]"""
            if graphs is not None and n_hist <= graphs.MAX_L and hist.shape[0] == graphs.batch:
                graphs.master[:, :n_hist].copy_(hist[:, :n_hist])
                return graphs.state(n_hist)
            return self.decoder(hist[:, :n_hist].contiguous())[:, -1:, :]

        n_hist = 1
        dec_state = label_states(n_hist)
        while True:
            torch.sub(1, done, out=need)
            while True:
                idx = (t.long()[:, None] + rows).clamp_(max=T - 1)                             # frames t_b .. t_b + n_frames - 1 (beyond T_b: ignored by the scan)
                logits = self.joint(enc_states[orig[:, None], idx], dec_state)                    # [rows, n_frames, 1, V]
                ops.greedy_scan_batch(logits[:, :, 0, :], t, T_len, need, key)
                ops.greedy_advance(key, n_frames, n_hist, hist, t, T_len, need, done, count, flags)
                pending, alive = flags.tolist()                                                   # the batch's one host round trip per block
                if pending == 0:
                    break
            if alive == 0:
                break
            if alive < hist.shape[0] and self.config.decode_batch_shrink is not False:
                # finished utterances leave the batch: every later label-encoder and joint call runs on the rows still decoding (per-utterance
                # arithmetic does not depend on who else is in the batch).  Their histories are kept; no host round trip - the row count is
                # `alive`, the rows are the first `alive` of a stable sort by the done flag
                final_hist.index_copy_(0, orig, hist)
                final_count.index_copy_(0, orig, count)
                keep = torch.argsort(done, stable=True)[:alive]
                hist, orig, t, T_len, count = hist[keep], orig[keep], t[keep].contiguous(), T_len[keep].contiguous(), count[keep].contiguous()
                need = torch.zeros(alive, dtype=torch.int32, device=dev)
                done = torch.zeros(alive, dtype=torch.int32, device=dev)
                key = torch.full((alive,), n_frames << 32, dtype=torch.int64, device=dev)
            n_hist += 1
            dec_state = label_states(n_hist)
        final_hist.index_copy_(0, orig, hist)
        final_count.index_copy_(0, orig, count)
        final = final_hist.cpu()
        counts = final_count.cpu().tolist()
        return [final[b, 1:1 + counts[b]].tolist() for b in range(B)]

    @torch.no_grad()
    def recognize(self, inputs, inputs_length=None, audio_mask=None):
        enc_states = self.encoder(inputs, audio_mask)
        if enc_states.is_cuda and inputs.size(0) > 1 and self.config.batched_decode is not False:
            return self.decode_batch(enc_states, inputs_length)
        return [self.decode(enc_states[b], inputs_length[b]) for b in range(inputs.size(0))]

    @torch.no_grad()
    def beam_search(self, enc_state, lengths, beam_width=5, block=64):
        """The reference's beam search (tt/model.py:110-179), quirks included - frames advance on the currently most probable hypothesis;
        expansion happens only when that hypothesis predicts a non-blank; on an expanding frame every hypothesis contributes its top
        `beam_width` non-blank symbols; child token lists persist across expansions (appended to, never re-seeded from their parents) - with
        the device doing the per-frame work (round 5; the round-2 restatement, one label-encoder call and one `.item()` per hypothesis and
        frame, is `_beam_search_per_frame`, selected with block = 0):
          * every hypothesis has the same length (each expansion appends one symbol to every child list), so the `beam_width` label states
            come from ONE label-encoder call per expansion, and only change at expansions;
          * between expansions the lead hypothesis is fixed: `block` frames are scored against ITS label state in one joint call and the first
            non-blank frame is found on the device (ttmi_greedy_scan), as in `decode`;
          * on the expanding frame one joint call scores all hypotheses, softmax + top-(beam_width + 1) run on the device, and ONE host read
            brings the beam_width x (beam_width + 1) (probability, symbol) pairs back for the reference's own bookkeeping.
        Host reads: two per expansion + one per scanned block, instead of (1 + beam_width) per frame."""
        if not enc_state.is_cuda or not block:
            return self._beam_search_per_frame(enc_state, lengths, beam_width)
        dev, W, T = enc_state.device, beam_width, int(lengths)
        hyps = [[0] for _ in range(W)]
        score = np.zeros((W,), dtype=float)
        child = [[[0] for _ in range(W)] for _ in range(W)]
        child_score = np.zeros((W, W), dtype=float)
        first = True

        def label_states():
            """[W, d]: the label encoder (no look-ahead mask, tt/model.py:118) on the W histories, all of one length"""
            return self.decoder(torch.tensor(hyps, dtype=torch.long, device=dev))[:, -1, :]

        dec = label_states()
        t = 0
        while t < T:
            lead = int(score.argmax())
            n = min(block, T - t)
            logits = self.joint(enc_state[t:t + n].unsqueeze(0), dec[lead:lead + 1].unsqueeze(0))          # [1, n, 1, V]
            row, tok = ops.greedy_scan(logits[0, :, 0, :])
            if tok is None:                                     # the lead hypothesis predicts blank on all n frames
                t += n
                continue
            if tok >= self.config.vocab_size:
                raise RuntimeError("beam search: the joint produced no finite maximum at frame %d (NaN logits?)" % (t + row))
            te = t + row
            z = self.joint(enc_state[te:te + 1].unsqueeze(0), dec.unsqueeze(0))[0, 0]                     # [W, V]: every hypothesis on the expanding frame
            values, indices = torch.topk(F.softmax(z.float(), dim=-1), k=W + 1, dim=-1)
            both = torch.cat([values.double(), indices.double()], dim=1).cpu().tolist()                   # ONE host read per expansion
            for k in range(W):
                vals, idxs = both[k][:W + 1], [int(v) for v in both[k][W + 1:]]
                drop = idxs.index(0) if 0 in idxs else len(idxs) - 1
                idxs.pop(drop)
                vals.pop(drop)
                for i, sym in enumerate(idxs):
                    if first:
                        child[i][k].append(sym)
                    else:
                        child[k][i].append(sym)
                if first:
                    child_score[:, k] = np.log(vals)
                else:
                    child_score[k] = score[k] + np.log(vals)
            if first:
                first = False
                for i in range(W):
                    hyps[i] = copy.deepcopy(child[i][0])
                    score[i] = child_score[i, 0]
            else:
                best = heapq.nlargest(W, range(W ** 2), child_score.take)
                for i, idx in enumerate(best):
                    score[i] = child_score[idx // W, idx % W]
                    hyps[i] = copy.deepcopy(child[idx // W][idx % W])
            dec = label_states()
            t = te + 1
        return hyps[int(score.argmax())][1:]

    @torch.no_grad()
    def _beam_search_per_frame(self, enc_state, lengths, beam_width=5):
        """(round 2; `beam_search(block=0)`) Restatement of the reference's beam search (tt/model.py:110-179) including its quirks: frames advance on the
        currently most probable hypothesis; expansion happens only when that hypothesis predicts a non-blank; child
        token lists persist across expansions (they are appended to, never re-seeded from their parents)."""
        dev = enc_state.device

        def posterior(tokens, t):
            d = self.decoder(torch.tensor([tokens], dtype=torch.long, device=dev))[:, -1, :]
            return F.softmax(self.joint(enc_state[t].view(-1), d.view(-1)), dim=0)

        hyps = [[0] for _ in range(beam_width)]
        score = np.zeros((beam_width,), dtype=float)
        child = [[[0] for _ in range(beam_width)] for _ in range(beam_width)]
        child_score = np.zeros((beam_width, beam_width), dtype=float)
        first = True
        for t in range(int(lengths)):
            lead = int(score.argmax())
            if int(torch.argmax(posterior(hyps[lead], t)).item()) == 0:
                continue
            for k in range(beam_width):
                values, indices = torch.topk(posterior(hyps[k], t), k=beam_width + 1, dim=0)
                values, indices = values.tolist(), indices.tolist()
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

    @torch.no_grad()
    def recognize_beam_search(self, inputs, inputs_length, audio_mask=None):
        enc_states = self.encoder(inputs, audio_mask)
        return [self.beam_search(enc_states[b], inputs_length[b], beam_width=5) for b in range(inputs.size(0))]
