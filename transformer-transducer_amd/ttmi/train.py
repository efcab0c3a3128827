"""Data-parallel training-step plumbing for the MI355X path (replaces nn.DataParallel + clip_grad_norm_ +
optimizer.step of train.py:55-65,214-219 - SURVEY.md §8a A11, §8e).

One process per GPU.  All parameters live in ONE flat f32 buffer and all gradients in another, so that
  * the gradient all-reduce is a handful of large bucketed RCCL calls over xGMI, launched from
    post-accumulate hooks as soon as a bucket's last gradient lands (overlapping the rest of backward),
  * the global grad-norm, the clip and the SGD/Adam update are three streaming HIP kernels with no host sync.
RNNTLoss(reduction='mean') divides by the LOCAL batch, so averaging gradients over equal-sized ranks
reproduces the global-batch mean.
"""
import torch
import torch.distributed as dist

from . import ops


class FlatModel:
    """Re-points every parameter (and its .grad) of `model` at views into two flat f32 buffers."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]          # keep every view 16-byte aligned
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        n = self.offsets[-1]
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            v = self.flat[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view_as(p)
            p._ttmi_direct = True           # the HIP backward kernels accumulate straight into p.grad (tt.transformer.grad_targets)
        self.numel = n
        self.shadow = None

    def zero_grad(self):
        if ops.wgrad_queue is not None:
            ops.wgrad_queue.discard()          # leftovers of an aborted backward pass
        self.grad.zero_()

    def enable_grouped_wgrads(self, layers_per_group=4, immediate_first_layer=False):
        """Defer the weight-gradient GEMMs of the audio-sized encoder layers (bf16 pipeline) and launch them `layers_per_group` layers at
        a time: 4 layers x 64 tiles fill the chip with one tile per CU over the whole reduction, so the K-splits and their f32 atomics go
        away and the gradients are bit-identical from run to run and across ranks (include/ttmi.h, ttmi_wgrad_group).  The gradients of
        the deferred weights appear when their group runs - at the latest when the first layer's backward pass ends; gradient-ready
        hooks (GradSync) fire then.  immediate_first_layer (data-parallel runs): the first layer keeps its own launches, so that the
        last gradients of the step are no larger than before.  Process-wide switch: the queue lives in ttmi.ops."""
        ops.wgrad_queue = ops.WgradQueue(layers_per_group, immediate_first_layer)
        return self

    def disable_grouped_wgrads(self):
        ops.wgrad_queue = None

    # ---- bf16 shadows of the GEMM weights (include/ttmi.h: ttmi_weight_shadow_*)
    def enable_shadows(self):
        """Keep a plain and a transposed bf16 copy of every 2-D GEMM weight, rebuilt by ONE launch after each optimiser step, and
        register them with the library: the bf16 pipeline's forward / backward calls then skip their per-call weight conversions
        (118 launches per step at C2).  Freshness: FusedOptimizer.step refreshes after its update; in-place changes made through the
        parameter itself (load_state_dict, `with torch.no_grad(): p.mul_(..)`) bump its version counter and are caught by `ensure_fresh()`
        at the next encoder / decoder / joint call; after writing through `p.data` or the flat buffer directly, call refresh_shadows()."""
        import weakref
        dev = self.flat.device
        if dev.type != "cuda":
            raise ValueError("bf16 weight shadows need the parameters on the GPU")
        ents = [(p, o) for p, o in zip(self.params, self.offsets) if p.dim() == 2 and p.shape[1] % 8 == 0 and p.shape[0] >= 64]
        n_plain = sum((p.numel() + 7) // 8 * 8 for p, _ in ents)
        plain = torch.zeros(2 * n_plain, dtype=torch.bfloat16, device=dev)        # [bf16(w) of every weight | bf16(w - bf16(w)) of every weight]: the second halves serve option 13
        ldts = [(p.shape[0] + 63) // 64 * 64 if p.shape[0] % 8 else p.shape[0] for p, _ in ents]
        trans = torch.zeros(sum((p.shape[1] * ld + 7) // 8 * 8 for (p, _), ld in zip(ents, ldts)), dtype=torch.bfloat16, device=dev)
        rows, po, to, tile0 = [], 0, 0, 0
        L = ops.lib()
        for (p, o), ld in zip(ents, ldts):
            R, C = p.shape
            w16, wT16 = plain[po:po + R * C], trans[to:to + C * ld]
            tx, ty = (C + 31) // 32, (ld + 31) // 32
            rows.append([p.data_ptr(), R, C, wT16.data_ptr(), ld, w16.data_ptr(), tile0, tx])
            ops.check(L.ttmi_weight_shadow_register(ops.c_void_p(p.data_ptr()), ops.c_int(R), ops.c_int(C), ops.c_void_p(w16.data_ptr()),
                                                    ops.c_void_p(wT16.data_ptr()), ops.c_long(ld)), "ttmi_weight_shadow_register")
            ops.check(L.ttmi_weight_shadow_register_lo(ops.c_void_p(p.data_ptr()), ops.c_void_p(w16.data_ptr() + 2 * n_plain)), "ttmi_weight_shadow_register_lo")
            po += (R * C + 7) // 8 * 8
            to += (C * ld + 7) // 8 * 8
            tile0 += tx * ty
        self.shadow = dict(plain=plain, trans=trans, table=torch.tensor(rows, dtype=torch.int64, device=dev), n=len(rows), tiles=tile0,
                           version=-1, ptrs=[r[0] for r in rows], lo_delta=n_plain)
        _shadowed.append(weakref.ref(self))
        self.refresh_shadows()
        return self

    def refresh_shadows(self):
        sh = self.shadow
        if sh is None:
            return
        ops.check(ops.lib().ttmi_weight_shadow_refresh_lo(ops.c_void_p(sh["table"].data_ptr()), ops.c_int(sh["n"]), ops.c_long(sh["tiles"]),
                                                          ops.c_long(sh["lo_delta"]), ops._stream()), "ttmi_weight_shadow_refresh_lo")
        sh["version"] = self._versions()

    def _versions(self):
        return sum(p._version for p in self.params)

    def disable_shadows(self):
        if self.shadow is not None:
            for ptr in self.shadow["ptrs"]:
                ops.lib().ttmi_weight_shadow_clear(ops.c_void_p(ptr))
            self.shadow = None

    def __del__(self):
        try:
            self.disable_shadows()
        except Exception:
            pass


_shadowed = []      # weak references to FlatModels with live shadows


def ensure_fresh():
    """called by the sub-layer autograd functions before they hand weights to the library: a parameter changed through torch since the
    last refresh (the flat buffer's version counter moved) gets its shadows rebuilt first"""
    for ref in _shadowed[:]:
        fm = ref()
        if fm is None or fm.shadow is None:
            _shadowed.remove(ref)
        elif fm._versions() != fm.shadow["version"]:
            fm.refresh_shadows()


class GradSync:
    """Bucketed SUM all-reduce of the flat gradient buffer, launched from autograd hooks.

    Buckets are contiguous slices of the flat buffer cut at parameter boundaries (~bucket_mb each).  Backward
    produces gradients roughly in reverse parameter order (joint first, encoder layer 0 last); a bucket is
    reduced asynchronously the moment all of its parameters have accumulated.  xGMI is a point-to-point mesh,
    so few large messages beat many small ones: default 32 MB buckets -> ~6 calls for the 193 MB payload."""

    def __init__(self, flat, bucket_mb=32, group=None, tail_mb=None, tail_buckets=2, always_reduce=False, broadcast=True, auto_finish=False):
        """tail_mb: size of the first `tail_buckets` buckets (the parameters whose gradients arrive LAST - audio-encoder layer 0 first in
        parameter order): the all-reduce of the bucket that completes last cannot overlap anything, so it is kept small (default
        bucket_mb / 4).  broadcast (world > 1): rank 0's parameters are broadcast to every rank and the replicas' bit patterns are
        compared before the first step ("same seed OR rank-0 broadcast" - nn.DataParallel re-broadcasts every step,
        train.py:214-219; here once, and again on request after a checkpoint was loaded on one rank: `broadcast_parameters()`).
        auto_finish: for loops that cannot call `finish()` themselves (the reference's own `train()`, train.py:51-65, driven by
        ttmi.dp_train): the first gradient of a step queues an end-of-backward callback (autograd's queue_callback, as DDP does) that
        waits for the reductions and turns the SUMS into MEANS in place, so that whatever follows `loss.backward()` -
        `clip_grad_norm_(model.parameters(), ...)`, then `optimizer.step()` with world = 1 - sees the global-batch gradient."""
        self.flat, self.group = flat, group
        self.auto_finish, self._queued, self.finished = auto_finish, False, False
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets, self.bucket_of = [], {}
        tail_mb = bucket_mb / 4.0 if tail_mb is None else tail_mb
        start, members = 0, []
        for i, p in enumerate(flat.params):
            members.append(i)
            end = flat.offsets[i + 1]
            limit = int((tail_mb if len(self.buckets) < tail_buckets else bucket_mb) * (1 << 20) / 4)
            if end - start >= limit or i == len(flat.params) - 1:
                for m in members:
                    self.bucket_of[m] = len(self.buckets)
                self.buckets.append((start, end, len(members)))
                start, members = end, []
        self.pending = [0] * len(self.buckets)
        self.seen = [False] * len(flat.params)
        self.writers = [set() for _ in self.buckets]
        self.works = []
        self.cuda = flat.flat.is_cuda
        # every rank starts from the same torch seed (identical initial weights); its dropout masks must still differ
        from tt import transformer as _tr
        _tr.set_seed_salt(dist.get_rank(group) if self.world > 1 else 0)
        self.active = self.world > 1 or (always_reduce and dist.is_available() and dist.is_initialized())   # always_reduce: a one-rank group still
        if self.active:                                                                                          # issues its collectives (tests)
            for i, p in enumerate(flat.params):
                hook = self._make_hook(i)
                p.register_post_accumulate_grad_hook(hook)      # gradients that arrive through autograd
                p._ttmi_on_grad = hook                           # gradients written in place by the HIP backward kernels
        if self.world > 1 and broadcast:
            self.broadcast_parameters()

    # ---- replica consistency (one-off collectives, outside the step)
    def broadcast_parameters(self, src=0, optimizer=None, check=True):
        """every rank takes rank `src`'s parameters (ONE broadcast of the flat buffer: 193 MB at C2) and, when given, its optimiser state and
        counters (resume: load_checkpoint on rank 0 only, then this); bf16 weight shadows are rebuilt; `check` compares the replicas after"""
        if self.world <= 1:
            return
        dist.broadcast(self.flat.flat, src, group=self.group)
        if optimizer is not None:
            for buf in optimizer.state:
                dist.broadcast(buf, src, group=self.group)
            optimizer.dropped_steps()           # the source rank's step count is the DEVICE's (a dropped step consumes none): refresh the host mirror first
            meta = torch.tensor([optimizer.lr, float(optimizer.steps_taken), float(optimizer.global_step), float(optimizer.current_epoch)],
                                dtype=torch.float64, device=self.flat.flat.device)
            dist.broadcast(meta, src, group=self.group)
            lr, steps, gstep, epoch = meta.tolist()
            optimizer.lr, optimizer.steps_taken = lr, int(steps)
            optimizer.global_step, optimizer.current_epoch = int(gstep), int(epoch)
        if self.flat.shadow is not None:
            self.flat.refresh_shadows()
        if check:
            self.check_replicas()

    def replica_checksum(self):
        """order-independent-free 64-bit checksum of the parameters' BIT PATTERNS (two weighted integer sums): equal on every rank iff
        the replicas agree (up to a 2^-64-class collision); a float sum would hide sign / NaN differences"""
        bits32 = self.flat.flat.view(torch.int32)
        s0 = torch.zeros((), dtype=torch.int64, device=bits32.device)
        s1 = torch.zeros((), dtype=torch.int64, device=bits32.device)
        chunk = 1 << 22                     # 4 M elements at a time: three 32 MB temporaries instead of three of the buffer's size (1.2 GB at C2, 2 GB at C4)
        for off in range(0, bits32.numel(), chunk):
            b = bits32[off:off + chunk].to(torch.int64)
            w = torch.arange(off + 1, off + 1 + b.numel(), device=b.device, dtype=torch.int64) % 65521 + 1
            s0 += b.sum()
            s1 += (b * w).sum()
        return torch.stack([s0, s1])

    def check_replicas(self):
        """raises on EVERY rank when the ranks' parameters differ (call it any time between steps, e.g. once per epoch)"""
        if self.world <= 1:
            return True
        c = self.replica_checksum()
        lo, hi = c.clone(), c.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            raise RuntimeError("GradSync: the data-parallel replicas' parameters differ (rank %d checksum %s; min %s max %s over ranks): "
                               "a checkpoint loaded on one rank, or different seeds - call broadcast_parameters()"
                               % (dist.get_rank(self.group), c.tolist(), lo.tolist(), hi.tolist()))
        return True

    def _make_hook(self, i):
        def hook(_param=None):
            if self.seen[i]:                # a parameter reports once per step, whichever path (in-place / autograd) is first
                return
            self.seen[i] = True
            if self.auto_finish and not self._queued:
                self._queued = True
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            b = self.bucket_of[i]
            self.pending[b] += 1
            if self.cuda:
                # the kernels that wrote this gradient were queued on the CURRENT stream (main stream, or the label encoder's side
                # stream): remember every stream that wrote into the bucket
                self.writers[b].add(torch.cuda.current_stream(self.flat.flat.device))
            if self.pending[b] == self.buckets[b][2]:
                self._reduce(b)
        return hook

    def _reduce(self, b):
        """queue bucket b's all-reduce behind EVERY stream that wrote into it.  The collective orders itself only after the stream
        that is current when it is issued; a bucket can hold gradients written on the side stream (label encoder) and on the main
        stream (audio encoder), and the hook that completes it fires on just one of them."""
        s, e, _ = self.buckets[b]
        if self.cuda:
            cur = torch.cuda.current_stream(self.flat.flat.device)
            for st in self.writers[b]:
                if st != cur:
                    cur.wait_stream(st)
        self.works.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _end_of_backward(self):
        """auto_finish: runs when the backward pass that fired the first hook completes"""
        self.finish()
        if self.world > 1:
            self.flat.grad.mul_(1.0 / self.world)      # SUM -> mean over the ranks (each rank's loss is its local-batch mean)
        self.finished = True

    def start_step(self):
        self._queued, self.finished = False, False
        self.pending = [0] * len(self.buckets)
        self.seen = [False] * len(self.flat.params)
        self.writers = [set() for _ in self.buckets]
        self.works = []

    def finish(self):
        """Wait for outstanding bucket reductions (stream-level wait, no host block on the GPU work);
        reduces any bucket whose hooks did not all fire (parameters unused in this step)."""
        ops.wgrad_flush(every_stream=True)  # (normally a no-op: each encoder's first layer launches what its stream still holds)
        ops.join_side_streams()             # label-encoder gradients are written in place on the side stream
        if not self.active:
            return
        for b, (s, e, n) in enumerate(self.buckets):
            if self.pending[b] != n:
                self._reduce(b)                # (join_side_streams above already ordered the current stream behind the side stream)
        for w in self.works:
            w.wait()
        self.works = []


class FusedOptimizer:
    """SGD(momentum)/Adam on the flat buffers with clip_grad_norm_ folded in (tt/optim.py:57-73, train.py:62-65), plus the
    bookkeeping of the reference's `Optimizer` wrapper (tt/optim.py:4-33): `global_step` (starts at 1), `current_epoch`,
    `decay_lr()` (lr *= decay_ratio) and a `state_dict()` in torch.optim's own layout, so the 'optimizer' entry of a reference
    checkpoint (tt/utils.py:80-91) loads here and ours loads into torch.optim.SGD / Adam."""

    def __init__(self, flat, kind="sgd", lr=0.00025, momentum=0.9, nesterov=False, weight_decay=0.0, betas=(0.9, 0.98),
                 eps=None, max_grad_norm=200.0, world=1, decay_ratio=0.5, rho=0.9):
        if kind not in ("sgd", "adam", "adadelta"):
            raise NotImplementedError("FusedOptimizer: optimizer type %r (tt/optim.py:57-84 builds sgd, adam and adadelta)" % (kind,))
        if eps is None:
            eps = 1e-6 if kind == "adadelta" else 1e-8        # torch.optim's defaults
        self.flat, self.kind, self.momentum, self.nesterov = flat, kind, momentum, nesterov
        self.weight_decay, self.betas, self.eps, self.max_grad_norm, self.world = weight_decay, betas, eps, max_grad_norm, world
        self.decay_ratio, self.rho = decay_ratio, rho
        self.state = [torch.zeros_like(flat.flat) for _ in range(1 if kind == "sgd" else 2)]
        self.normsq = torch.zeros(1, dtype=torch.float32, device=flat.flat.device)
        # what changes between steps lives on the DEVICE, where the update kernels read it at run time: (learning rate, optimiser steps taken).
        # A step captured into a HIP graph (GraphedStep) would otherwise replay the values of the moment of capture - decay_lr() silently
        # ignored, Adam's bias correction frozen.  `lr` and `steps_taken` are host mirrors; assigning to them writes the device copy.
        # hyper[2] counts the steps the device DROPPED (non-finite gradient norm): those consume no step count (`dropped_steps()`).
        self.hyper = torch.zeros(3, dtype=torch.float32, device=flat.flat.device)      # {lr, steps taken, steps dropped}: include/ttmi.h, csrc/optim.hip
        self.lr = lr
        self.global_step = 1                # tt/optim.py:8
        self.current_epoch = 0
        self.steps_taken = 0                # Adam's bias-correction exponent (torch keeps it per parameter in state['step'])

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        self.hyper[0:1].fill_(self._lr)     # (a device fill on the current stream: ordered before the next step / replay issued on it)

    @property
    def steps_taken(self):
        return self._steps

    @steps_taken.setter
    def steps_taken(self, value):
        self._steps = int(value)
        self.hyper[1:2].fill_(float(self._steps))

    def dropped_steps(self):
        """steps the update kernels dropped because the gradient norm was not finite (device counter; reading it waits for the stream).
        A dropped step leaves parameters, state AND the device's step count untouched; the host mirror is re-read here."""
        h = self.hyper.tolist()
        self._steps = int(h[1])
        return int(h[2])

    def host_counters(self):
        return self.global_step, self._steps

    def set_host_counters(self, counters):
        """host mirrors only (the device count is advanced by the kernels themselves): GraphedStep around a capture / per replay"""
        self.global_step, self._steps = counters

    def step(self):
        """gradients in flat.grad are SUMS over ranks; the 1/world averaging is folded into the update.  A step whose gradient norm is
        not finite is dropped on the device (parameters and state untouched), with or without clipping."""
        flat = self.flat
        for p, o in zip(flat.params, flat.offsets):       # the kernels read flat.grad: a .grad that was re-pointed (zero_grad(set_to_none),
            if p.grad is None or p.grad.data_ptr() != flat.grad.data_ptr() + 4 * o:       # p.grad = ...) would be silently ignored
                raise RuntimeError("FusedOptimizer.step: a parameter's .grad no longer aliases the flat gradient buffer "
                                   "(use FlatModel.zero_grad(), not zero_grad(set_to_none=True))")
        self.global_step += 1
        self._steps += 1                    # (host mirror; the kernels advance hyper[1] themselves)
        ops.wgrad_flush(every_stream=True)                # grouped weight gradients still queued (none after a complete backward pass)
        ops.join_side_streams()
        scale = 1.0 / self.world
        max_norm = self.max_grad_norm or 0.0
        self.normsq.zero_()
        ops.sumsq(flat.grad, self.normsq)   # always: the norm is also what drops a NaN step (train.py's own clip_grad_norm_ leaves NaNs in place)
        if self.kind == "adam":
            ops.adam_step(flat.flat, flat.grad, self.state[0], self.state[1], self._lr, self.betas, self.eps,
                          self.weight_decay, self._steps, max_norm, self.normsq, scale, self.hyper)
        elif self.kind == "adadelta":
            ops.adadelta_step(flat.flat, flat.grad, self.state[0], self.state[1], self._lr, self.rho, self.eps, self.weight_decay,
                              max_norm, self.normsq, scale, self.hyper)
        else:
            ops.sgd_step(flat.flat, flat.grad, self.state[0], self._lr, self.momentum, self.weight_decay,
                         self.nesterov, max_norm, self.normsq, scale, self.hyper)
        flat.refresh_shadows()              # the kernels above changed the weights behind torch's back: rebuild the bf16 copies (one launch)

    def grad_norm(self):
        return self.normsq.sqrt() / self.world

    # ---- tt/optim.py:17-33
    def epoch(self):
        self.current_epoch += 1

    def zero_grad(self):
        self.flat.zero_grad()

    def decay_lr(self):
        self.lr *= self.decay_ratio

    # ---- checkpointing in torch.optim's layout (train.py:196-212 restores it with optimizer.load_state_dict)
    def _views(self, buf):
        f = self.flat
        return [buf[o:o + p.numel()].view_as(p) for p, o in zip(f.params, f.offsets)]

    def state_dict(self):
        n = len(self.flat.params)
        if self.flat.flat.is_cuda:
            self.dropped_steps()            # the step count that goes into the checkpoint is the device's (dropped steps do not count)
        if self.kind == "sgd":
            group = dict(lr=self.lr, momentum=self.momentum, dampening=0, weight_decay=self.weight_decay, nesterov=self.nesterov)
            state = {}
            if self.steps_taken > 0 and self.momentum != 0:
                state = {i: {"momentum_buffer": v.clone()} for i, v in enumerate(self._views(self.state[0]))}
        elif self.kind == "adadelta":
            group = dict(lr=self.lr, rho=self.rho, eps=self.eps, weight_decay=self.weight_decay)
            state = {}
            if self.steps_taken > 0:
                sq, acc = self._views(self.state[0]), self._views(self.state[1])
                state = {i: {"step": torch.tensor(float(self.steps_taken)), "square_avg": sq[i].clone(), "acc_delta": acc[i].clone()}
                         for i in range(n)}
        else:
            group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay, amsgrad=False)
            state = {}
            if self.steps_taken > 0:
                m, v = self._views(self.state[0]), self._views(self.state[1])
                state = {i: {"step": torch.tensor(float(self.steps_taken)), "exp_avg": m[i].clone(), "exp_avg_sq": v[i].clone()}
                         for i in range(n)}
        group["params"] = list(range(n))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.params):
            raise ValueError("FusedOptimizer.load_state_dict: expected one parameter group with %d parameters" % len(self.flat.params))
        g = groups[0]
        self.lr = g["lr"]
        self.weight_decay = g.get("weight_decay", self.weight_decay)
        if self.kind == "sgd":
            self.momentum, self.nesterov = g.get("momentum", self.momentum), g.get("nesterov", self.nesterov)
        elif self.kind == "adadelta":
            self.rho, self.eps = g.get("rho", self.rho), g.get("eps", self.eps)
        else:
            self.betas, self.eps = tuple(g.get("betas", self.betas)), g.get("eps", self.eps)
        for b in self.state:
            b.zero_()
        self.steps_taken = 0
        ids = g["params"]
        st = sd["state"]
        names = {"sgd": ("momentum_buffer",), "adam": ("exp_avg", "exp_avg_sq"), "adadelta": ("square_avg", "acc_delta")}[self.kind]
        for k, name in enumerate(names):
            for pos, view in enumerate(self._views(self.state[k])):
                ent = st.get(ids[pos], st.get(str(ids[pos])))
                if ent is not None and ent.get(name) is not None:
                    view.copy_(ent[name])
        steps = [int(e["step"]) for e in st.values() if "step" in e]
        if steps:
            if min(steps) != max(steps):
                raise ValueError("FusedOptimizer.load_state_dict: per-parameter Adam step counts differ; one flat update needs one count")
            self.steps_taken = steps[0]
        elif st:
            self.steps_taken = 1            # SGD: torch seeds the momentum buffer with the first gradient; ours has the same value after step 1


class GraphedStep:
    """A whole training step - forward, loss, backward, gradient all-reduce, clip + optimiser, shadow refresh - captured ONCE into a HIP
    graph and replayed with one launch per step: no Python, no ~400 kernel launches on the host (8 ranks of a data-parallel job share one
    host).

        step = GraphedStep(lambda: one_step(static_inputs...), optimizer=opt, exp_state=model.joint.exp_shift_state(dev))
        for batch in loader:  static_inputs.copy_(batch); loss = step()

    `step_fn` must be free of host synchronisation and must read its batch from tensors that stay at the same address.  What makes the
    step replayable here:
    * dropout seeds are drawn on the host per call and would be frozen in the graph, so every dropout site mixes in a device word (`salt`,
      ttmi_set_dropout_salt) that the graph bumps before anything else - each replay draws new masks, forward and backward of one replay
      agree;
    * the learning rate and the optimiser's step count (Adam's bias corrections) are DEVICE scalars the update kernels read at run time
      (FusedOptimizer.hyper): `decay_lr()` / `opt.lr = ...` between replays takes effect, all three optimiser kinds replay.  Other
      hyper-parameters (momentum, betas, weight decay, clip norm, world size) are by-value kernel arguments: GraphedStep snapshots them at
      capture and RAISES at the next call if one changed (re-capture with `recapture()`);
    * `GradSync`'s bucketed all-reduces are launched from autograd hooks inside the captured region and become graph nodes (RCCL
      collectives are capturable; every rank must capture and replay the same sequence); with a process group initialised the capture
      runs in `thread_local` error mode so that the process group's watchdog thread may keep polling its events;
    * the exp-domain loss form's range check (tt.model._ExpShift) is looked at between replays: after a raised flag (that replay's step was
      dropped on the device: NaN gradients) the NEXT call runs ONE eager step in place of a replay - the plain loss form, which re-seeds
      the shift - and returns its loss; the call after that captures again (a capture executes nothing) and replays.  The batch of
      the FLAGGED replay is lost (its step was dropped on the device: no update, `optimizer.dropped_steps()` counts it); every other
      batch gets exactly one optimiser update.
    Host-side counters (`optimizer.global_step`, its step-count mirror) are restored after a capture - which runs `step_fn`'s Python once
    without executing a step - and advanced by one per replay.  The `warmup` eager steps of the constructor ARE real steps on whatever the
    static input tensors hold (they update the weights and the counters); fill the static inputs with the first batch before constructing."""

    def __init__(self, step_fn, device=None, warmup=3, exp_state=None, on_replay=(), optimizer=None, capture_error_mode=None):
        self.step_fn, self.exp_state, self.on_replay, self.optimizer = step_fn, exp_state, tuple(on_replay), optimizer
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        if capture_error_mode is None:
            capture_error_mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        self.capture_error_mode = capture_error_mode
        self.salt = torch.zeros(1, dtype=torch.int32, device=self.device)
        ops.set_dropout_salt(self.salt)
        self.stream = torch.cuda.Stream(self.device)
        self.graph, self.out, self.captures, self.eager_steps = None, None, 0, 0
        self._warm(warmup)
        self._capture()

    def _baked(self):
        """by-value kernel arguments of the optimiser step: frozen in the graph"""
        o = self.optimizer
        if o is None:
            return None
        return (o.kind, o.momentum, o.nesterov, o.weight_decay, tuple(o.betas), o.eps, o.rho, o.max_grad_norm, o.world)

    def _eager(self):
        """one eager step on the capture stream (same streams, arenas and allocator pools as the captured one)"""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.salt.add_(1)
            self.out = self.step_fn()
        cur.wait_stream(self.stream)
        self.eager_steps += 1
        return self.out

    def _warm(self, n):
        """eager steps: scratch arenas, fork streams, kernel attributes, the exp form's shift"""
        for _ in range(n):
            self._eager()
        torch.cuda.synchronize(self.device)

    def _capture(self):
        if self.exp_state is not None and not self.exp_state.valid:
            raise RuntimeError("GraphedStep: the exp-domain loss form has no valid shift after the warm-up steps")
        counters = self.optimizer.host_counters() if self.optimizer is not None else None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode=self.capture_error_mode):
            self.salt.add_(1)
            self.out = self.step_fn()
        if counters is not None:
            self.optimizer.set_host_counters(counters)          # the capture ran step_fn's Python, not a step
        self.baked = self._baked()
        self.captures += 1

    def recapture(self):
        """capture again (after changing a by-value hyper-parameter of the optimiser, or the step function's behaviour)"""
        torch.cuda.synchronize(self.device)
        self.graph = None
        self._capture()

    def __call__(self):
        st = self.exp_state
        if st is not None:
            st.poll()
            if not st.valid:                # the previous replay was flagged (and dropped on the device): this batch gets ONE eager step -
                self.graph = None           # the plain form, which re-seeds the shift -, the next call captures again
                out = self._eager()
                return out
        if self.graph is None:
            self._capture()
        if self.baked != self._baked():
            raise RuntimeError("GraphedStep: an optimiser hyper-parameter that is a by-value kernel argument (kind, momentum, nesterov, "
                               "weight_decay, betas, eps, rho, max_grad_norm, world) changed since the capture; call recapture().  "
                               "(lr and the step count are read on the device and may change freely.)")
        self.graph.replay()
        if self.optimizer is not None:
            g, s = self.optimizer.host_counters()
            self.optimizer.set_host_counters((g + 1, s + 1))
        for cb in self.on_replay:
            cb()
        if st is not None:
            st.watch()
        return self.out


def save_checkpoint(model, optimizer, path, multi_gpu=False):
    """the reference's `.chkpt` layout (tt/utils.py:80-91): per-module state_dicts (un-prefixed keys), the optimizer's state_dict,
    'epoch' and 'step'.  Rank 0 writes it in data-parallel runs."""
    m = model.module if multi_gpu else model
    torch.save({"encoder": m.encoder.state_dict(), "decoder": m.decoder.state_dict(), "joint": m.joint.state_dict(),
                "optimizer": optimizer.state_dict(), "epoch": optimizer.current_epoch, "step": optimizer.global_step}, path)


def load_checkpoint(model, optimizer, path, mode="continue", map_location=None):
    """train.py:196-212 + 234-241: weights always; optimizer state, epoch and step only in 'continue' mode.  Parameters stay views of
    the flat buffer (load_state_dict copies in place)."""
    ck = torch.load(path, map_location=map_location)
    model.encoder.load_state_dict(ck["encoder"])
    model.decoder.load_state_dict(ck["decoder"])
    model.joint.load_state_dict(ck["joint"])
    if optimizer is not None and mode == "continue":
        optimizer.load_state_dict(ck["optimizer"])
        optimizer.global_step, optimizer.current_epoch = ck["step"], ck["epoch"]
    return ck
