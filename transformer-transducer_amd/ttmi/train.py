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

    def zero_grad(self):
        self.grad.zero_()


class GradSync:
    """Bucketed SUM all-reduce of the flat gradient buffer, launched from autograd hooks.

    Buckets are contiguous slices of the flat buffer cut at parameter boundaries (~bucket_mb each).  Backward
    produces gradients roughly in reverse parameter order (joint first, encoder layer 0 last); a bucket is
    reduced asynchronously the moment all of its parameters have accumulated.  xGMI is a point-to-point mesh,
    so few large messages beat many small ones: default 32 MB buckets -> ~6 calls for the 193 MB payload."""

    def __init__(self, flat, bucket_mb=32, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets, self.bucket_of = [], {}
        limit = int(bucket_mb * (1 << 20) / 4)
        start, members = 0, []
        for i, p in enumerate(flat.params):
            members.append(i)
            end = flat.offsets[i + 1]
            if end - start >= limit or i == len(flat.params) - 1:
                for m in members:
                    self.bucket_of[m] = len(self.buckets)
                self.buckets.append((start, end, len(members)))
                start, members = end, []
        self.pending = [0] * len(self.buckets)
        self.seen = [False] * len(flat.params)
        self.works = []
        if self.world > 1:
            for i, p in enumerate(flat.params):
                hook = self._make_hook(i)
                p.register_post_accumulate_grad_hook(hook)      # gradients that arrive through autograd
                p._ttmi_on_grad = hook                           # gradients written in place by the HIP backward kernels

    def _make_hook(self, i):
        def hook(_param=None):
            if self.seen[i]:                # a parameter reports once per step, whichever path (in-place / autograd) is first
                return
            self.seen[i] = True
            b = self.bucket_of[i]
            self.pending[b] += 1
            if self.pending[b] == self.buckets[b][2]:
                s, e, _ = self.buckets[b]
                self.works.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return hook

    def start_step(self):
        self.pending = [0] * len(self.buckets)
        self.seen = [False] * len(self.flat.params)
        self.works = []

    def finish(self):
        """Wait for outstanding bucket reductions (stream-level wait, no host block on the GPU work);
        reduces any bucket whose hooks did not all fire (parameters unused in this step)."""
        ops.join_side_streams()             # label-encoder gradients are written in place on the side stream
        if self.world == 1:
            return
        for b, (s, e, n) in enumerate(self.buckets):
            if self.pending[b] != n:
                self.works.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self.works:
            w.wait()
        self.works = []


class FusedOptimizer:
    """SGD(momentum)/Adam on the flat buffers with clip_grad_norm_ folded in (tt/optim.py:57-73, train.py:62-65)."""

    def __init__(self, flat, kind="sgd", lr=0.00025, momentum=0.9, nesterov=False, weight_decay=0.0, betas=(0.9, 0.98),
                 eps=1e-8, max_grad_norm=200.0, world=1):
        self.flat, self.kind, self.lr, self.momentum, self.nesterov = flat, kind, lr, momentum, nesterov
        self.weight_decay, self.betas, self.eps, self.max_grad_norm, self.world = weight_decay, betas, eps, max_grad_norm, world
        self.state = [torch.zeros_like(flat.flat) for _ in range(2 if kind == "adam" else 1)]
        self.normsq = torch.zeros(1, dtype=torch.float32, device=flat.flat.device)
        self.global_step = 0

    def step(self):
        """gradients in flat.grad are SUMS over ranks; the 1/world averaging is folded into the update."""
        self.global_step += 1
        ops.join_side_streams()
        scale = 1.0 / self.world
        self.normsq.zero_()
        ops.sumsq(self.flat.grad, self.normsq)
        if self.kind == "adam":
            ops.adam_step(self.flat.flat, self.flat.grad, self.state[0], self.state[1], self.lr, self.betas, self.eps,
                          self.weight_decay, self.global_step, self.max_grad_norm, self.normsq, scale)
        else:
            ops.sgd_step(self.flat.flat, self.flat.grad, self.state[0], self.lr, self.momentum, self.weight_decay,
                         self.nesterov, self.max_grad_norm, self.normsq, scale)

    def grad_norm(self):
        return self.normsq.sqrt() / self.world
