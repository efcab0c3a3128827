"""`tt.optim.Optimizer(parameters, config)` with the reference's surface (tt/optim.py:4-33: step / epoch / zero_grad / state_dict /
load_state_dict / decay_lr, attributes lr, global_step, current_epoch, decay_ratio) on the MI355X path: the parameters are re-pointed
at one flat f32 buffer (`ttmi.train.FlatModel`) and `step()` is one fused HIP kernel.  `train.py` keeps calling
`torch.nn.utils.clip_grad_norm_` itself (train.py:62-63), so no clipping is folded in here; `ttmi.train.FusedOptimizer` used directly
(bench.py) folds the clip into the update.  All three types build_optimizer knows (tt/optim.py:57-84): sgd, adam, adadelta."""
from ttmi.train import FlatModel, FusedOptimizer


class _Params:
    def __init__(self, parameters):
        self._p = [p for p in parameters]

    def parameters(self):
        return iter(self._p)


class Optimizer(FusedOptimizer):
    def __init__(self, parameters, config):
        self.config = config
        if config.type not in ("sgd", "adam", "adadelta"):
            raise NotImplementedError                       # as build_optimizer does (tt/optim.py:83-84)
        flat = FlatModel(_Params(parameters))
        extra = dict(rho=0.9 if config.rho is None else config.rho, eps=config.eps) if config.type == "adadelta" else {}
        super().__init__(flat, kind=config.type, lr=config.lr, momentum=config.momentum or 0.0, nesterov=bool(config.nesterov),
                         weight_decay=config.weight_decay or 0.0, max_grad_norm=0.0, decay_ratio=config.decay_ratio, **extra)
        self.optimizer = self                   # the reference exposes the wrapped torch optimizer under this name
        self.epoch_decay_flag = False

    @property
    def param_groups(self):
        return self.state_dict()["param_groups"]
