"""Fused clip_grad_norm_ + AdamW over a root module's flat parameter buffer
(reference: torch.optim.AdamW(lr, betas=(beta1, 0.999)) + nn.utils.clip_grad_norm_,
/root/reference/mggan/abstract_train.py:45-50, model/train.py:131-135,209-213,656-658)
and the per-epoch cosine schedule (abstract_train.py:52-57,199-200)."""
import ctypes
import math

import torch

from mggan.hip.lib import lib


class FlatAdamW:
    def __init__(self, root, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.root = root.ensure_flat()
        self.base_lr = self._lr = lr
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        f = root._flat
        self.exp_avg = torch.zeros_like(f)
        self.exp_avg_sq = torch.zeros_like(f)
        self.nseg = len(root._flat_items)
        self.seg_step = torch.zeros(self.nseg, dtype=torch.int32, device=f.device)
        # (mggan_clip_adamw: 256 partial sums + the two counters of its grid barrier; zero once, the launches keep it ready)
        self._ws = torch.zeros(600, dtype=torch.float64, device=f.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=f.device)
        self._flat_id = f.data_ptr()
        # the learning rate the kernel reads lives on the device: a captured iteration follows the schedule
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float64, device=f.device)

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        self._lr_dev.fill_(self._lr)

    def zero_grad(self):
        self.root.zero_grad_flat()

    def step(self, max_norm=0.0, zero_grad=False, exchange=None):
        """Clip the gradients of the touched parameters to `max_norm` (0 = no clipping), then AdamW-update them.
        Parameters that received no gradient since zero_grad() are skipped, like `p.grad is None` in torch.
        zero_grad=True leaves the consumed gradients at zero (instead of their clipped values), which turns the
        next zero_grad() into bookkeeping only.
        exchange (sharded training, DistContext.all_reduce_grads(defer=True)): (comm, tail descriptor, keep-alive) -- the
        gradient all-reduce over the peer-mapped arenas and / or the finalize of the conv1 gradient tail run inside this
        launch."""
        r = self.root
        if r._flat.data_ptr() != self._flat_id:
            raise RuntimeError("the module's flat parameter buffer was rebuilt after the optimizer was created")
        if not r.flat_is_current_full():
            raise RuntimeError("a parameter of the module was re-seated outside its flat buffer (p.data = ..., a replaced "
                               "nn.Parameter, load_state_dict(assign=True)): call flatten_parameters_() and rebuild the "
                               "optimizer")
        from mggan.hip.functions import join_side_stream

        join_side_stream()  # weight-gradient GEMMs issued on the side stream must have landed
        r.adopt_foreign_grads()  # gradients that arrived through plain torch autograd count as touched too
        mask = r.touched_mask()
        from mggan.hip.functions import _s

        st = _s()
        lib.mggan_clip_adamw(r._flat.data_ptr(), r._flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                             self.exp_avg_sq.data_ptr(), r._flat.numel(), r._elem_seg.data_ptr(), self.nseg,
                             mask.data_ptr(), self.seg_step.data_ptr(), float(max_norm), float(self.lr),
                             self._lr_dev.data_ptr(), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                             1 if zero_grad else 0, self._ws.data_ptr(), self.grad_norm.data_ptr(),
                             exchange[0] if exchange else 0,
                             ctypes.addressof(exchange[1]) if (exchange and exchange[1] is not None) else 0, st)
        if exchange and exchange[2] is not None and torch.cuda.is_current_stream_capturing():
            self._keep = exchange[2]  # (graph memory: the tail's tensors are read by every replay)
        from mggan.hip.functions import bump_weight_version

        bump_weight_version(r, tuple(r._touched))  # folded LSTM weights of the updated parameters are stale from here on
        if zero_grad:
            r._grad_clean = True  # every gradient written since the last memset has been consumed and zeroed

    # torch.optim-compatible checkpoint surface (abstract_train.py:235-244 saves optimizer state_dicts)
    def state_dict(self):
        state = {}
        steps = self.seg_step.cpu()
        wm = getattr(self.root, "_width_map", None)  # narrower model on padded kernels: the reference's shapes (model/widths.py)
        for i, (p, o) in enumerate(self.root._flat_items):
            if int(steps[i]) == 0:
                continue
            n = p.numel()
            m, v = self.exp_avg[o:o + n].view(p.shape).clone(), self.exp_avg_sq[o:o + n].view(p.shape).clone()
            if wm is not None:
                name = self.root._flat_names[i]
                m, v = wm.to_logical(name, m), wm.to_logical(name, v)
            state[i] = {"step": torch.tensor(float(steps[i])), "exp_avg": m, "exp_avg_sq": v}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "params": list(range(self.nseg))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        steps = torch.zeros(self.nseg, dtype=torch.int32)
        wm = getattr(self.root, "_width_map", None)
        for i, st in sd["state"].items():
            p, o = self.root._flat_items[int(i)]
            n = p.numel()
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if wm is not None:
                name = self.root._flat_names[int(i)]
                m, v = wm.to_physical(name, m, p.shape), wm.to_physical(name, v, p.shape)
            self.exp_avg[o:o + n].copy_(m.reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(v.reshape(-1))
            steps[int(i)] = int(st["step"])
        self.seg_step.copy_(steps)
        self.lr = sd["param_groups"][0]["lr"]


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max, eta_min=0), stepped once per epoch."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.opt, self.T_max, self.eta_min, self.epoch = optimizer, T_max, eta_min, 0

    def step(self):
        self.epoch += 1
        base = self.opt.base_lr
        self.opt.lr = self.eta_min + (base - self.eta_min) * (1 + math.cos(math.pi * self.epoch / self.T_max)) / 2
