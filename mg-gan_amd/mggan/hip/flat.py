"""Flat parameter / gradient storage for the HIP modules.

All parameters of a root module (generator or discriminator) live in ONE f32
buffer (nn.Parameters are views into it, state_dict keys unchanged), with a
twin gradient buffer.  This is what lets
  * clip_grad_norm_ + AdamW run as one fused launch over the whole model,
  * the per-generator decoder weights be addressed as base + g * stride,
  * a multi-GPU gradient all-reduce be ONE RCCL call per optimizer step.
Backward kernels accumulate parameter gradients straight into the twin buffer;
`p.grad` is attached as a view of it (zeroed on first touch after zero_grad()).
"""
import torch
from torch import nn

ALIGN = 4  # floats (16 B) -> every tensor can be read with float4 loads


class FlatModule(nn.Module):
    """Mixin for root modules (MultiGenerator, MultiDiscriminatorTrajectory)."""

    _flat = None

    def flat_is_current(self):
        f = self._flat
        if f is None:
            return False
        # Module._apply (.to / .cuda / .double) and load_state_dict(assign=True) re-seat EVERY parameter: three probes
        # (first, middle, last) see that; walking all ~150 parameters cost 10 us, thirty times per eager iteration
        base, items = f.data_ptr(), self._flat_items
        if not items:
            return True
        for p, off in (items[0], items[len(items) // 2], items[-1]):
            if p.data_ptr() != base + 4 * off:
                return False
        return True

    def flat_is_current_full(self):
        """Every parameter checked (the optimizer step does this once per step: a SINGLE re-seated parameter --
        `p.data = ...`, `module.weight = nn.Parameter(...)`, a partial load_state_dict(assign=True) -- escapes the three
        probes of the hot path, and the kernels would keep reading and writing its stale slot)."""
        f = self._flat
        if f is None:
            return False
        base = f.data_ptr()
        # nn.Module.parameters() re-walks the module tree through named_modules(): 150 us per call at ~60 modules, three
        # optimizer steps per iteration.  The module list is walked once per flat buffer (and again every 64th call: a
        # REPLACED submodule is the one thing its cached form cannot see); the parameters are read from the modules' own
        # dictionaries every time, so a re-seated or replaced nn.Parameter is still caught at the very next step
        mods = self.__dict__.get("_flat_mods")
        n = self.__dict__["_flat_checks"] = self.__dict__.get("_flat_checks", 0) + 1
        if mods is None or mods[0] is not f or n % 64 == 0:
            mods = self.__dict__["_flat_mods"] = (f, list(self.modules()))
        seen, named = set(), []
        for m in mods[1]:
            for p in m._parameters.values():
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    named.append(p)
        if len(named) != len(self._flat_items):
            return False
        return all(p is q and q.data_ptr() == base + 4 * off for p, (q, off) in zip(named, self._flat_items))

    def ensure_flat(self):
        if not self.flat_is_current():
            self.flatten_parameters_()
        return self

    def flatten_parameters_(self):
        named = list(self.named_parameters())
        if not named:
            return self
        dev = named[0][1].device
        off, table = 0, []
        for name, p in named:
            assert p.dtype == torch.float32, name
            table.append((name, p, off, p.numel()))
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        grad = torch.zeros(off, dtype=torch.float32, device=dev)
        elem_seg = torch.full((off,), -1, dtype=torch.int32)
        with torch.no_grad():
            for si, (name, p, o, n) in enumerate(table):
                flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = flat[o:o + n].view(p.shape)
                if p.grad is not None:
                    grad[o:o + n].copy_(p.grad.reshape(-1))
                    p.grad = grad[o:o + n].view(p.shape)
                elem_seg[o:o + n] = si
        self._flat, self._flat_grad = flat, grad
        self._flat_items = [(p, o) for _, p, o, _ in table]
        self._flat_names = [name for name, _, _, _ in table]
        self._flat_seg = {name: si for si, (name, _, _, _) in enumerate(table)}
        self._flat_off = {id(p): (o, n) for _, p, o, n in table}
        self._elem_seg = elem_seg.to(dev)
        self._touched = set()
        self._mask_cache = {}
        for m in self.modules():
            if m is not self:
                object.__setattr__(m, "_flat_root", self)
        return self

    # ---- gradient buffer access -------------------------------------------------
    def grad_ptr(self, p):
        """device pointer of p's slot in the flat gradient buffer; attaches p.grad on first touch."""
        o, n = self._flat_off[id(p)]
        self._touched.add(id(p))
        object.__setattr__(self, "_grad_clean", False)  # (nn.Module.__setattr__ costs 3 us per call)
        if p.grad is None:
            view = self._flat_grad[o:o + n].view(p.shape)
            view.zero_()
            p.grad = view
        return self._flat_grad.data_ptr() + 4 * o

    def adopt_foreign_grads(self):
        """A gradient that reached a parameter through plain torch autograd (an index / cat path of the masked
        discriminator forward, or any future torch operator) lives outside the flat buffer.  It is added into the
        parameter's slot and the parameter counts as touched, so that clip + AdamW see exactly the reference's
        `p.grad is not None` set.  (The HIP backward kernels never take this path: host-side pointer compares only.)"""
        base = self._flat_grad.data_ptr()
        for p, o in self._flat_items:
            g = p.grad
            if g is None or g.data_ptr() == base + 4 * o:
                continue
            n = p.numel()
            slot = self._flat_grad[o:o + n]
            if id(p) in self._touched:
                slot.add_(g.reshape(-1))
            else:
                slot.copy_(g.reshape(-1))
                self._touched.add(id(p))
            self._grad_clean = False
            p.grad = slot.view(p.shape)

    def zero_grad_flat(self):
        """Trainer fast path: one memset, every p.grad stays attached.  The memset is skipped when the last
        optimizer step already left the buffer zeroed (FlatAdamW.step(zero_grad=True)) and nothing wrote since."""
        if not getattr(self, "_grad_clean", False):
            self._flat_grad.zero_()
        self._touched.clear()
        for p, o in self._flat_items:
            if p.grad is None:
                n = p.numel()
                p.grad = self._flat_grad[o:o + n].view(p.shape)

    def touched_mask(self):
        """uint8 mask over parameter segments that received a gradient since the last zero_grad_flat()
        (== the reference's `p.grad is not None` set that AdamW / clip_grad_norm_ act on)."""
        key = tuple(1 if id(p) in self._touched else 0 for p, _ in self._flat_items)
        m = self._mask_cache.get(key)
        if m is None:
            m = torch.tensor(key, dtype=torch.uint8).to(self._flat.device)
            self._mask_cache[key] = m
        return m

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none)
        if self._flat is not None:
            self._touched.clear()


def root_of(module):
    r = getattr(module, "_flat_root", None)
    if r is None:
        if not isinstance(module, FlatModule):
            raise RuntimeError("{} is not attached to a flattened root module".format(type(module).__name__))
        r = module
    r.ensure_flat()
    return r
