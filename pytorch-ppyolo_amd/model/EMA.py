"""ExponentialMovingAverage of the trainable parameters (reference model/EMA.py:15-64), same methods.

The reference keeps the shadows as numpy arrays on the host and round-trips every parameter through `.cpu()` at every
update; here the shadows live in ONE flat device buffer and an update is one HIP launch per parameter tensor
(ppy_ema_update_f32: `decay * old + (1 - decay) * new` in fp32, the factors rounded to fp32 first -- the arithmetic numpy
performs on the reference's float32 arrays; tests/golden g13).  apply() / restore() copy INTO the parameters (the reference
rebinds `param.data`), so optimizers and the HIP training step keep seeing the same storage."""
import torch


class ExponentialMovingAverage():
    def __init__(self, model, decay, thres_steps=True):
        self._model = model
        self._decay = decay
        self._thres_steps = thres_steps
        self._shadow = {}
        self._backup = {}

    def _params(self):
        return [(n, p) for n, p in self._model.named_parameters() if p.requires_grad is True]

    def register(self):
        self._update_step = 0
        ps = self._params()
        if not ps or ps[0][1].device.type != 'cuda':
            raise RuntimeError('the HIP EMA needs the model on a ROCm device; there is no CPU path')
        offs, total = {}, 0
        for n, p in ps:
            offs[n] = (total, p.numel())
            total += (p.numel() + 63) // 64 * 64
        self._flat = torch.zeros(total, dtype=torch.float32, device=ps[0][1].device)
        for n, p in ps:
            o, m = offs[n]
            self._shadow[n] = self._flat[o:o + m].view(p.shape)
            self._shadow[n].copy_(p.detach())

    def update(self):
        from ppyolo_hip import ops as K
        decay = None
        step = self._update_step if self._thres_steps else None
        for n, p in self._params():
            assert n in self._shadow
            src = p.detach()
            decay = K.ema_update(self._shadow[n].view(-1), src.contiguous().view(-1), step if step is not None else 10 ** 12, self._decay)
        self._update_step += 1
        return decay

    def apply(self):
        with torch.no_grad():
            for n, p in self._params():
                assert n in self._shadow
                self._backup[n] = p.detach().clone()
                p.copy_(self._shadow[n])

    def restore(self):
        with torch.no_grad():
            for n, p in self._params():
                assert n in self._backup
                p.copy_(self._backup[n])
        self._backup = {}
