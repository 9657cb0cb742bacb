"""Exponential moving average of parameters (reference lib/algorithms/ema.py).

The evaluation drivers construct it and call `load_state_dict` only (run/opt_main.py:79,135); `copy_to` is
never called there, so the raw `model_state_dict` weights are what the sampler runs.
The full interface is kept so that checkpoints round-trip.
"""
import torch


def _trainable(parameters):
    return [p for p in parameters if p.requires_grad]


class ExponentialMovingAverage:
    """shadow <- shadow - (1 - d) (shadow - p), with d = min(decay, (1 + n) / (10 + n)) while updates are counted."""

    _FIELDS = ("decay", "num_updates", "shadow_params")

    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay, self.num_updates = decay, (0 if use_num_updates else None)
        self.shadow_params = [p.clone().detach() for p in _trainable(parameters)]
        self.collected_params = []

    def _rate(self):
        if self.num_updates is None:
            return self.decay
        self.num_updates += 1
        return min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))

    @torch.no_grad()
    def update(self, parameters):
        keep = self._rate()
        for shadow, p in zip(self.shadow_params, _trainable(parameters)):
            shadow.sub_((1.0 - keep) * (shadow - p))

    @torch.no_grad()
    def copy_to(self, parameters):
        # p.copy_ (not p.data.copy_): `.data` has a version counter of its own, and the parameter's counter is what
        # ScoreModelFC_Adv watches to know that its packed device copy is stale
        for shadow, p in zip(self.shadow_params, _trainable(parameters)):
            p.copy_(shadow.data)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    @torch.no_grad()
    def restore(self, parameters):
        for saved, p in zip(self.collected_params, parameters):
            p.copy_(saved.data)

    def state_dict(self):
        return {k: getattr(self, k) for k in self._FIELDS}

    def load_state_dict(self, state_dict):
        for k in self._FIELDS:
            setattr(self, k, state_dict[k])
