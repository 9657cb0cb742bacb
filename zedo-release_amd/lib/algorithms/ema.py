"""Exponential moving average of parameters (reference lib/algorithms/ema.py).

The evaluation drivers construct it and call `load_state_dict` only (run/opt_main.py:79,135); `copy_to` is
never called there, so the raw `model_state_dict` weights are what the sampler runs.
The full interface is kept so that checkpoints round-trip.
"""
import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def update(self, parameters):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for s, p in zip(self.shadow_params, [p for p in parameters if p.requires_grad]):
                s.sub_((1.0 - decay) * (s - p))

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, [p for p in parameters if p.requires_grad]):
            p.data.copy_(s.data)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict["decay"]
        self.num_updates = state_dict["num_updates"]
        self.shadow_params = state_dict["shadow_params"]
