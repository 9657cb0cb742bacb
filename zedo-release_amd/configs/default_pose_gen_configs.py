"""Default values read by the sampling path (the fields of reference configs/default_pose_gen_configs.py that
run/opt_main.py, run/inference.py and lib/algorithms/advanced/* look at)."""
import torch

from configs._configdict import ConfigDict

_DEFAULTS = {
    "training": dict(continuous=True, reduce_mean=False, cond_pose_mask_prob=0.0, cond_part_mask_prob=0.0,
                     cond_joint_mask_prob=0.0, cond_3d_prob=0.0),
    "sampling": dict(n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.16),
    "data": dict(dataset="h36m", centered=False),
    "model": dict(sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20.0, dropout=0.1,
                  embedding_type="fourier"),
}


def get_default_configs():
    config = ConfigDict()
    for section, values in _DEFAULTS.items():
        node = ConfigDict()
        for key, value in values.items():
            setattr(node, key, value)
        setattr(config, section, node)
    config.OUTPUT_DIR = "./output"
    config.seed = 42
    config.device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    return config
