"""Default values read by the sampling path (reference configs/default_pose_gen_configs.py)."""
import torch

from configs._configdict import ConfigDict


def get_default_configs():
    config = ConfigDict()
    config.OUTPUT_DIR = "./output"

    config.training = training = ConfigDict()
    training.continuous = True
    training.reduce_mean = False
    training.cond_pose_mask_prob = 0.0
    training.cond_part_mask_prob = 0.0
    training.cond_joint_mask_prob = 0.0
    training.cond_3d_prob = 0.0

    config.sampling = sampling = ConfigDict()
    sampling.n_steps_each = 1
    sampling.noise_removal = True
    sampling.probability_flow = False
    sampling.snr = 0.16

    config.data = data = ConfigDict()
    data.dataset = "h36m"
    data.centered = False

    config.model = model = ConfigDict()
    model.sigma_min = 0.01
    model.sigma_max = 50
    model.num_scales = 1000
    model.beta_min = 0.1
    model.beta_max = 20.0
    model.dropout = 0.1
    model.embedding_type = "fourier"

    config.seed = 42
    config.device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
    return config
