"""ZeDO optimisation config for '3dhp' (values of reference configs/optim/concat_pose_optimization_3dhp.py)."""
from configs._configdict import ConfigDict
from configs.default_pose_gen_configs import get_default_configs


def get_config():
    config = get_default_configs()
    config.training.sde = "subvpsde"
    config.training.continuous = True
    config.training.reduce_mean = True

    config.sampling.method = "pc"
    config.sampling.predictor = "euler_maruyama"
    config.sampling.corrector = "none"

    config.data.centered = True
    config.data.dataset = "3dhp"

    model = config.model
    model.scale_by_sigma = False
    model.ema_rate = 0.9999
    model.embedding_type = "positional"
    model.t = 0.1            # sde.T: the sampler integrates from t = 0.1 down to sampling_eps

    config.ZeDO = ZeDO = ConfigDict()
    ZeDO.IPO_iterations = 500
    ZeDO.IPO_keylist = [0, 1, 4]
    ZeDO.RotAxes = "z"
    ZeDO.IPO_T = 3
    ZeDO.IPO_minScaleT = 0.5
    ZeDO.IPO_maxScaleT = 2
    ZeDO.OIL_iterations = 1000
    ZeDO.sample = 3
    ZeDO.batch = 959
    ZeDO.sampling_eps = 0.01
    return config
