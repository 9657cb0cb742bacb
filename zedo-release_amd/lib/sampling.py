"""Name-compatibility alias.  The reference's lib/sampling.py is an older duplicate of
lib/algorithms/advanced/sampling.py that nothing imports and that cannot be imported;
BASELINE.json names it, so the module path exists here and re-exports the live sampler."""
from lib.algorithms.advanced.sampling import *  # noqa: F401,F403
from lib.algorithms.advanced.sampling import get_sampling_fn, get_pc_sampler  # noqa: F401
