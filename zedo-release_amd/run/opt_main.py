"""python -m run.opt_main --config configs/optim/concat_pose_optimization_h36m.py --ckpt_dir D --ckpt_name F --hypo H [--gt]

Same entry point and flags as the reference's run/opt_main.py (:42-50, :230-232); absl / ml_collections are
not required (the config file is imported directly).  `--synthetic N` runs without dataset, cluster and
checkpoint files.  Multi-GPU: launch with torchrun, one process per GPU.
"""
import sys

from run._driver import build_parser, run


def parse_args(argv):
    return build_parser("valid score model").parse_args(argv[1:])


def main(args):
    _, errs = run(args, inference=False)
    return errs


if __name__ == "__main__":
    main(parse_args(sys.argv))
