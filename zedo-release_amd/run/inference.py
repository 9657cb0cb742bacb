"""python -m run.inference --config configs/optim/concat_pose_optimization_wild.py --ckpt_dir D --ckpt_name F --hypo H
       --data poses.npz [--eval]

The reference's in-the-wild driver (run/inference.py): every hypothesis is saved to results.npy
([N, H, 17, 3]); the best-of-H metric is printed only with --eval (:239-241).
"""
import sys

from run._driver import build_parser, run


def parse_args(argv):
    return build_parser("ZeDO inference", inference=True).parse_args(argv[1:])


def main(args):
    results, errs = run(args, inference=True)
    return results, errs


if __name__ == "__main__":
    main(parse_args(sys.argv))
