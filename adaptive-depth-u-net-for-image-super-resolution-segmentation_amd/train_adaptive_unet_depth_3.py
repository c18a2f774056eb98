"""Experiment-1 wrapper: mirror of /root/reference/Super_resolution/code/train_adaptive_unet_depth_3.py
(pins depth_override = 3 and delegates to train())."""
from .train_adaptive_unet import parse_args, train as _train

FIXED_DEPTH = 3


def train(args):
    args.depth_override = FIXED_DEPTH
    return _train(args)


if __name__ == "__main__":
    train(parse_args())
