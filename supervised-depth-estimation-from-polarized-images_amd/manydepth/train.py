"""Entry point: ``python -m manydepth.train <flags>`` (reference manydepth/train.py:7-18).

Single GPU: run as is.  Data parallel: ``python -m torch.distributed.run --nproc-per-node N -m
manydepth.train <flags>`` -- one process per GPU, RCCL gradient all-reduce over xGMI.
"""
import os

import torch

from .trainer import Trainer
from .options import MonodepthOptions


def main():
    opts = MonodepthOptions().parse()
    if int(os.environ.get("WORLD_SIZE", 1)) > 1 and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.distributed.init_process_group("nccl")      # "nccl" is RCCL on ROCm
    Trainer(opts).train()


if __name__ == "__main__":
    main()
