#!/usr/bin/env python3
"""Entry point with the reference's surface (GLfusion/main.py:885-965): the same `config` dict keys, `--mode
{train,val}`, `main(rank, config)`, `Trainer(config)` -- running the MI355X HIP engine on synthetic clips (the
reference's data files are not shipped).  Multi-GPU: `python -m torch.distributed.run --nproc-per-node N main.py
--mode train` (one process per GPU, RCCL gradient all-reduce) instead of the reference's nn.DataParallel."""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch
import torch.distributed as dist


def main(rank, config):
    if "local_rank" not in config["train"]:
        config["train"]["local_rank"] = config["train"]["global_rank"] = rank
    config["train"]["device"] = torch.device("cuda", config["train"]["local_rank"])
    parser = argparse.ArgumentParser()
    parser.add_argument("--mode", type=str, required=True)
    args = parser.parse_args()
    from glfusion_amd.engine import Trainer
    trainer = Trainer(config)
    if args.mode == "train":
        trainer.train(is_backbone=False, is_cycle=True)
    elif args.mode == "val":
        print(trainer.eval(net_path="./path/to/ckpt", is_fuse=True, raw_data=True))
    else:
        raise SystemExit("--mode visual (matplotlib dumps, main.py:546-648) is out of scope of the hot path")


if __name__ == "__main__":
    config = {
        "train": {
            "cudnn": True, "enable_GPUs_id": [0], "device_ids": [0], "batch_size": 8, "num_workers": 8,
            "num_epochs": 1, "clip_length": 40, "view_num": ["1", "3", "4"], "test_view": ["1", "3", "4"],
            "dense_cyc": False, "seg_parts": True, "record_params": False, "save_dir": "./result/ckpt",
            "log_dir": "./result/log_info/log_01", "data_list_path": "./", "use_data": ["rmyy"], "alpha": 0.8,
            "is_load": False,
            # build-only keys
            "iters_per_epoch": 2,
        },
        "net": {"opt": {"opt_name": "Adam", "lr": 3e-4, "step_size": 50, "params": (0.9, 0.999), "weight_decay": 1e-5},
                "opt_loss_weight": {"opt_name": "Adam", "lr": 1e-3, "step_size": 50, "params": (0.9, 0.999), "weight_decay": 1e-5}},
    }
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    config["train"]["world_size"] = world
    config["train"]["distributed"] = world > 1
    config["train"]["local_rank"], config["train"]["global_rank"] = local, rank
    if world > 1:
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    main(local, config)
