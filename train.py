#!/usr/bin/env python3
"""Training driver in the shape of the reference's train.py (:26-175) without Lightning / wandb: config yaml ->
PharmacophoreDiff + CrossdockedDataModule -> epochs of training_step / backward (HIP gradient kernels) / fused Adam,
validation loss, ReduceLROnPlateau on 'val total loss', Lightning-format checkpoints (<output_dir>/<run>/checkpoints/
last.ckpt + config.yaml next to them, which generate_pharmacophores.py reads).  One process per GPU
(torchrun --nproc-per-node N train.py ...): each rank draws its own batches, gradients are averaged with one all-reduce."""
import argparse
import os
from datetime import datetime
from pathlib import Path

import torch
import yaml


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--resume', type=Path, default=None)
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--max_steps', type=int, default=None, help='stop after this many optimiser steps (smoke runs)')
    args = p.parse_args()
    import pharmacoforge_amd as pfa
    from pharmacoforge_amd.dataset import data_module_from_config

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: the kernels have no CPU fallback")
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device("cuda", local % torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)
    model = pfa.model_from_config(config).to(dev)
    if args.resume is not None:
        ck = args.resume / 'checkpoints' / 'last.ckpt' if args.resume.is_dir() else args.resume
        model.load_state_dict(torch.load(str(ck), map_location='cpu', weights_only=False)['state_dict'], strict=True)
    dm = data_module_from_config(config)
    dm.setup('fit')
    run_dir = Path(config['training']['output_dir']) / f"{config['wandb']['name'].replace(' ', '_')}_{datetime.now():%Y-%m-%d_%H-%M-%S}"
    if rank == 0:
        (run_dir / 'checkpoints').mkdir(parents=True, exist_ok=True)
        with open(run_dir / 'config.yaml', 'w') as f:
            yaml.dump(config, f)
    lr_cfg = config['lr_scheduler']
    opt = pfa.FlatAdam(model.dynamics, lr=lr_cfg['base_lr'], weight_decay=lr_cfg.get('weight_decay', 0.0))
    plateau = dict(lr_cfg.get('reducelronplateau', {}))
    best, bad_epochs, step = float('inf'), 0, 0
    for epoch in range(config['training']['trainer_args']['max_epochs']):
        model.train()
        sampler = torch.utils.data.distributed.DistributedSampler(dm.train_dataset, world, rank, shuffle=True) if world > 1 else None
        if sampler is not None:
            sampler.set_epoch(epoch)
        for i, g in enumerate(dm.train_dataloader(shuffle=sampler is None, sampler=sampler)):
            opt.zero_grad()
            loss = model.training_step(g.to(dev), i)
            loss.backward()
            if world > 1:
                model.dynamics.allreduce_gradients()
            opt.step()
            step += 1
            if rank == 0 and step % 50 == 0:
                print(f"epoch {epoch} step {step} " + " ".join(f"{k}={float(v):.4f}" for k, v in model.last_metrics.items()), flush=True)
            if args.max_steps and step >= args.max_steps:
                break
        model.eval()
        tot, n = 0.0, 0
        with torch.no_grad():
            for i, g in enumerate(dm.val_dataloader()):
                tot += float(model.validation_step(g.to(dev), i)) * g.batch_size
                n += g.batch_size
        val = tot / max(n, 1)
        if plateau:                                   # ReduceLROnPlateau(mode='min', factor, patience, min_lr)
            if val < best - 1e-12:
                best, bad_epochs = val, 0
            else:
                bad_epochs += 1
                if bad_epochs > plateau.get('patience', 10):
                    opt.param_groups[0]['lr'] = max(opt.param_groups[0]['lr'] * plateau.get('factor', 0.1), plateau.get('min_lr', 0.0))
                    bad_epochs = 0
        if rank == 0:
            print(f"epoch {epoch}: val total loss {val:.5f} lr {opt.param_groups[0]['lr']:.2e}", flush=True)
            model.save_checkpoint(run_dir / 'checkpoints' / 'last.ckpt')
        if args.max_steps and step >= args.max_steps:
            break
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
