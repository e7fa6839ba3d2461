#!/usr/bin/env python3
"""Training driver in the shape of the reference's train.py (:26-175) without Lightning / wandb: config yaml ->
PharmacophoreDiff + CrossdockedDataModule -> epochs of training_step / backward (HIP gradient kernels) / fused Adam,
validation loss, ReduceLROnPlateau on 'val total loss', Lightning-format checkpoints (<output_dir>/<run>/checkpoints/
last.ckpt + config.yaml next to them, which generate_pharmacophores.py reads).

Data parallel, one process per GPU (``torchrun --nproc-per-node N train.py ...``):
  * every rank builds the model, then rank 0's parameters are BROADCAST, so all replicas start from the same weights
    whatever their RNG state; only afterwards does the seed get its rank offset (data order, noise, dropout draws);
  * each rank draws its own batches (DistributedSampler); gradients are averaged with ONE all-reduce of the flat
    gradient vector per step, so the replicas stay bitwise identical (``--check_replicas`` verifies that every epoch);
  * the validation loss is summed over the ranks before the plateau logic, so every rank takes the same LR decision;
  * rank 0 writes the checkpoint, which carries the optimiser moments, step count, learning rate, epoch and plateau
    state next to the weights; ``--resume`` restores all of them (a Lightning checkpoint's 'optimizer_states' /
    'lr_schedulers' / 'epoch' / 'global_step' keys are used for that)."""
import argparse
import os
from datetime import datetime
from pathlib import Path

import torch
import yaml


def _collective(fn, t, **kw):
    """all_reduce / broadcast on ``t``; the gloo fallback (more ranks than GPUs: dry runs on one card) goes through a host
    copy, RCCL works on the device tensor."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() != "nccl":
        c = t.cpu()
        fn(c, **kw)
        t.copy_(c)
    else:
        fn(t, **kw)
    return t


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--resume', type=Path, default=None, help='run directory or checkpoint to continue from')
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--max_steps', type=int, default=None, help='stop after this many optimiser steps (smoke runs)')
    p.add_argument('--check_replicas', action='store_true',
                   help='data parallel: verify after every epoch that all ranks hold bitwise identical parameters')
    p.add_argument('--bind_prefetch', action='store_true',
                   help="bind the next batch on the dynamics' twin handle from a worker thread while a step is enqueued: the host's time per "
                        "step drops by a third (for ranks that share a busy host); the device-bound step itself pays ~1.5 %% for it")
    p.add_argument('--train_precision', choices=['f32', 'bf16'], default='f32',
                   help="arithmetic of the dense Linears in the training step: f32 (the reference's) or the bf16 leg (bf16 matrix "
                        "instructions, fp32 accumulation, fp32 master weights and optimiser)")
    args = p.parse_args()
    import pharmacoforge_amd as pfa
    from pharmacoforge_amd.dataset import data_module_from_config

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # the rank's slice of the host and its pool sizes, before anything touches the GPU (a step is as long on the host as on the device)
    from pharmacoforge_amd.sharding import pin_host_threads
    host = pin_host_threads(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world > 1:
        print(f"[rank {rank}] host share: {host}", flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: the kernels have no CPU fallback")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if world <= ndev:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:                                   # ranks share a card: RCCL needs one GPU per rank
            dist.init_process_group("gloo", rank=rank, world_size=world)
    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    model = pfa.model_from_config(config).to(dev)
    model.dynamics.set_train_precision(args.train_precision)
    lr_cfg = config['lr_scheduler']
    opt = pfa.FlatAdam(model.dynamics, lr=lr_cfg['base_lr'], weight_decay=lr_cfg.get('weight_decay', 0.0))
    plateau = dict(lr_cfg.get('reducelronplateau', {}))
    best, bad_epochs, step, first_epoch = float('inf'), 0, 0, 0
    if args.resume is not None:
        ck_path = args.resume / 'checkpoints' / 'last.ckpt' if args.resume.is_dir() else args.resume
        ck = torch.load(str(ck_path), map_location='cpu', weights_only=False)
        model.load_state_dict(ck['state_dict'], strict=True)
        model.dynamics.engine()                                  # flat parameter vector on the device
        if ck.get('optimizer_states'):
            opt.load_state_dict(ck['optimizer_states'][0])
        if ck.get('lr_schedulers'):
            best, bad_epochs = ck['lr_schedulers'][0].get('best', best), ck['lr_schedulers'][0].get('num_bad_epochs', 0)
        first_epoch, step = int(ck.get('epoch', -1)) + 1, int(ck.get('global_step', 0))
    model.dynamics.engine()
    if world > 1:
        # identical replicas: rank 0's weights (fresh or resumed) and optimiser state go to everyone
        _collective(dist.broadcast, model.dynamics._flat, src=0)
        model.dynamics._weights_stamp = None                     # the engine re-reads the flat vector
        meta = [opt.state_dict_meta(), best, bad_epochs, step, first_epoch] if rank == 0 else [None] * 5
        dist.broadcast_object_list(meta, src=0)
        opt.ensure_state(*meta[0])
        best, bad_epochs, step, first_epoch = meta[1:]
        _collective(dist.broadcast, opt.exp_avg, src=0)
        _collective(dist.broadcast, opt.exp_avg_sq, src=0)
    if args.seed is not None:
        torch.manual_seed(args.seed + 1000003 * (rank + 1))      # from here on: per-rank data order, noise, dropout
    dm = data_module_from_config(config)
    dm.setup('fit')
    run_dir = Path(config['training']['output_dir']) / f"{config['wandb']['name'].replace(' ', '_')}_{datetime.now():%Y-%m-%d_%H-%M-%S}"
    if rank == 0:
        (run_dir / 'checkpoints').mkdir(parents=True, exist_ok=True)
        with open(run_dir / 'config.yaml', 'w') as f:
            yaml.dump(config, f)
    stop = False
    for epoch in range(first_epoch, config['training']['trainer_args']['max_epochs']):
        model.train()
        sampler = torch.utils.data.distributed.DistributedSampler(dm.train_dataset, world, rank, shuffle=True) if world > 1 else None
        if sampler is not None:
            sampler.set_epoch(epoch)
        loader = dm.train_dataloader(shuffle=sampler is None, sampler=sampler)
        model.attach_trainer(dm, loader, current_epoch=epoch, optimizer=opt)
        # one batch of look-ahead: while a step's backward and optimiser step are being enqueued, the NEXT batch is bound on the
        # dynamics' twin handle by a worker thread (the bind is half a step's host time; PharmRecDynamicsGVP.prefetch_graph)
        batches = iter(loader)
        nxt = next(batches, None)
        nxt = None if nxt is None else nxt.to(dev)
        i = -1
        while nxt is not None:
            i += 1
            g = nxt
            nxt = next(batches, None)
            nxt = None if nxt is None else nxt.to(dev)
            opt.zero_grad(lazy=True)
            loss = model.training_step(g, i)
            if nxt is not None and args.bind_prefetch:
                model.dynamics.prefetch_graph(nxt)
            loss.backward()
            if world > 1:
                model.dynamics.allreduce_gradients()
            opt.step()
            step += 1
            if rank == 0 and step % 50 == 0:
                print(f"epoch {epoch} step {step} " + " ".join(f"{k}={float(v):.4f}" for k, v in model.last_metrics.items()), flush=True)
            if args.max_steps and step >= args.max_steps:
                stop = True
                break
        model.eval()
        acc = torch.zeros(2, dtype=torch.float64)
        with torch.no_grad():
            for i, g in enumerate(dm.val_dataloader()):
                if i % world != rank:                            # validation batches dealt over the ranks
                    continue
                acc[0] += float(model.validation_step(g.to(dev), i)) * g.batch_size
                acc[1] += g.batch_size
        if world > 1:                                            # every rank sees the same 'val total loss'
            acc = acc.to(dev) if dist.get_backend() == "nccl" else acc
            dist.all_reduce(acc)
            acc = acc.cpu()
        val = float(acc[0] / acc[1].clamp(min=1))
        if plateau:                                              # ReduceLROnPlateau(mode='min', factor, patience, min_lr)
            if val < best - 1e-12:
                best, bad_epochs = val, 0
            else:
                bad_epochs += 1
                if bad_epochs > plateau.get('patience', 10):
                    opt.param_groups[0]['lr'] = max(opt.param_groups[0]['lr'] * plateau.get('factor', 0.1), plateau.get('min_lr', 0.0))
                    bad_epochs = 0
        if args.check_replicas and world > 1:
            flat = model.dynamics._flat
            sig = torch.stack([flat.double().sum(), flat.double().abs().sum(), flat[::997].double().square().sum()])
            lo, hi = sig.clone(), sig.clone()
            _collective(dist.all_reduce, lo, op=dist.ReduceOp.MIN)
            _collective(dist.all_reduce, hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                raise RuntimeError(f"replicas diverged after epoch {epoch}: parameter checksums differ by {(hi - lo).tolist()}")
            if rank == 0:
                print(f"epoch {epoch}: replicas identical on {world} ranks (checksum {float(sig[0]):.9e})", flush=True)
        if rank == 0:
            print(f"epoch {epoch}: val total loss {val:.5f} lr {opt.param_groups[0]['lr']:.2e}", flush=True)
            model.save_checkpoint(run_dir / 'checkpoints' / 'last.ckpt', epoch=epoch, global_step=step,
                                  optimizer_states=[opt.state_dict()],
                                  lr_schedulers=[{'best': best, 'num_bad_epochs': bad_epochs}])
        if stop:
            break
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
