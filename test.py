#!/usr/bin/env python3
"""Dataset-scale sampling driver with the flags and output layout of the reference's test.py (:23-262): for every pocket of
a data split, ``samples_per_pocket`` pharmacophores -> <output_dir>/pocket_<idx>/pharms.xyz (or trajectories),
sample_time.txt, optional validity / type-frequency metrics.  Unlike the reference (one pocket per batch) the (pocket,
sample) graphs of many pockets share batches of ``max_batch_size`` (PharmacophoreDiff.sample), and with
`torchrun --nproc-per-node N test.py ...` the pockets are dealt round-robin over the GPUs; only the metric counters are
all-reduced (BASELINE config 4)."""
import argparse
import os
import pickle
import shutil
import time
from pathlib import Path

import torch
import yaml


def parse_arguments():
    p = argparse.ArgumentParser()
    p.add_argument('--ckpt', type=Path, help='Path to checkpoint file. Must be inside model dir.', default=None)
    p.add_argument('--model_dir', type=Path, default=None, help='Directory of output from a training run. Will use last.ckpt in this directory.')
    p.add_argument('--samples_per_pocket', type=int, default=1, help="number of samples generated per pocket")
    p.add_argument('--pharm_sizes', nargs="*", type=int, default=[], help="number of pharmacophore centers in each sample, must be of length samples per pocket")
    p.add_argument('--max_batch_size', type=int, default=128, help='maximum feasible batch size due to memory constraints')
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--output_dir', type=Path, default=None)
    p.add_argument('--max_tries', type=int, default=1, help='maximum number of batches to sample per pocket')
    p.add_argument('--dataset_size', type=int, default=None, help='truncate test dataset')
    p.add_argument('--dataset_idx', type=int, default=None)
    p.add_argument('--dataset_idx_as_start', action='store_true', help="Use dataset idx as starting index and sample dataset size")
    p.add_argument('--split', type=str, default='val', help="Specifying which data split to use; options are val or train")
    p.add_argument('--use_ref_pharm_com', action='store_true', help="Initialize each pharmacophore's position at the reference pharmacophore's center of mass")
    p.add_argument('--visualize_trajectory', action='store_true', help="Visualize trajectories of generated pharmacophores")
    p.add_argument('--metrics', action='store_true', help='compute metrics on generated pharmacophores')
    p.add_argument('--pockets_per_call', type=int, default=64, help='pockets whose samples are batched together')
    args = p.parse_args()
    if args.ckpt is None and args.model_dir is None:
        raise ValueError('Must provide either --ckpt or --model_dir')
    if args.pharm_sizes and len(args.pharm_sizes) != args.samples_per_pocket:
        raise ValueError("If pharm_sizes list is provided, must of length sample per pocket")
    return args


def main():
    import pharmacoforge_amd as pfa
    from pharmacoforge_amd.dataset import data_module_from_config
    args = parse_arguments()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("test.py needs an MI355X: the denoising kernels have no CPU fallback")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    device = torch.device('cuda', local % ndev)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29551")
        if world <= ndev:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        group = dist.group.WORLD
    run_dir, model_file = ((args.ckpt.parent.parent, args.ckpt) if args.ckpt is not None
                           else (args.model_dir, args.model_dir / 'checkpoints' / 'last.ckpt'))
    output_dir = args.output_dir if args.output_dir is not None else run_dir / 'samples'
    output_dir.mkdir(exist_ok=True)
    config_file = run_dir / 'config.yaml'
    if not config_file.exists():
        config_file = run_dir / 'config.yml'
        if not config_file.exists():
            raise FileNotFoundError(f'config file not found in {run_dir}')
    with open(config_file, 'r') as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    if rank == 0:
        print(f'{device=}', flush=True)
    torch.manual_seed(args.seed + rank)
    dm = data_module_from_config(config)
    if args.split == 'train':
        dm.setup('fit')
        dataset = dm.train_dataset
    else:
        dm.setup('test')
        dataset = dm.val_dataset
    try:
        model = pfa.PharmacophoreDiff.load_from_checkpoint(model_file).to(device)
    except TypeError:
        model = pfa.PharmacophoreDiff.load_from_checkpoint(model_file, ph_type_map=config['dataset']['ph_type_map']).to(device)
    model.eval()
    if args.dataset_idx is None:
        idxs = list(range(args.dataset_size if args.dataset_size is not None else len(dataset)))
    elif args.dataset_idx_as_start:
        if args.dataset_size is None:
            raise ValueError('Must provide dataset size if dataset_idx_as_start is used')
        idxs = list(range(args.dataset_idx, args.dataset_idx + args.dataset_size))
    else:
        idxs = [args.dataset_idx]
    mine = idxs[rank::world]                       # pockets are independent units: no data-path collective
    all_pharms = []
    for c0 in range(0, len(mine), args.pockets_per_call):
        chunk = mine[c0:c0 + args.pockets_per_call]
        t0 = time.time()
        graphs = [dataset[i] for i in chunk]
        n_pharms = [list(args.pharm_sizes) if args.pharm_sizes else
                    model.pharm_size_dist.sample_uniformly(args.samples_per_pocket).tolist() for _ in chunk]
        init_com = None
        if args.use_ref_pharm_com:
            init_com = torch.stack([g.pharm_x0.mean(dim=0) for g in graphs], dim=0)
        with torch.no_grad():
            per_pocket = model.sample(graphs, n_pharms, max_batch_size=args.max_batch_size, init_pharm_com=init_com,
                                      visualize_trajectory=args.visualize_trajectory)
        dt = (time.time() - t0) / max(len(chunk), 1)
        for dataset_idx, pharms in zip(chunk, per_pocket):
            pocket_dir = output_dir / f'pocket_{dataset_idx}'
            pocket_dir.mkdir(exist_ok=True)
            all_pharms.extend(pharms)
            with open(pocket_dir / 'sample_time.txt', 'w') as f:
                f.write(f'{dt:.2f}')
            raw_dir, prot_file, _ = dataset.get_files(dataset_idx)
            if prot_file is not None and (Path(raw_dir) / prot_file).exists():
                ref_files_dir = pocket_dir / 'reference_files'
                ref_files_dir.mkdir(exist_ok=True)
                shutil.copy(Path(raw_dir) / prot_file, ref_files_dir / Path(prot_file).name)
            if args.visualize_trajectory:
                for k, ph in enumerate(pharms):
                    ph.traj_to_xyz(pocket_dir / f'pharm_{k}_traj.xyz')
            else:
                with open(pocket_dir / 'pharms.xyz', 'w') as f:
                    f.write(''.join(ph.to_xyz_file() for ph in pharms))
        if rank == 0:
            print(f'pockets {chunk[0]}..{chunk[-1]}: {dt:.3f} s per pocket, {dt / max(args.samples_per_pocket, 1):.4f} s per pharmacophore', flush=True)
    if args.metrics:
        analyzer = pfa.SampleAnalyzer()
        metrics = analyzer.analyze(all_pharms, process_group=group)
        freqs = analyzer.pharm_feat_freq(all_pharms, process_group=group)
        if rank == 0:
            print(metrics)
            with open(output_dir / 'metrics.txt', 'w') as f:
                f.write('\n'.join(f'{k}: {v:.3f}' for k, v in metrics.items()))
            with open(output_dir / 'metrics.pkl', 'wb') as f:
                pickle.dump(metrics, f)
            with open(output_dir / f'pharm_counts_{args.dataset_idx}.txt', 'w') as f:
                f.write(str(freqs.tolist()))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
