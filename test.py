#!/usr/bin/env python3
"""Dataset-scale sampling (BASELINE config 4): for every pocket of a data split draw ``samples_per_pocket``
pharmacophores and write <output_dir>/pocket_<idx>/{pharms.xyz | pharm_<k>_traj.xyz, sample_time.txt,
reference_files/}, plus validity / type-frequency metrics over the whole job.

Flag names, types, defaults and the output tree are those of the reference's test.py (:23-60, :113-262).  The execution
model is this repository's: the (pocket, sample) graphs of ``--pockets_per_call`` pockets share device batches of
``--max_batch_size`` (PharmacophoreDiff.sample; the reference runs one pocket per batch), and under
``torchrun --nproc-per-node N test.py ...`` the pockets are dealt over the GPUs by work -- greedy on
(pp edges x samples), pharmacoforge_amd.sharding -- with no data-path collective; only the eight metric counters are
all-reduced (RCCL when every rank has its own GPU)."""
import argparse
import os
import pickle
import shutil
import time
from pathlib import Path

import torch
import yaml

_FLAGS = [
    ('--ckpt', dict(type=Path, default=None, help='checkpoint file at <run>/checkpoints/<name>.ckpt')),
    ('--model_dir', dict(type=Path, default=None, help='training run directory; checkpoints/last.ckpt is used')),
    ('--samples_per_pocket', dict(type=int, default=1, help='pharmacophores drawn for each pocket')),
    ('--pharm_sizes', dict(nargs='*', type=int, default=[], help='centers per pharmacophore, one integer per sample (default: uniform 3..8)')),
    ('--max_batch_size', dict(type=int, default=128, help='graphs per device batch')),
    ('--seed', dict(type=int, default=42, help='torch seed (each rank adds its rank)')),
    ('--output_dir', dict(type=Path, default=None, help='default: <run>/samples')),
    ('--max_tries', dict(type=int, default=1, help='accepted for compatibility; every requested sample is produced in one pass')),
    ('--dataset_size', dict(type=int, default=None, help='use only this many pockets')),
    ('--dataset_idx', dict(type=int, default=None, help='a single pocket, or the first one with --dataset_idx_as_start')),
    ('--dataset_idx_as_start', dict(action='store_true', help='sample --dataset_size pockets beginning at --dataset_idx')),
    ('--split', dict(type=str, default='val', help="'val' or 'train'")),
    ('--use_ref_pharm_com', dict(action='store_true', help='start the centers at the reference pharmacophore\'s mean position')),
    ('--visualize_trajectory', dict(action='store_true', help='write every denoising frame, one xyz file per sample')),
    ('--metrics', dict(action='store_true', help='validity and feature-type counts over all samples of all ranks')),
    ('--pockets_per_call', dict(type=int, default=64, help='pockets whose samples are batched together (not in the reference)')),
]


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    a = parser.parse_args(argv)
    if a.ckpt is None and a.model_dir is None:
        raise ValueError('no model given: pass --ckpt or --model_dir')
    if a.pharm_sizes and len(a.pharm_sizes) != a.samples_per_pocket:
        raise ValueError(f'--pharm_sizes lists {len(a.pharm_sizes)} sizes for --samples_per_pocket {a.samples_per_pocket}')
    if a.dataset_idx_as_start and (a.dataset_idx is None or a.dataset_size is None):
        raise ValueError('--dataset_idx_as_start needs both --dataset_idx and --dataset_size')
    return a


def selected_pockets(args, n_total):
    if args.dataset_idx is None:
        return list(range(n_total if args.dataset_size is None else args.dataset_size))
    if args.dataset_idx_as_start:
        return list(range(args.dataset_idx, args.dataset_idx + args.dataset_size))
    return [args.dataset_idx]


def init_ranks():
    """-> (rank, world, device, process group or None).  RCCL when every rank owns a GPU, gloo for dry runs on fewer."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from pharmacoforge_amd.sharding import pin_host_threads
    pin_host_threads(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))       # before anything touches the GPU
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    device = torch.device('cuda', local % ndev)
    if world == 1:
        return rank, world, device, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29551")
    if world <= ndev:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, device, dist.group.WORLD


def write_pocket(out_root, dataset, idx, pharms, seconds, trajectories):
    pocket_dir = out_root / f'pocket_{idx}'
    pocket_dir.mkdir(exist_ok=True)
    (pocket_dir / 'sample_time.txt').write_text(f'{seconds:.2f}')
    raw_dir, prot_file, _ = dataset.get_files(idx)
    if prot_file is not None and (Path(raw_dir) / prot_file).exists():
        ref_dir = pocket_dir / 'reference_files'
        ref_dir.mkdir(exist_ok=True)
        shutil.copy(Path(raw_dir) / prot_file, ref_dir / Path(prot_file).name)
    if trajectories:
        for k, ph in enumerate(pharms):
            ph.traj_to_xyz(pocket_dir / f'pharm_{k}_traj.xyz')
    else:
        (pocket_dir / 'pharms.xyz').write_text(''.join(ph.to_xyz_file() for ph in pharms))


def main(argv=None):
    import pharmacoforge_amd as pfa
    from pharmacoforge_amd.dataset import data_module_from_config
    from pharmacoforge_amd.sharding import shard_by_work
    from generate_pharmacophores import load_model, locate_run

    args = parse_arguments(argv)
    if not torch.cuda.is_available():
        raise SystemExit("test.py needs an MI355X: the denoising kernels have no CPU fallback")
    rank, world, device, group = init_ranks()
    run_dir, ckpt, config = locate_run(args)
    out_root = args.output_dir if args.output_dir is not None else run_dir / 'samples'
    out_root.mkdir(exist_ok=True)
    if rank == 0:
        print(f'{device=}', flush=True)
    torch.manual_seed(args.seed + rank)
    dm = data_module_from_config(config)
    dm.setup('fit' if args.split == 'train' else 'test')
    dataset = dm.train_dataset if args.split == 'train' else dm.val_dataset
    model = load_model(pfa, ckpt, config, device)

    pockets = selected_pockets(args, len(dataset))
    # pockets are independent units; a pocket's cost is its pp edge count times its sample count (sizes 3-8 change it by
    # < 5 %): every rank computes the same greedy assignment, nothing is communicated
    epp = dataset.pp_edge_counts()
    mine = [pockets[j] for j in shard_by_work([float(epp[i] + 1) * args.samples_per_pocket for i in pockets], world)[rank]]
    all_pharms = []
    for c0 in range(0, len(mine), args.pockets_per_call):
        chunk = mine[c0:c0 + args.pockets_per_call]
        t0 = time.time()
        graphs = [dataset[i] for i in chunk]
        sizes = [list(args.pharm_sizes) or model.pharm_size_dist.sample_uniformly(args.samples_per_pocket).tolist()
                 for _ in chunk]
        coms = torch.stack([g.pharm_x0.mean(dim=0) for g in graphs]) if args.use_ref_pharm_com else None
        with torch.no_grad():
            per_pocket = model.sample(graphs, sizes, max_batch_size=args.max_batch_size, init_pharm_com=coms,
                                      visualize_trajectory=args.visualize_trajectory)
        seconds = (time.time() - t0) / len(chunk)
        for idx, pharms in zip(chunk, per_pocket):
            write_pocket(out_root, dataset, idx, pharms, seconds, args.visualize_trajectory)
            all_pharms += pharms
        if rank == 0:
            print(f'pockets {chunk[0]}..{chunk[-1]}: {seconds:.3f} s per pocket, '
                  f'{seconds / max(args.samples_per_pocket, 1):.4f} s per pharmacophore', flush=True)
    if args.metrics:
        analyzer = pfa.SampleAnalyzer()
        metrics = analyzer.analyze(all_pharms, process_group=group, device=device)
        freqs = analyzer.pharm_feat_freq(all_pharms, process_group=group, device=device)
        if rank == 0:
            print(metrics)
            (out_root / 'metrics.txt').write_text('\n'.join(f'{k}: {v:.3f}' for k, v in metrics.items()))
            with open(out_root / 'metrics.pkl', 'wb') as f:
                pickle.dump(metrics, f)
            (out_root / f'pharm_counts_{args.dataset_idx}.txt').write_text(str(freqs.tolist()))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
