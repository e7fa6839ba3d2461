#!/usr/bin/env python3
"""Sampling CLI with the flags, directory layout and output files of the reference's generate_pharmacophores.py
(:29-66, :236-392), on the MI355X library: PDB receptor + SDF reference ligand (or a residue list) -> pocket graph ->
PharmacophoreDiff.sample_given_receptor -> <output_dir>/<receptor>/pharms.xyz (or per-sample trajectories).
Receptor parsing needs neither Biopython nor rdkit (pharmacoforge_amd.pocket_io)."""
import argparse
import pickle
import shutil
import time
from pathlib import Path

import torch
import yaml


def parse_arguments():
    p = argparse.ArgumentParser()
    p.add_argument('receptor_file', type=Path, help='PDB file of the receptor')
    p.add_argument('--ref_ligand_file', type=Path, help='sdf file of ligand used to define the pocket')
    p.add_argument('--residue_list', nargs="+", type=str, default=[], help="Residues that define the pocket in the form chain ID:residue idx")
    p.add_argument('--ckpt', type=Path, help='Path to checkpoint file. Must be inside model dir.', default=None)
    p.add_argument('--model_dir', type=Path, default=None, help='Directory of output from a training run. Will use last.ckpt in this directory.')
    p.add_argument('--samples_per_pocket', type=int, default=1, help="number of samples generated per pocket")
    p.add_argument('--pharm_sizes', nargs="+", type=int, default=[], help="number of pharmacophore centers in each sample, must be of length samples per pocket")
    p.add_argument('--output_dir', type=str, default='generated_pharms/')
    p.add_argument('--receptor_name', type=str, default=None)
    p.add_argument('--max_batch_size', type=int, default=128, help='maximum feasible batch size due to memory constraints')
    p.add_argument('--seed', type=int, default=42, help='random seed as an integer.')
    p.add_argument('--use_ref_lig_com', action='store_true', help="Initialize each pharmacophore's position at the reference ligand's center of mass")
    p.add_argument('--visualize_trajectory', action='store_true', help="Visualize trajectories of generated pharmacophores")
    p.add_argument('--metrics', action='store_true', help='compute metrics on generated pharmacophores')
    args = p.parse_args()
    if args.ckpt is not None and args.model_dir is not None:
        raise ValueError('only model_file or model_dir can be specified but not both')
    if args.ckpt is None and args.model_dir is None:
        raise ValueError('one of model_file or model_dir must be specified')
    if args.pharm_sizes and len(args.pharm_sizes) != args.samples_per_pocket:
        raise ValueError("If pharm_sizes list is provided, must be of length sample per pocket")
    if args.ref_ligand_file is None and len(args.residue_list) == 0:
        raise ValueError('Either ref_ligand or residue_list must be specified')
    if args.ref_ligand_file is not None and len(args.residue_list) != 0:
        print("WARNING: Both reference ligand file and residue list specified. Reference ligand will be used to define pocket in this case.")
    return args


def main():
    import pharmacoforge_amd as pfa
    from pharmacoforge_amd.pocket_io import get_prot_atom_ph_type_maps, process_ligand_and_pocket

    args = parse_arguments()
    output_dir = Path(args.output_dir)
    output_dir.mkdir(exist_ok=True)
    if args.ckpt is not None:
        run_dir, model_file = args.ckpt.parent.parent, args.ckpt
    else:
        run_dir, model_file = args.model_dir, args.model_dir / 'checkpoints' / 'last.ckpt'
    config_file = run_dir / 'config.yaml'
    if not config_file.exists():
        config_file = run_dir / 'config.yml'
        if not config_file.exists():
            raise FileNotFoundError(f'config file not found in {run_dir}')
    with open(config_file, 'r') as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    if not torch.cuda.is_available():
        raise SystemExit("generate_pharmacophores.py needs an MI355X: the denoising kernels have no CPU fallback")
    device = torch.device('cuda')
    print(f'{device=}', flush=True)
    torch.manual_seed(args.seed)
    dataset_config = config['dataset']
    prot_element_map, ph_type_map = get_prot_atom_ph_type_maps(dataset_config)
    try:
        model = pfa.PharmacophoreDiff.load_from_checkpoint(model_file).to(device)
    except TypeError:
        model = pfa.PharmacophoreDiff.load_from_checkpoint(model_file, ph_type_map=config['dataset']['ph_type_map']).to(device)
    model.eval()

    rec_file, ref_lig_file = args.receptor_file, args.ref_ligand_file
    if not rec_file.exists():
        raise ValueError('receptor file does not exist')
    if ref_lig_file and not ref_lig_file.exists():
        raise ValueError('ligand file does not exist')
    rec_name = args.receptor_name or rec_file.name.split(".")[0]
    pocket_dir = output_dir / f'{rec_name}'
    pocket_dir.mkdir(exist_ok=True)
    ref_graph = process_ligand_and_pocket(rec_file, pocket_dir, lig_file=ref_lig_file, residue_list=args.residue_list,
                                          prot_element_map=prot_element_map, graph_cutoffs=config['graph']['graph_cutoffs'],
                                          pocket_cutoff=dataset_config['pocket_cutoff'], remove_hydrogen=True).to(device)
    ref_lig_com = ref_graph.pharm_x0 if args.use_ref_lig_com else None

    start = time.time()
    sampled_pharms = []
    while True:
        batch_size = min(args.samples_per_pocket - len(sampled_pharms), args.max_batch_size)
        pharm_sizes = args.pharm_sizes or model.pharm_size_dist.sample_uniformly(args.samples_per_pocket)
        g_batch = pfa.batch(pfa.copy_graph(ref_graph, batch_size, pharm_feats_per_copy=pharm_sizes))
        init_pharm_com = ref_lig_com.repeat(batch_size, 1) if args.use_ref_lig_com else None
        with torch.no_grad():
            sampled_pharms.extend(model.sample_given_receptor(g_batch, init_pharm_com=init_pharm_com,
                                                              visualize_trajectory=args.visualize_trajectory))
        if len(sampled_pharms) >= args.samples_per_pocket:
            break
    pocket_sample_time = time.time() - start
    with open(pocket_dir / 'sample_time.txt', 'w') as f:
        f.write(f'{pocket_sample_time:.2f}')
    with open(pocket_dir / 'sample_time.pkl', 'wb') as f:
        pickle.dump([pocket_sample_time], f)
    print(f'Pocket {rec_name} sampling time: {pocket_sample_time:.2f} seconds')
    print(f'Pocket {rec_name} sampling time per pharmacophore: {pocket_sample_time / len(sampled_pharms):.2f} seconds')
    ref_files_dir = pocket_dir / 'reference_files'
    ref_files_dir.mkdir(exist_ok=True)
    shutil.copy(rec_file, ref_files_dir / rec_file.name)
    if ref_lig_file is not None:
        shutil.copy(ref_lig_file, ref_files_dir / ref_lig_file.name)
    if args.visualize_trajectory:
        for i, ph in enumerate(sampled_pharms):
            ph.traj_to_xyz(pocket_dir / f'pharm_{i}_traj.xyz')
    else:
        with open(pocket_dir / 'pharms.xyz', 'w') as f:
            f.write(''.join(ph.to_xyz_file() for ph in sampled_pharms))
    if args.metrics:
        print(pfa.SampleAnalyzer().analyze(sampled_pharms))


if __name__ == "__main__":
    main()
