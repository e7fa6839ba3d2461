#!/usr/bin/env python3
"""Pharmacophore sampling for one receptor pocket on the MI355X library.

Command-line contract of the reference's generate_pharmacophores.py (flag names, types and defaults of its :29-46; the
run-directory convention <run>/config.y[a]ml + <run>/checkpoints/last.ckpt; the output tree
<output_dir>/<receptor>/{pharms.xyz | pharm_<i>_traj.xyz, pocket.pdb, sample_time.txt, sample_time.pkl,
reference_files/}), so scripts written against the reference keep working.  Everything behind the flags is this
repository's: receptor parsing without Biopython / rdkit (pharmacoforge_amd.pocket_io: PDB or mmCIF receptor, SDF
ligand), PharmacophoreDiff.sample_given_receptor -> pf_sample (the whole reverse process enqueued on the GPU)."""
import argparse
import pickle
import shutil
import sys
import time
from pathlib import Path

import torch
import yaml

# (flag, argparse keywords): names / types / defaults are the reference's, the descriptions are ours
_FLAGS = [
    ('receptor_file', dict(type=Path, help='receptor structure, .pdb or .cif / .mmcif')),
    ('--ref_ligand_file', dict(type=Path, help='SDF ligand; residues with an atom within the config\'s pocket_cutoff of it form the pocket')),
    ('--residue_list', dict(nargs='+', type=str, default=[], help='pocket given explicitly instead: entries CHAIN:RESSEQ, e.g. A:105 A:106')),
    ('--ckpt', dict(type=Path, default=None, help='checkpoint file at <run>/checkpoints/<name>.ckpt (the config is looked up in <run>)')),
    ('--model_dir', dict(type=Path, default=None, help='training run directory <run>; its checkpoints/last.ckpt is loaded')),
    ('--samples_per_pocket', dict(type=int, default=1, help='how many pharmacophores to draw')),
    ('--pharm_sizes', dict(nargs='+', type=int, default=[], help='centers per pharmacophore, one integer per sample (default: uniform 3..8)')),
    ('--output_dir', dict(type=str, default='generated_pharms/', help='root of the output tree')),
    ('--receptor_name', dict(type=str, default=None, help='sub-directory name (default: receptor file stem)')),
    ('--max_batch_size', dict(type=int, default=128, help='graphs per device batch')),
    ('--seed', dict(type=int, default=42, help='torch seed of the run')),
    ('--use_ref_lig_com', dict(action='store_true', help='start the centers at the ligand\'s mean position instead of the pocket\'s')),
    ('--visualize_trajectory', dict(action='store_true', help='write every denoising frame, one xyz file per sample')),
    ('--metrics', dict(action='store_true', help='print the validity of the samples against receptor pharmacophore features')),
]


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    a = parser.parse_args(argv)
    problems = []
    if (a.ckpt is None) == (a.model_dir is None):
        problems.append('give exactly one of --ckpt and --model_dir')
    if a.pharm_sizes and len(a.pharm_sizes) != a.samples_per_pocket:
        problems.append(f'--pharm_sizes lists {len(a.pharm_sizes)} sizes for --samples_per_pocket {a.samples_per_pocket}')
    if a.ref_ligand_file is None and not a.residue_list:
        problems.append('the pocket is undefined: pass --ref_ligand_file or --residue_list')
    if problems:
        raise ValueError('; '.join(problems))
    if a.ref_ligand_file is not None and a.residue_list:
        print('note: --residue_list is ignored because --ref_ligand_file defines the pocket', file=sys.stderr)
    return a


def locate_run(args):
    """-> (run directory, checkpoint file, parsed config)."""
    if args.ckpt is not None:
        run_dir, ckpt = args.ckpt.parent.parent, args.ckpt
    else:
        run_dir, ckpt = args.model_dir, args.model_dir / 'checkpoints' / 'last.ckpt'
    for name in ('config.yaml', 'config.yml'):
        if (run_dir / name).exists():
            with open(run_dir / name) as f:
                return run_dir, ckpt, yaml.load(f, Loader=yaml.FullLoader)
    raise FileNotFoundError(f'{run_dir} holds neither config.yaml nor config.yml')


def load_model(pfa, ckpt, config, device):
    try:
        model = pfa.PharmacophoreDiff.load_from_checkpoint(ckpt)
    except TypeError:              # checkpoints written before ph_type_map became a hyper-parameter
        model = pfa.PharmacophoreDiff.load_from_checkpoint(ckpt, ph_type_map=config['dataset']['ph_type_map'])
    return model.to(device).eval()


def main(argv=None):
    import pharmacoforge_amd as pfa
    from pharmacoforge_amd.pocket_io import get_prot_atom_ph_type_maps, process_ligand_and_pocket

    args = parse_arguments(argv)
    for path, what in ((args.receptor_file, 'receptor'), (args.ref_ligand_file, 'ligand')):
        if path is not None and not path.exists():
            raise ValueError(f'{what} file {path} not found')
    if not torch.cuda.is_available():
        raise SystemExit("generate_pharmacophores.py needs an MI355X: the denoising kernels have no CPU fallback")
    _, ckpt, config = locate_run(args)
    device = torch.device('cuda')
    print(f'{device=}', flush=True)
    torch.manual_seed(args.seed)
    prot_element_map, _ = get_prot_atom_ph_type_maps(config['dataset'])
    model = load_model(pfa, ckpt, config, device)

    name = args.receptor_name or args.receptor_file.name.split('.')[0]
    pocket_dir = Path(args.output_dir) / name
    pocket_dir.mkdir(parents=True, exist_ok=True)
    pocket = process_ligand_and_pocket(args.receptor_file, pocket_dir, lig_file=args.ref_ligand_file,
                                       residue_list=args.residue_list, prot_element_map=prot_element_map,
                                       graph_cutoffs=config['graph']['graph_cutoffs'],
                                       pocket_cutoff=config['dataset']['pocket_cutoff'], remove_hydrogen=True).to(device)

    t0 = time.time()
    pharms = []
    while len(pharms) < args.samples_per_pocket:
        n = min(args.samples_per_pocket - len(pharms), args.max_batch_size)
        sizes = args.pharm_sizes or model.pharm_size_dist.sample_uniformly(args.samples_per_pocket)
        # like the reference (:329-333, utils/unorganized_utils.py:48) every chunk reads the size list from its start
        copies = pfa.batch(pfa.copy_graph(pocket, n, pharm_feats_per_copy=sizes))
        com = pocket.pharm_x0.repeat(n, 1) if args.use_ref_lig_com else None
        with torch.no_grad():
            pharms += model.sample_given_receptor(copies, init_pharm_com=com, visualize_trajectory=args.visualize_trajectory)
    elapsed = time.time() - t0

    (pocket_dir / 'sample_time.txt').write_text(f'{elapsed:.2f}')
    with open(pocket_dir / 'sample_time.pkl', 'wb') as f:
        pickle.dump([elapsed], f)
    print(f'{name}: {len(pharms)} pharmacophores in {elapsed:.2f} s ({elapsed / len(pharms):.3f} s each)')
    ref_dir = pocket_dir / 'reference_files'
    ref_dir.mkdir(exist_ok=True)
    for src in (args.receptor_file, args.ref_ligand_file):
        if src is not None:
            shutil.copy(src, ref_dir / src.name)
    if args.visualize_trajectory:
        for i, ph in enumerate(pharms):
            ph.traj_to_xyz(pocket_dir / f'pharm_{i}_traj.xyz')
    else:
        (pocket_dir / 'pharms.xyz').write_text(''.join(ph.to_xyz_file() for ph in pharms))
    if args.metrics:
        print(pfa.SampleAnalyzer().analyze(pharms))


if __name__ == "__main__":
    main()
