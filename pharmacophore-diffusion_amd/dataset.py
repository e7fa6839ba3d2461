"""Processed-dataset loader without DGL / rdkit.

Mirrors pharmacoforge/dataset/protein_pharm_dataset.py:19-179,268-276 and protein_pharmacophore_datamodule.py of the
reference on the on-disk layout that process_crossdocked.py:246-252 writes, so reference-processed data is usable as is:

    <processed_data_dir>/<anything ending in the split digit>/prot_pharm_tensors.npz
        prot_pos [Np_tot,3]  prot_feat [Np_tot] (element index)  prot_idx [G,2] (start, end)
        pharm_pos / pharm_feat / pharm_idx   prot_ph_pos / prot_ph_feat / prot_ph_idx   (same scheme)
        prot_file_names.pkl.gz  (optional here)   lig_rdmol.pkl.gz  (rdkit pickles: ignored)

``__getitem__`` returns a single-pocket ``PocketGraph`` (one-hot features, optional pharmacophore subsampling, static pp
radius graph); ``collate_fn`` batches them like ``dgl.batch``.  The pp radius graph of a pocket is cached after its first
construction (the reference rebuilds it on every access).
"""
from __future__ import annotations

import gzip
import pickle
import random
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch
from torch.nn.functional import one_hot

from .graph import PocketGraph, batch as batch_graphs, build_initial_complex_graph

_KEYS = ("pharm_pos", "pharm_feat", "prot_pos", "prot_feat", "pharm_idx", "prot_idx", "prot_ph_feat", "prot_ph_pos",
         "prot_ph_idx")


class ProteinPharmacophoreDataset(torch.utils.data.Dataset):
    def __init__(self, name: str, split_idxs: List[int], raw_data_dir: str, processed_data_dir: str, graph_cutoffs: dict,
                 prot_elements: List[str], ph_type_map: List[str], subsample_pharms: bool = False, subsample_min: int = 3,
                 subsample_max: int = 9, pp_edges_fn=None, **kwargs):
        self.name = name
        self.graph_cutoffs, self.prot_elements, self.ph_type_map = graph_cutoffs, prot_elements, ph_type_map
        self.raw_data_dir = raw_data_dir
        self.subsample_pharms, self.subsample_min, self.subsample_max = subsample_pharms, subsample_min, subsample_max
        self.processed_data_dir = Path(processed_data_dir)
        if not self.processed_data_dir.exists():
            raise FileNotFoundError(f'Could not find processed data directory at {self.processed_data_dir}')
        self._pp_edges_fn = pp_edges_fn           # (prot_pos) -> (src, dst); default: the library's GPU radius graph
        self._pp_cache: Dict[int, tuple] = {}
        arrs = {k: [] for k in _KEYS}
        self.prot_file_names: List[str] = []
        for split_dir in sorted(self.processed_data_dir.iterdir()):
            if not split_dir.is_dir():
                continue
            split_idx = int(split_dir.name.split('_')[-1][-1])            # protein_pharm_dataset.py:70
            if split_idx not in split_idxs:
                continue
            names = split_dir / 'prot_file_names.pkl.gz'
            if names.exists():
                with gzip.open(names, 'rb') as f:
                    self.prot_file_names.extend(pickle.load(f))
            data = np.load(split_dir / 'prot_pharm_tensors.npz')
            for k in _KEYS:
                arrs[k].append(data[k])
        if not arrs["prot_idx"]:
            raise FileNotFoundError(f'no split directories for splits {split_idxs} in {self.processed_data_dir}')
        for k in ("pharm_pos", "pharm_feat", "prot_pos", "prot_feat", "prot_ph_feat", "prot_ph_pos"):
            setattr(self, k, torch.from_numpy(np.concatenate(arrs[k], axis=0)))
        for k in ("pharm_idx", "prot_idx", "prot_ph_idx"):               # make the per-file (start, end) pairs global
            out, off = [], 0
            for a in arrs[k]:
                out.append(a + off)
                if len(a):
                    off = int(out[-1][-1, 1])
            setattr(self, k, torch.from_numpy(np.concatenate(out, axis=0)))

    def __len__(self):
        return self.prot_idx.shape[0]

    def __getitem__(self, i) -> PocketGraph:
        ps, pe = (int(v) for v in self.pharm_idx[i])
        rs, re_ = (int(v) for v in self.prot_idx[i])
        hs, he = (int(v) for v in self.prot_ph_idx[i])
        pharm_pos, prot_pos, prot_ph_pos = self.pharm_pos[ps:pe], self.prot_pos[rs:re_], self.prot_ph_pos[hs:he]
        prot_feat = one_hot(self.prot_feat[rs:re_].long(), num_classes=len(self.prot_elements)).float()
        pharm_feat = one_hot(self.pharm_feat[ps:pe].long(), num_classes=len(self.ph_type_map)).float()
        prot_ph_feat = one_hot(self.prot_ph_feat[hs:he].long(), num_classes=len(self.ph_type_map)).float()
        if self.subsample_pharms and len(pharm_pos) > self.subsample_min - 1:      # protein_pharm_dataset.py:151-161
            smax = min(self.subsample_max, len(pharm_pos))
            n = self.subsample_min if self.subsample_min == smax else random.randint(self.subsample_min, smax)
            idx = random.sample(range(len(pharm_pos)), n)
            pharm_pos, pharm_feat = pharm_pos[idx], pharm_feat[idx]
        if i not in self._pp_cache:
            if self._pp_edges_fn is not None:
                self._pp_cache[i] = self._pp_edges_fn(prot_pos.float())
            else:
                g0 = build_initial_complex_graph(prot_pos.float(), prot_feat, cutoffs=self.graph_cutoffs)
                self._pp_cache[i] = (g0.pp_src, g0.pp_dst)
        return build_initial_complex_graph(prot_pos.float(), prot_feat, cutoffs=self.graph_cutoffs,
                                           pharm_atom_positions=pharm_pos.float(), pharm_atom_features=pharm_feat,
                                           prot_ph_pos=prot_ph_pos.float(), prot_ph_feat=prot_ph_feat,
                                           pp_edges=self._pp_cache[i])

    def get_files(self, idx: int):
        return self.raw_data_dir, (self.prot_file_names[idx] if self.prot_file_names else None), None


def collate_fn(complex_graphs: List[PocketGraph]) -> PocketGraph:
    return batch_graphs(complex_graphs)


def get_dataloader(dataset, batch_size: int, num_workers: int = 0, **kwargs):
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, drop_last=False, num_workers=num_workers,
                                       collate_fn=collate_fn, **kwargs)


class CrossdockedDataModule:
    """protein_pharmacophore_datamodule.py:16-66 (train/val split by split index)."""

    def __init__(self, dataset_config: dict, batch_size: int, num_workers: int, validation_splits: List[int] = []):
        if len(validation_splits) == 0:
            raise NotImplementedError("training without a validation split has not yet been implemented")
        if len(validation_splits) >= 3:
            raise ValueError("validation split indices must be a subset of [0, 1, 2]")
        for s in validation_splits:
            if s not in [0, 1, 2]:
                raise ValueError("validation split index must be 0, 1, or 2")
        self.dataset_config, self.batch_size, self.num_workers = dataset_config, batch_size, num_workers
        self.train_split_idxs = [s for s in (0, 1, 2) if s not in validation_splits]
        self.val_split_idxs = [s for s in (0, 1, 2) if s in validation_splits]

    def setup(self, stage: str = 'fit'):
        if stage == 'fit':
            self.train_dataset = ProteinPharmacophoreDataset(name='train', split_idxs=self.train_split_idxs, **self.dataset_config)
        self.val_dataset = ProteinPharmacophoreDataset(name='val', split_idxs=self.val_split_idxs, **self.dataset_config)

    def train_dataloader(self, **kw):
        return get_dataloader(self.train_dataset, self.batch_size, self.num_workers, **kw)

    def val_dataloader(self, **kw):
        return get_dataloader(self.val_dataset, self.batch_size, self.num_workers, **kw)


def data_module_from_config(config: dict) -> CrossdockedDataModule:
    """config_utils/load_from_config.py:34-45."""
    dataset_config = dict(config['dataset'])
    dataset_config['graph_cutoffs'] = config['graph']['graph_cutoffs']
    return CrossdockedDataModule(dataset_config=dataset_config, batch_size=config['training']['batch_size'],
                                 num_workers=config['training']['num_workers'],
                                 validation_splits=config['training']['validation_splits'])
