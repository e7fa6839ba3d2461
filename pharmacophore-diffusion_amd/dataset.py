"""Processed-dataset loader without DGL / rdkit.

Mirrors pharmacoforge/dataset/protein_pharm_dataset.py:19-179,268-276 and protein_pharmacophore_datamodule.py of the
reference on the on-disk layout that process_crossdocked.py:246-252 writes, so reference-processed data is usable as is:

    <processed_data_dir>/<anything ending in the split digit>/prot_pharm_tensors.npz
        prot_pos [Np_tot,3]  prot_feat [Np_tot] (element index)  prot_idx [G,2] (start, end)
        pharm_pos / pharm_feat / pharm_idx   prot_ph_pos / prot_ph_feat / prot_ph_idx   (same scheme)
        prot_file_names.pkl.gz  (optional here)   lig_rdmol.pkl.gz  (rdkit pickles: ignored)

``__getitem__`` returns a single-pocket ``PocketGraph`` (one-hot features, optional pharmacophore subsampling, static pp
radius graph); ``collate_fn`` batches them like ``dgl.batch``.

The reference rebuilds a pocket's pp radius graph with torch_cluster on the CPU at every access
(protein_pharm_dataset.py:163 -> :234-236).  Here the radius graphs of ALL pockets of the dataset are built once, in the
main process, by the library's batched radius kernel (pf_build_pp_edges: a chunk of pockets per launch) and kept as one
compact CSR (per-pocket edge ranges + pocket-local int16/int32 indices: ~4-8 bytes per edge instead of two int64 tensors
per pocket in a dict).  ``__getitem__`` then only slices host tensors, so ``DataLoader`` workers (forked after the
parent initialised HIP) never touch the GPU; a worker that finds the table missing raises instead of initialising HIP
in a forked child.
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

from .graph import PocketGraph, batch as batch_graphs, build_initial_complex_graph, radius_graph_pp

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
        self._pp_ptr = self._pp_src = self._pp_dst = None      # CSR of every pocket's pp edges (precompute_pp_edges)
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
        return build_initial_complex_graph(prot_pos.float(), prot_feat, cutoffs=self.graph_cutoffs,
                                           pharm_atom_positions=pharm_pos.float(), pharm_atom_features=pharm_feat,
                                           prot_ph_pos=prot_ph_pos.float(), prot_ph_feat=prot_ph_feat,
                                           pp_edges=self.pp_edges(i))

    # -- static pp edges of every pocket, built once ------------------------------------------------------------------
    def precompute_pp_edges(self, chunk: int = 2048):
        """Radius graphs (cutoff graph_cutoffs['pp'], max 100 neighbours: protein_pharm_dataset.py:234-236) of all
        pockets -> CSR.  Must run in the process that owns the GPU (CrossdockedDataModule.setup calls it)."""
        if self._pp_ptr is not None:
            return
        G = len(self)
        counts = torch.zeros(G, dtype=torch.int64)
        srcs, dsts = [], []
        starts = self.prot_idx[:, 0].long()
        sizes = (self.prot_idx[:, 1] - self.prot_idx[:, 0]).long()
        wide = int(sizes.max()) > 32767 if G else False
        idt = torch.int32 if wide else torch.int16
        cutoff = float(self.graph_cutoffs['pp'])
        for g0 in range(0, G, chunk):
            g1 = min(g0 + chunk, G)
            if cutoff <= 0:
                continue
            if self._pp_edges_fn is not None:             # caller-supplied builder: one pocket at a time
                for i in range(g0, g1):
                    rs, re_ = int(self.prot_idx[i, 0]), int(self.prot_idx[i, 1])
                    s_, d_ = self._pp_edges_fn(self.prot_pos[rs:re_].float())
                    srcs.append(s_.to(idt)); dsts.append(d_.to(idt)); counts[i] = s_.numel()
                continue
            # pockets g0..g1 as one batch of graphs (their atoms are contiguous in prot_pos when the (start, end) pairs
            # are; gather them otherwise)
            rows = torch.cat([torch.arange(int(starts[i]), int(starts[i] + sizes[i])) for i in range(g0, g1)]) \
                if not bool((starts[g0 + 1:g1] == (starts[g0:g1 - 1] + sizes[g0:g1 - 1])).all()) else \
                torch.arange(int(starts[g0]), int(starts[g1 - 1] + sizes[g1 - 1]))
            ptr = torch.zeros(g1 - g0 + 1, dtype=torch.int64)
            ptr[1:] = torch.cumsum(sizes[g0:g1], 0)
            s_, d_ = radius_graph_pp(self.prot_pos[rows].float(), ptr, cutoff, 100)
            gid = torch.searchsorted(ptr[1:].contiguous(), d_, right=True)        # edges come grouped by target
            counts[g0:g1] = torch.bincount(gid, minlength=g1 - g0)
            srcs.append((s_ - ptr[gid]).to(idt)); dsts.append((d_ - ptr[gid]).to(idt))
        self._pp_ptr = torch.zeros(G + 1, dtype=torch.int64)
        self._pp_ptr[1:] = torch.cumsum(counts, 0)
        self._pp_src = torch.cat(srcs) if srcs else torch.zeros(0, dtype=idt)
        self._pp_dst = torch.cat(dsts) if dsts else torch.zeros(0, dtype=idt)

    def pp_edges(self, i):
        if self._pp_ptr is None:
            if torch.utils.data.get_worker_info() is not None:
                raise RuntimeError("ProteinPharmacophoreDataset: the pp edge table was not built before the DataLoader "
                                   "workers were started (call dataset.precompute_pp_edges() / CrossdockedDataModule.setup() "
                                   "in the main process); a forked worker must not initialise the GPU")
            self.precompute_pp_edges()
        a, b = int(self._pp_ptr[i]), int(self._pp_ptr[i + 1])
        return self._pp_src[a:b].long(), self._pp_dst[a:b].long()

    def pp_edge_counts(self) -> torch.Tensor:
        """[G] pp edges per pocket: the work estimate used to deal pockets over GPUs (sharding.shard_by_work)."""
        if self._pp_ptr is None:
            self.precompute_pp_edges()
        return self._pp_ptr[1:] - self._pp_ptr[:-1]

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
        # the pp radius graphs of every pocket, once, in this (GPU-owning) process: DataLoader workers only slice them
        for ds in ((self.train_dataset,) if stage == 'fit' else ()) + (self.val_dataset,):
            ds.precompute_pp_edges()

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
