"""Output objects and sample metrics of the path (pharmacoforge/analysis/pharm_builder.py:7-71,
pharmacoforge/analysis/metrics.py:9-86, pharmacoforge/utils/unorganized_utils.py:111-128).
Host code on small per-pharmacophore tensors; the only multi-GPU exchange on the sampling path is
the all-reduce of the validity numerator/denominator and the type counts (8 numbers)."""
from pathlib import Path
from typing import List, Optional

import torch

ph_idx_to_type = ['Aromatic', 'HydrogenDonor', 'HydrogenAcceptor', 'PositiveIon', 'NegativeIon', 'Hydrophobic']


_XYZ_ELEMENTS = ('P', 'S', 'F', 'N', 'O', 'C')      # file format of pharms.xyz: one pseudo-element per pharmacophore type


def _xyz_block(elements, coords) -> str:
    """One xyz frame: the center count, then `element x y z` with three decimals per center (the text
    pharm_builder.py:35-42 / unorganized_utils.py:111-128 emit; this is the file format)."""
    rows = torch.as_tensor(coords, dtype=torch.float64).reshape(-1, 3).tolist()
    if len(rows) != len(elements):
        raise ValueError(f"{len(elements)} element symbols for {len(rows)} positions")
    body = "".join(f"{e} {x:.3f} {y:.3f} {z:.3f}\n" for e, (x, y, z) in zip(elements, rows))
    return f"{len(rows)}\n" + body


def _emit(text: str, filename):
    """Return the text, or write it when a file name is given (the reference's writers do the same)."""
    if filename is None:
        return text
    Path(filename).write_text(text)
    return None


class SampledPharmacophore:
    """One generated pharmacophore: the final graph, its argmax feature types and, optionally, the frames of its
    reverse-diffusion trajectory.  Attribute and method names follow pharmacoforge/analysis/pharm_builder.py:7-71
    (callers and the metrics read them); the bodies are this repository's."""
    type_idx_to_elem = list(_XYZ_ELEMENTS)

    def __init__(self, g, pharm_type_map: List[str], traj_frames=None, ref_prot_file: Path = None, ref_rdkit_lig=None):
        if len(pharm_type_map) != len(_XYZ_ELEMENTS):
            raise AssertionError(f"a pharmacophore type map of {len(_XYZ_ELEMENTS)} entries is required, got {len(pharm_type_map)}")
        self.g = g                                       # single-graph PocketGraph with x_0 / h_0 of the centers
        self.pharm_type_map = list(pharm_type_map)
        self.ph_type_to_elem = dict(zip(self.pharm_type_map, _XYZ_ELEMENTS))
        self.ref_prot_file, self.ref_rdkit_lig = ref_prot_file, ref_rdkit_lig
        self.ph_coords = g.pharm_x0
        self.ph_feats_idxs = torch.argmax(g.pharm_h0, dim=1)
        self.ph_types = self._names(self.ph_feats_idxs)
        self.n_ph_centers = int(self.ph_coords.shape[0])
        self.pos_frames, self.feat_frames = traj_frames if traj_frames is not None else (None, None)

    def _names(self, type_indices) -> List[str]:
        return [self.pharm_type_map[k] for k in torch.as_tensor(type_indices).tolist()]

    def pharm_to_xyz(self, pos: torch.Tensor, types: List[str]) -> str:
        return _xyz_block([self.ph_type_to_elem[t] for t in types], pos)

    def to_xyz_file(self, filename: str = None):
        return _emit(self.pharm_to_xyz(self.ph_coords, self.ph_types), filename)

    def traj_to_xyz(self, filename: str = None):
        if self.pos_frames is None or self.feat_frames is None:
            raise ValueError("this SampledPharmacophore holds no trajectory: sample with visualize_trajectory=True to record one")
        per_frame_types = torch.argmax(self.feat_frames, dim=2)
        frames = (self.pharm_to_xyz(x, self._names(k)) for x, k in zip(self.pos_frames, per_frame_types))
        return _emit("".join(frames), filename)


def write_pharmacophore_file(coords_list, atom_types_list, pharm_type_map: list, filename: str = None):
    """Several pharmacophores, given as coordinates and type INDICES, as consecutive xyz frames (unorganized_utils.py:111-128)."""
    frames = []
    for coords, type_indices in zip(coords_list, atom_types_list):
        frames.append(_xyz_block([_XYZ_ELEMENTS[int(k)] for k in type_indices], coords))
    return _emit("".join(frames), filename)


_MATCHING_TYPES = {'Aromatic': ['Aromatic', 'PositiveIon'], 'HydrogenDonor': ['HydrogenAcceptor'],
                   'HydrogenAcceptor': ['HydrogenDonor'], 'PositiveIon': ['NegativeIon', 'Aromatic'],
                   'NegativeIon': ['PositiveIon'], 'Hydrophobic': ['Hydrophobic']}
_MATCHING_DIST = {"Aromatic": 7, "Hydrophobic": 5, "HydrogenAcceptor": 4, "HydrogenDonor": 4, "NegativeIon": 5,
                  "PositiveIon": 5}


def compute_complementarity(pharm_types, pharm_pos, prot_ph_types, prot_ph_pos, return_count=False):
    """metrics.py:53-86: number (or fraction) of centers with a complementary receptor feature within
    the type-specific distance."""
    if len(prot_ph_types) == 0 or len(pharm_types) == 0:
        count = torch.tensor(0)
    else:
        distances = torch.cdist(pharm_pos, prot_ph_pos)
        lim = torch.tensor([_MATCHING_DIST[t] for t in pharm_types], dtype=distances.dtype).reshape(-1, 1)
        matching = torch.tensor([[r in _MATCHING_TYPES[p] for r in prot_ph_types] for p in pharm_types])
        count = ((distances <= lim) & matching).any(dim=1).sum()
    if return_count:
        return count
    return count / max(len(pharm_types), 1)          # (the reference divides by an undefined name here, metrics.py:85)


def _allreduce_sum(counts: torch.Tensor, process_group, device=None) -> torch.Tensor:
    """Sum a small host vector over the ranks.  The buffer lives where the group's backend needs it: RCCL ("nccl")
    reduces device memory only, so the buffer goes to ``device`` (default: this process's current GPU); gloo reduces the
    host tensor directly."""
    import torch.distributed as dist
    if dist.get_backend(process_group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        buf = counts.to(dev)
    else:
        buf = counts.clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
    return buf.cpu()


class SampleAnalyzer:
    def analyze(self, sample: List[SampledPharmacophore], process_group=None, device=None):
        """metrics.py:9-35.  With ``process_group`` the numerator / denominator are all-reduced (sum)
        before the division, so every rank reports the validity of the whole job."""
        num = den = 0
        for ph in sample:
            g = ph.g
            if g.prot_ph_h is None or g.prot_ph_x is None:
                prot_types, prot_pos = [], torch.zeros(0, 3)
            else:
                prot_types = [ph_idx_to_type[int(i)] for i in g.prot_ph_h.argmax(dim=1)]
                prot_pos = g.prot_ph_x
            num += int(compute_complementarity(ph.ph_types, ph.ph_coords, prot_types, prot_pos, return_count=True))
            den += ph.n_ph_centers
        counts = torch.tensor([float(num), float(den)], dtype=torch.float64)
        if process_group is not None:
            counts = _allreduce_sum(counts, process_group, device)
        return {'validity': float(counts[0] / counts[1]) if counts[1] > 0 else 0.0}

    def pharm_feat_freq(self, sample: List[SampledPharmacophore], process_group=None, device=None):
        """metrics.py:37-51: counts of each predicted feature type (optionally summed over ranks)."""
        counts = torch.zeros(6, dtype=torch.float64)
        for ph in sample:
            for val in ph.ph_feats_idxs:
                counts[int(val)] += 1
        if process_group is not None:
            counts = _allreduce_sum(counts, process_group, device)
        return counts
