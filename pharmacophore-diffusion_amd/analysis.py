"""Output objects and sample metrics of the path (pharmacoforge/analysis/pharm_builder.py:7-71,
pharmacoforge/analysis/metrics.py:9-86, pharmacoforge/utils/unorganized_utils.py:111-128).
Host code on small per-pharmacophore tensors; the only multi-GPU exchange on the sampling path is
the all-reduce of the validity numerator/denominator and the type counts (8 numbers)."""
from pathlib import Path
from typing import List, Optional

import torch

ph_idx_to_type = ['Aromatic', 'HydrogenDonor', 'HydrogenAcceptor', 'PositiveIon', 'NegativeIon', 'Hydrophobic']


class SampledPharmacophore:
    type_idx_to_elem = ['P', 'S', 'F', 'N', 'O', 'C']

    def __init__(self, g, pharm_type_map: List[str], traj_frames=None, ref_prot_file: Path = None, ref_rdkit_lig=None):
        self.g = g                                       # single-graph PocketGraph holding the final x_0 / h_0
        self.pharm_type_map = pharm_type_map
        self.ref_prot_file, self.ref_rdkit_lig = ref_prot_file, ref_rdkit_lig
        self.ph_coords = g.pharm_x0
        self.ph_feats_idxs = g.pharm_h0.argmax(dim=1)
        self.ph_types = [pharm_type_map[int(i)] for i in self.ph_feats_idxs]
        self.n_ph_centers = self.ph_coords.shape[0]
        self.pos_frames, self.feat_frames = (None, None) if traj_frames is None else traj_frames
        assert len(pharm_type_map) == len(self.type_idx_to_elem), \
            f"pharm_type_map must have {len(self.type_idx_to_elem)} elements"
        self.ph_type_to_elem = {pharm_type_map[i]: self.type_idx_to_elem[i] for i in range(len(pharm_type_map))}

    def pharm_to_xyz(self, pos: torch.Tensor, types: List[str]):
        out = f'{len(pos)}\n'
        for i in range(len(pos)):
            out += f"{self.ph_type_to_elem[types[i]]} {pos[i, 0]:.3f} {pos[i, 1]:.3f} {pos[i, 2]:.3f}\n"
        return out

    def to_xyz_file(self, filename: str = None):
        out = self.pharm_to_xyz(self.ph_coords, self.ph_types)
        if filename is None:
            return out
        with open(filename, 'w') as f:
            f.write(out)

    def traj_to_xyz(self, filename: str = None):
        if self.pos_frames is None:
            raise ValueError("Cannot write trajectory because no trajectory frames were passed to the SampledPharmacophore object")
        out = ""
        frame_type_idxs = self.feat_frames.argmax(dim=2)
        for i in range(self.pos_frames.shape[0]):
            out += self.pharm_to_xyz(self.pos_frames[i], [self.pharm_type_map[int(j)] for j in frame_type_idxs[i]])
        if filename is None:
            return out
        with open(filename, 'w') as f:
            f.write(out)


def write_pharmacophore_file(coords_list, atom_types_list, pharm_type_map: list, filename: str = None):
    elem = ['P', 'S', 'F', 'N', 'O', 'C']
    out = ""
    for coords, atom_types in zip(coords_list, atom_types_list):
        assert len(coords) == len(atom_types)
        out += f"{len(coords)}\n"
        for i in range(len(coords)):
            out += f"{elem[atom_types[i]]} {coords[i, 0]:.3f} {coords[i, 1]:.3f} {coords[i, 2]:.3f}\n"
    if filename is None:
        return out
    with open(filename, 'w') as f:
        f.write(out)


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
