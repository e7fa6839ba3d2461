"""Receptor / ligand ingestion for the sampling CLI without Biopython / rdkit (neither exists on the GPU image).

Mirrors generate_pharmacophores.py:68-220 of the reference:

    parse_ligand                 :68-96    SDF (V2000 / V3000) -> atom coordinates, optional hydrogen removal
    element_fixer, onehot_encode_elements   :98-118
    process_ligand_and_pocket    :120-233  pocket = standard amino-acid residues with any atom closer than
                                           ``pocket_cutoff`` to any ligand atom (or an explicit chain:resseq list),
                                           hydrogens dropped, elements outside ``prot_elements`` dropped, static pp graph,
                                           one dummy pharmacophore node at the ligand / pocket centre, ``pocket.pdb`` written

The PDB reader follows the fixed-column PDB format (ATOM / HETATM records of the first MODEL); like Bio.PDB it keeps, for
an atom with alternate locations, the location with the highest occupancy, and takes the element from columns 77-78
(falling back to the atom name).  mmCIF receptors ('.mmcif' as in the reference, generate_pharmacophores.py:11,131-132;
'.cif' too) are read from the ``_atom_site`` loop with the column choices of Bio.PDB.MMCIFParser: author chain and
residue numbers (auth_asym_id / auth_seq_id), label_atom_id / label_comp_id names, pdbx_PDB_ins_code, label_alt_id,
type_symbol as the element, first pdbx_PDB_model_num only; their pocket.pdb records are formatted here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graph import PocketGraph, build_initial_complex_graph

# Bio.PDB.Polypeptide.is_aa(resname, standard=True): the 20 standard residues
STANDARD_AA = frozenset("ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL".split())


@dataclass
class Atom:
    name: str
    element: str
    coord: np.ndarray          # float32 [3]
    occupancy: float
    altloc: str
    line: str                  # the record as read (re-emitted into pocket.pdb)


@dataclass
class Residue:
    chain: str
    resseq: int
    icode: str
    resname: str
    hetero: bool
    atoms: List[Atom] = field(default_factory=list)

    @property
    def key(self):
        return (self.chain, self.resseq, self.icode, self.hetero)

    def get_atoms(self) -> List[Atom]:
        return self.atoms


def _guess_element(atom_name: str) -> str:
    """Fallback when columns 77-78 are blank: the element is right-justified in columns 13-14 of the 4-column atom-name
    field (' CA ' = carbon alpha, 'CA  ' = calcium); 4-character names ('HD11', '1HB ') are hydrogens / start with a digit."""
    name = (atom_name + "    ")[:4]
    if name[0] == " " or name[0].isdigit():
        return name[1].upper()
    if name[3] != " " and name[0] == "H":
        return "H"
    two = name[:2].strip().upper()
    return two if len(two) == 2 and two[1].isalpha() else name[0].upper()


def read_pdb(path) -> List[Residue]:
    """Residues of the first model, in file order."""
    residues: List[Residue] = []
    index: Dict[tuple, Residue] = {}
    with open(path, "r") as f:
        for line in f:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            line = line.rstrip("\n")
            name = line[12:16]
            altloc = line[16:17]
            resname = line[17:20].strip()
            chain = line[21:22]
            resseq = int(line[22:26])
            icode = line[26:27]
            xyz = np.array([float(line[30:38]), float(line[38:46]), float(line[46:54])], dtype=np.float32)
            occ = float(line[54:60]) if line[54:60].strip() else 1.0
            elem = line[76:78].strip().upper() if len(line) >= 78 and line[76:78].strip() else _guess_element(name)
            key = (chain, resseq, icode, rec == "HETATM")
            res = index.get(key)
            if res is None:
                res = Residue(chain, resseq, icode, resname, rec == "HETATM")
                index[key] = res
                residues.append(res)
            atom = Atom(name.strip(), elem, xyz, occ, altloc, line)
            prev = next((a for a in res.atoms if a.name == atom.name), None)
            if prev is None:
                res.atoms.append(atom)
            elif altloc != " " and occ > prev.occupancy:        # disordered atom: keep the most occupied location
                res.atoms[res.atoms.index(prev)] = atom
    return residues


def _cif_tokens(line: str) -> List[str]:
    """Whitespace-separated mmCIF values of one line; a value may be wrapped in single or double quotes (the closing
    quote is the one followed by whitespace or the end of the line, so O5' and "N" both survive)."""
    out, i, n = [], 0, len(line)
    while i < n:
        c = line[i]
        if c.isspace():
            i += 1
        elif c in "'\"" :
            j = i + 1
            while j < n and not (line[j] == c and (j + 1 == n or line[j + 1].isspace())):
                j += 1
            out.append(line[i + 1:j])
            i = j + 1
        elif c == "#":
            break
        else:
            j = i
            while j < n and not line[j].isspace():
                j += 1
            out.append(line[i:j])
            i = j
    return out


def _pdb_record(hetero: bool, serial: int, name: str, altloc: str, resname: str, chain: str, resseq: int, icode: str,
                xyz, occ: float, bfac: float, elem: str) -> str:
    """One fixed-column ATOM / HETATM record (atom names of fewer than four characters of a one-letter element start
    in column 14, as in wwPDB files)."""
    nm = name if len(name) >= 4 or len(elem) >= 2 else " " + name
    return (f"{'HETATM' if hetero else 'ATOM  '}{serial % 100000:>5} {nm:<4}{altloc:1}{resname:>3} {chain[:1]:1}{resseq:>4}{icode:1}   "
            f"{xyz[0]:8.3f}{xyz[1]:8.3f}{xyz[2]:8.3f}{occ:6.2f}{bfac:6.2f}          {elem:>2}")


def read_mmcif(path) -> List[Residue]:
    """Residues of the first model of an mmCIF file, in file order (the ``_atom_site`` loop; semantics as read_pdb)."""
    cols: List[str] = []
    rows: List[List[str]] = []
    in_header = in_rows = False
    with open(path, "r") as f:
        for raw in f:
            line = raw.rstrip("\n")
            st = line.strip()
            if not in_header and not in_rows:
                if st == "loop_":
                    cols, in_header = [], True
                continue
            if in_header:
                if st.startswith("_atom_site."):
                    cols.append(st.split()[0][len("_atom_site."):])
                    continue
                if st.startswith("_") or not cols:          # another category's loop
                    in_header = False
                    cols = []
                    if st == "loop_":
                        in_header = True
                    continue
                in_header, in_rows = False, True            # first data row of the _atom_site loop
            if in_rows:
                if not st or st.startswith("#") or st.startswith("_") or st == "loop_" or st.startswith("data_"):
                    break
                tok = _cif_tokens(line)
                if len(tok) != len(cols):
                    raise ValueError(f"{path}: _atom_site row with {len(tok)} values for {len(cols)} columns: {line!r}")
                rows.append(tok)
    if not rows:
        raise ValueError(f"{path}: no _atom_site loop found")
    ix = {c: i for i, c in enumerate(cols)}

    def col(row, *names, default=None):
        for nme in names:
            if nme in ix and row[ix[nme]] not in (".", "?"):
                return row[ix[nme]]
        return default
    residues: List[Residue] = []
    index: Dict[tuple, Residue] = {}
    first_model = col(rows[0], "pdbx_PDB_model_num", default="1")
    for serial, row in enumerate(rows, 1):
        if col(row, "pdbx_PDB_model_num", default=first_model) != first_model:
            break
        hetero = col(row, "group_PDB", default="ATOM") == "HETATM"
        name = col(row, "label_atom_id", "auth_atom_id", default="X")
        resname = col(row, "label_comp_id", "auth_comp_id", default="UNK")
        chain = col(row, "auth_asym_id", "label_asym_id", default=" ")
        seq = col(row, "auth_seq_id", "label_seq_id")
        if seq is None:
            continue                                        # Bio.PDB skips atoms without a residue number too
        resseq = int(seq)
        icode = col(row, "pdbx_PDB_ins_code", default=" ")
        altloc = col(row, "label_alt_id", default=" ")
        xyz = np.array([float(row[ix["Cartn_x"]]), float(row[ix["Cartn_y"]]), float(row[ix["Cartn_z"]])], dtype=np.float32)
        occ = float(col(row, "occupancy", default="1.0"))
        bfac = float(col(row, "B_iso_or_equiv", default="0.0"))
        elem = (col(row, "type_symbol") or _guess_element(name)).upper()
        key = (chain, resseq, icode, hetero)
        res = index.get(key)
        if res is None:
            res = Residue(chain, resseq, icode, resname, hetero)
            index[key] = res
            residues.append(res)
        atom = Atom(name, elem, xyz, occ, altloc, _pdb_record(hetero, serial, name, altloc, resname, chain, resseq, icode,
                                                              xyz, occ, bfac, elem.capitalize() if len(elem) > 1 else elem))
        prev = next((a for a in res.atoms if a.name == atom.name), None)
        if prev is None:
            res.atoms.append(atom)
        elif altloc != " " and occ > prev.occupancy:
            res.atoms[res.atoms.index(prev)] = atom
    return residues


def read_receptor(path) -> List[Residue]:
    """generate_pharmacophores.py:129-136: the parser follows the file suffix."""
    suffix = Path(path).suffix.lower()
    if suffix == '.pdb':
        return read_pdb(path)
    if suffix in ('.mmcif', '.cif'):
        return read_mmcif(path)
    raise ValueError(f'unsupported receptor file type: {suffix}, must be .pdb or .mmcif')


def is_aa(resname: str, standard: bool = True) -> bool:
    return resname.upper() in STANDARD_AA


def parse_ligand(ligand_path, remove_hydrogen: bool = False) -> Tuple[List[str], torch.Tensor]:
    """generate_pharmacophores.py:68-96: (elements, positions [N,3] float32) of the single molecule of an SDF file."""
    text = Path(ligand_path).read_text()
    blocks = [b for b in text.split("$$$$") if b.strip()]
    if len(blocks) > 1:
        raise NotImplementedError('Multiple ligands found. Code is not written to handle multiple ligands.')
    if not blocks:
        raise ValueError(f"no molecule in {ligand_path}")
    lines = blocks[0].splitlines()
    counts_i = next(i for i, l in enumerate(lines) if l.rstrip().endswith("V2000") or l.rstrip().endswith("V3000"))
    elems: List[str] = []
    pos: List[List[float]] = []
    if lines[counts_i].rstrip().endswith("V2000"):
        n_atoms = int(lines[counts_i][0:3])
        for l in lines[counts_i + 1: counts_i + 1 + n_atoms]:
            pos.append([float(l[0:10]), float(l[10:20]), float(l[20:30])])
            elems.append(l[31:34].strip())
    else:
        in_atoms = False
        for l in lines[counts_i + 1:]:
            if "BEGIN ATOM" in l:
                in_atoms = True
                continue
            if "END ATOM" in l:
                break
            if in_atoms:
                tok = l.split()            # M  V30 idx type x y z aamap ...
                elems.append(tok[3])
                pos.append([float(tok[4]), float(tok[5]), float(tok[6])])
    keep = [i for i, e in enumerate(elems) if not (remove_hydrogen and e.upper() in ("H", "D"))]
    if not keep:
        raise ValueError(f"ligand {ligand_path} has no atoms")
    return [elems[i] for i in keep], torch.tensor([pos[i] for i in keep], dtype=torch.float32)


def element_fixer(element: str) -> str:
    """generate_pharmacophores.py:98-103."""
    if len(element) > 1:
        element = element[0] + element[1:].lower()
    return element


def onehot_encode_elements(atom_elements: Iterable[str], element_map: Dict[str, int]) -> np.ndarray:
    """generate_pharmacophores.py:105-118 (unknown elements -> 'other')."""
    idx = np.fromiter((element_map.get(e, element_map['other']) for e in atom_elements), int)
    out = np.zeros((idx.size, len(element_map)))
    out[np.arange(idx.size), idx] = 1
    return out


def get_prot_atom_ph_type_maps(dataset_config: dict):
    """utils/unorganized_utils.py:97-106."""
    prot_element_map = {e: i for i, e in enumerate(dataset_config['prot_elements'])}
    prot_element_map['other'] = len(dataset_config['prot_elements'])
    ph_type_map = {e: i for i, e in enumerate(dataset_config['ph_type_map'])}
    return prot_element_map, ph_type_map


def write_pocket_file(residues: Sequence[Residue], path) -> None:
    """pocket.pdb: the ATOM/HETATM records of the selected residues (PocketSelector, receptor_utils.py:74-81)."""
    with open(path, "w") as f:
        for r in residues:
            for a in r.atoms:
                f.write(a.line + "\n")
        f.write("END\n")


def select_pocket_residues(residues: Sequence[Residue], lig_coords: torch.Tensor, pocket_cutoff: float) -> List[Residue]:
    """generate_pharmacophores.py:148-166: standard residues whose closest atom is < pocket_cutoff from the ligand."""
    lig = lig_coords.double().numpy()
    out = []
    for res in residues:
        if not is_aa(res.resname, standard=True):
            continue
        rc = np.array([a.coord for a in res.atoms], dtype=np.float64)
        d = np.sqrt(((lig[:, None, :] - rc[None, :, :]) ** 2).sum(-1))       # scipy.spatial.distance.cdist (euclidean)
        if d.min() < pocket_cutoff:
            out.append(res)
    return out


def process_ligand_and_pocket(rec_file: Path, output_dir: Optional[Path], prot_element_map: Dict[str, int],
                              graph_cutoffs: dict, pocket_cutoff: float, lig_file: Path = None, residue_list: list = [],
                              remove_hydrogen: bool = True, pp_edges=None) -> PocketGraph:
    """generate_pharmacophores.py:120-233 -> single-pocket PocketGraph with one dummy pharmacophore node at the
    ligand / pocket centre (``pharm_x0``).  The static pp radius graph is built on the GPU (pf_build_pp_edges) unless
    ``pp_edges`` is given."""
    rec_file = Path(rec_file)
    if lig_file is None and len(residue_list) == 0:
        raise ValueError("Either reference ligand or pocket residue list must be provided.")
    residues = read_receptor(rec_file)
    if lig_file is not None:
        _, lig_coords = parse_ligand(lig_file, remove_hydrogen=remove_hydrogen)
        init_com = lig_coords.mean(dim=0).reshape(1, 3)
        pocket_residues = select_pocket_residues(residues, lig_coords, pocket_cutoff)
        if len(pocket_residues) == 0:
            raise ValueError('no valid pocket residues found.')
    else:
        by_id = {(r.chain, r.resseq): r for r in residues if not r.hetero and r.icode == " "}
        pocket_residues = []
        for spec in residue_list:
            chain, idx = spec.split(':')
            if (chain, int(idx)) not in by_id:
                raise KeyError(f"residue {spec} not found in {rec_file}")
            pocket_residues.append(by_id[(chain, int(idx))])
        pc = torch.tensor(np.array([a.coord for r in pocket_residues for a in r.atoms]), dtype=torch.float32)
        init_com = pc.mean(dim=0).reshape(1, 3)
    atoms = [a for r in pocket_residues for a in r.atoms if not (remove_hydrogen and a.element == "H")]
    coords = torch.tensor(np.array([a.coord for a in atoms]), dtype=torch.float32)
    onehot = onehot_encode_elements([element_fixer(a.element) for a in atoms], prot_element_map)
    other = torch.tensor(onehot[:, -1] == 1)
    feats = torch.tensor(onehot[:, :-1]).float()
    coords, feats = coords[~other], feats[~other]
    g = build_initial_complex_graph(coords, feats, cutoffs=graph_cutoffs, pharm_atom_positions=init_com,
                                    pharm_atom_features=torch.zeros((1, 6)), pp_edges=pp_edges)
    if output_dir is not None:
        write_pocket_file(pocket_residues, Path(output_dir) / 'pocket.pdb')
    return g
