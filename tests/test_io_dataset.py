"""CPU tests of the receptor / ligand ingestion (pocket_io) and the processed-dataset loader (dataset): the semantics of
generate_pharmacophores.py:68-233 and protein_pharm_dataset.py:19-179 on hand-written files."""
import gzip
import os
import pickle

import numpy as np
import pytest
import torch

import pharmacoforge_amd as pfa
from pharmacoforge_amd import pocket_io as P
from pharmacoforge_amd import dataset as D
from oracle import pf_oracle as O

PROT_ELEMENTS = ['C', 'N', 'O', 'S', 'P', 'F', 'Cl', 'Br', 'I', 'B', 'D']
CUTOFFS = {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9}


def pdb_line(rec, serial, name, alt, resn, chain, resi, x, y, z, occ, elem):
    name_f = name if len(name) == 4 else " " + name.ljust(3)
    return f"{rec:<6}{serial:>5} {name_f}{alt}{resn:>3} {chain}{resi:>4}    {x:8.3f}{y:8.3f}{z:8.3f}{occ:6.2f}{20.0:6.2f}          {elem:>2}"


def make_pdb(path):
    L = []
    s = 1
    # residue A:1 ALA near the ligand; one hydrogen; CB has two alternate locations (B more occupied)
    for name, xyz, elem, alt, occ in [("N", (0, 0, 0), "N", " ", 1.0), ("CA", (1.4, 0, 0), "C", " ", 1.0), ("C", (2.0, 1.3, 0), "C", " ", 1.0),
                                      ("O", (1.4, 2.3, 0), "O", " ", 1.0), ("CB", (2.0, -1.2, 0.5), "C", "A", 0.3),
                                      ("CB", (2.1, -1.1, 0.6), "C", "B", 0.7), ("HA", (1.5, 0.1, 1.0), "H", " ", 1.0)]:
        L.append(pdb_line("ATOM", s, name, alt, "ALA", "A", 1, *xyz, occ, elem)); s += 1
    # residue A:2 MET with a selenium (not in prot_elements -> 'other' -> dropped), 6 A away: still inside the 8 A pocket
    for name, xyz, elem in [("N", (6, 0, 0), "N"), ("CA", (7.4, 0, 0), "C"), ("SE", (7.5, 1.5, 0), "SE")]:
        L.append(pdb_line("ATOM", s, name, " ", "MET", "A", 2, *xyz, 1.0, elem)); s += 1
    # residue A:3 GLY far away (30 A): outside
    for name, xyz, elem in [("N", (30, 0, 0), "N"), ("CA", (31.4, 0, 0), "C")]:
        L.append(pdb_line("ATOM", s, name, " ", "GLY", "A", 3, *xyz, 1.0, elem)); s += 1
    # a water and a non-standard residue right next to the ligand: never pocket residues (is_aa(standard=True))
    L.append(pdb_line("HETATM", s, "O", " ", "HOH", "A", 101, 0.5, 0.5, 0.5, 1.0, "O")); s += 1
    L.append(pdb_line("HETATM", s, "CA", " ", "MSE", "A", 4, 0.2, 0.2, 0.2, 1.0, "C")); s += 1
    L.append("ENDMDL")
    L.append(pdb_line("ATOM", s, "N", " ", "ALA", "B", 9, 0.1, 0.1, 0.1, 1.0, "N"))      # second model: ignored
    path.write_text("\n".join(L) + "\nEND\n")


SDF = """lig
  test

  4  3  0  0  0  0  0  0  0  0999 V2000
    0.5000    0.5000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0
    1.5000    0.5000    0.0000 O   0  0  0  0  0  0  0  0  0  0  0  0
    0.5000    1.5000    0.0000 N   0  0  0  0  0  0  0  0  0  0  0  0
    0.5000    0.5000    1.0000 H   0  0  0  0  0  0  0  0  0  0  0  0
  1  2  1  0
  1  3  1  0
  1  4  1  0
M  END
$$$$
"""


def oracle_pp(pos):
    e = O.radius_graph(pos, 3.5, torch.tensor([0, pos.shape[0]]), 100)
    return e[0], e[1]


def test_pdb_sdf_pocket(tmp_path):
    pdb, sdf = tmp_path / "rec.pdb", tmp_path / "lig.sdf"
    make_pdb(pdb)
    sdf.write_text(SDF)
    res = P.read_pdb(pdb)
    assert [(r.resname, r.chain, r.resseq, r.hetero) for r in res] == [("ALA", "A", 1, False), ("MET", "A", 2, False),
                                                                         ("GLY", "A", 3, False), ("HOH", "A", 101, True), ("MSE", "A", 4, True)]
    cb = [a for a in res[0].atoms if a.name == "CB"]
    assert len(cb) == 1 and cb[0].altloc == "B" and np.allclose(cb[0].coord, [2.1, -1.1, 0.6])
    elems, pos = P.parse_ligand(sdf, remove_hydrogen=True)
    assert elems == ["C", "O", "N"] and pos.shape == (3, 3)
    assert P.parse_ligand(sdf)[1].shape == (4, 3)
    emap, tmap = P.get_prot_atom_ph_type_maps({'prot_elements': PROT_ELEMENTS, 'ph_type_map': pfa.analysis.ph_idx_to_type})
    assert emap['other'] == 11 and tmap['Hydrophobic'] == 5
    # the radius graph needs the GPU library; here the oracle's radius graph is injected
    all_pos = torch.tensor(np.array([a.coord for r in res[:2] for a in r.atoms if a.element not in ("H", "SE")]))
    g = P.process_ligand_and_pocket(pdb, tmp_path, emap, CUTOFFS, 8.0, lig_file=sdf, pp_edges=oracle_pp(all_pos))
    assert g.num_nodes('prot') == 7                        # ALA: N CA C O CB (no H), MET: N CA (SE dropped); GLY / HOH / MSE out
    assert torch.allclose(g.prot_x, all_pos)
    assert g.prot_h.shape == (7, 11) and g.prot_h.sum(1).eq(1).all()
    assert g.prot_h[:, :3].sum(0).tolist() == [4.0, 2.0, 1.0]            # 4 C, 2 N, 1 O
    assert torch.allclose(g.pharm_x0, pos.mean(0, keepdim=True)) and g.pharm_h0.shape == (1, 6)
    # value level, written out by hand (generate_pharmacophores.py:148-220 on these files): the atom set in file order, its
    # one-hots over prot_elements, the ligand centre, and the static pp edges -- every pair closer than 3.5 A, both directions
    # (O -- CB of the alanine are 3.523 A apart: no edge; the methionine's N -- CA are the only pair of the second residue)
    want_xyz = [(0, 0, 0), (1.4, 0, 0), (2.0, 1.3, 0), (1.4, 2.3, 0), (2.1, -1.1, 0.6), (6, 0, 0), (7.4, 0, 0)]
    want_elem = ["N", "C", "C", "O", "C", "N", "C"]
    assert torch.allclose(g.prot_x, torch.tensor(want_xyz, dtype=torch.float32))
    assert g.prot_h.argmax(1).tolist() == [PROT_ELEMENTS.index(e) for e in want_elem]
    assert torch.allclose(g.pharm_x0, torch.tensor([[2.5 / 3, 2.5 / 3, 0.0]]))          # mean of the ligand's heavy atoms C, O, N
    undirected = {(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (5, 6)}
    assert set(zip(g.pp_src.tolist(), g.pp_dst.tolist())) == undirected | {(b, a) for a, b in undirected}
    out = (tmp_path / "pocket.pdb").read_text().splitlines()
    assert sum(l.startswith("ATOM") for l in out) == 9 and not any("HOH" in l or "GLY" in l for l in out)
    # residue-list variant: centre = mean of ALL atoms of the listed residues (hydrogens included), :170-173
    g2 = P.process_ligand_and_pocket(pdb, None, emap, CUTOFFS, 8.0, residue_list=["A:1"],
                                     pp_edges=oracle_pp(all_pos[:5]))
    ala = np.array([a.coord for a in res[0].atoms])
    assert g2.num_nodes('prot') == 5 and torch.allclose(g2.pharm_x0, torch.tensor(ala.mean(0, keepdims=True)))
    with pytest.raises(ValueError):
        P.process_ligand_and_pocket(pdb, None, emap, CUTOFFS, 8.0)
    two = tmp_path / "two.sdf"
    two.write_text(SDF + SDF)
    with pytest.raises(NotImplementedError):
        P.parse_ligand(two)


def make_split(d, seed, n_graphs):
    rng = np.random.default_rng(seed)
    d.mkdir(parents=True)
    np_, nf_, nh_ = rng.integers(8, 14, n_graphs), rng.integers(3, 12, n_graphs), rng.integers(0, 5, n_graphs)

    def idx(c):
        e = np.cumsum(c)
        return np.stack([e - c, e], 1)
    np.savez(d / 'prot_pharm_tensors.npz',
             prot_pos=rng.normal(size=(np_.sum(), 3)).astype(np.float32) * 4, prot_feat=rng.integers(0, 11, np_.sum()),
             prot_idx=idx(np_), pharm_pos=rng.normal(size=(nf_.sum(), 3)).astype(np.float32), pharm_feat=rng.integers(0, 6, nf_.sum()),
             pharm_idx=idx(nf_), prot_ph_pos=rng.normal(size=(nh_.sum(), 3)).astype(np.float32),
             prot_ph_feat=rng.integers(0, 6, nh_.sum()), prot_ph_idx=idx(nh_))
    with gzip.open(d / 'prot_file_names.pkl.gz', 'wb') as f:
        pickle.dump([f"{d.name}_{i}.pdb" for i in range(n_graphs)], f)
    return np_, nf_, nh_


def test_processed_dataset_and_collate(tmp_path):
    root = tmp_path / "processed"
    a = make_split(root / "split_0", 0, 3)
    b = make_split(root / "split_1", 1, 2)
    make_split(root / "split_2", 2, 4)
    cfg = dict(raw_data_dir=str(tmp_path), processed_data_dir=str(root), graph_cutoffs=CUTOFFS, prot_elements=PROT_ELEMENTS,
               ph_type_map=pfa.analysis.ph_idx_to_type, pp_edges_fn=oracle_pp)
    ds = D.ProteinPharmacophoreDataset('train', [0, 1], **cfg)
    assert len(ds) == 5 and len(ds.prot_file_names) == 5
    np_all, nf_all = np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]])
    raw1 = np.load(root / "split_1" / "prot_pharm_tensors.npz")
    for i in range(5):
        g = ds[i]
        assert g.num_nodes('prot') == np_all[i] and g.num_nodes('pharm') == nf_all[i]
        assert g.prot_h.shape[1] == 11 and g.pharm_h0.shape[1] == 6 and g.prot_h.sum(1).eq(1).all()
        src, dst = oracle_pp(g.prot_x)
        assert torch.equal(g.pp_src, src) and torch.equal(g.pp_dst, dst)
    # graph 3 = first graph of the second file: its rows start where the first file ended (global index fix-up)
    g3 = ds[3]
    assert np.allclose(g3.prot_x.numpy(), raw1['prot_pos'][:b[0][0]])
    assert np.allclose(g3.pharm_x0.numpy(), raw1['pharm_pos'][:b[1][0]])
    gb = D.collate_fn([ds[0], ds[3], ds[4]])
    assert gb.batch_size == 3 and gb.num_nodes('prot') == np_all[[0, 3, 4]].sum()
    assert gb.prot_ptr.tolist() == [0] + np.cumsum(np_all[[0, 3, 4]]).tolist()
    # subsampling keeps between subsample_min and min(subsample_max, n) centers
    import random
    random.seed(0)
    ds_s = D.ProteinPharmacophoreDataset('train', [0, 1], subsample_pharms=True, subsample_min=3, subsample_max=5, **cfg)
    for i in range(5):
        n = ds_s[i].num_nodes('pharm')
        assert (3 <= n <= min(5, nf_all[i])) if nf_all[i] >= 3 else n == nf_all[i]
    dm = D.CrossdockedDataModule(dataset_config=cfg, batch_size=2, num_workers=0, validation_splits=[2])
    dm.setup('fit')
    assert len(dm.train_dataset) == 5 and len(dm.val_dataset) == 4
    sizes = [g.batch_size for g in dm.val_dataloader()]
    assert sizes == [2, 2]
    with pytest.raises(NotImplementedError):
        D.CrossdockedDataModule(cfg, 2, 0, [])


def test_dataloader_workers_slice_the_precomputed_pp_table(tmp_path):
    """DataLoader workers are forked after the parent has initialised HIP, so they must never build a radius graph on
    the GPU: the data module builds the pp-edge table of every pocket in setup() (main process) and __getitem__ only
    slices it; a worker that finds no table raises instead of touching the GPU."""
    root = tmp_path / "processed"
    for i in range(3):
        make_split(root / f"split_{i}", i, 3)
    cfg = dict(raw_data_dir=str(tmp_path), processed_data_dir=str(root), graph_cutoffs=CUTOFFS, prot_elements=PROT_ELEMENTS,
               ph_type_map=pfa.analysis.ph_idx_to_type, pp_edges_fn=oracle_pp)
    dm = D.CrossdockedDataModule(dataset_config=cfg, batch_size=2, num_workers=2, validation_splits=[2])
    dm.setup('fit')
    ds = dm.train_dataset
    assert ds._pp_ptr is not None and ds._pp_src.dtype == torch.int16 and int(ds._pp_ptr[-1]) == ds._pp_src.numel()
    assert ds.pp_edge_counts().tolist() == [int(oracle_pp(ds[i].prot_x)[0].numel()) for i in range(len(ds))]
    got = [g for g in dm.train_dataloader()]                       # two forked workers
    assert sum(g.batch_size for g in got) == len(ds)
    ref = D.collate_fn([ds[0], ds[1]])
    assert torch.equal(got[0].pp_src, ref.pp_src) and torch.equal(got[0].prot_x, ref.prot_x)
    cold = D.ProteinPharmacophoreDataset('val', [2], **cfg)         # no table, no setup()
    with pytest.raises(RuntimeError, match="pp edge table"):
        next(iter(D.get_dataloader(cold, 2, num_workers=1)))


def make_mmcif(path):
    """The receptor of make_pdb as an mmCIF ``_atom_site`` loop (wwPDB column set), preceded by another loop, with a
    quoted atom name, '.' / '?' null markers, label and author numbering that differ, and a second model."""
    rows = []
    s = 1

    def add(group, name, alt, comp, lasym, lseq, x, y, z, occ, elem, aseq, aasym, model=1):
        nonlocal s
        qn = f'"{name}"' if "'" in name else name
        rows.append(f"{group} {s} {elem} {qn} {alt} {comp} {lasym} 1 {lseq} ? {x:.3f} {y:.3f} {z:.3f} {occ:.2f} 20.00 ? {aseq} {comp} {aasym} {qn} {model}")
        s += 1
    for name, xyz, elem, alt, occ in [("N", (0, 0, 0), "N", ".", 1.0), ("CA", (1.4, 0, 0), "C", ".", 1.0), ("C", (2.0, 1.3, 0), "C", ".", 1.0),
                                      ("O", (1.4, 2.3, 0), "O", ".", 1.0), ("CB", (2.0, -1.2, 0.5), "C", "A", 0.3),
                                      ("CB", (2.1, -1.1, 0.6), "C", "B", 0.7), ("HA", (1.5, 0.1, 1.0), "H", ".", 1.0)]:
        add("ATOM", name, alt, "ALA", "X", 11, *xyz, occ, elem, 1, "A")       # label chain X / seq 11 vs author A / 1
    for name, xyz, elem in [("N", (6, 0, 0), "N"), ("CA", (7.4, 0, 0), "C"), ("SE", (7.5, 1.5, 0), "SE")]:
        add("ATOM", name, ".", "MET", "X", 12, *xyz, 1.0, elem, 2, "A")
    for name, xyz, elem in [("N", (30, 0, 0), "N"), ("CA", (31.4, 0, 0), "C")]:
        add("ATOM", name, ".", "GLY", "X", 13, *xyz, 1.0, elem, 3, "A")
    add("HETATM", "O", ".", "HOH", "Y", ".", 0.5, 0.5, 0.5, 1.0, "O", 101, "A")
    add("HETATM", "CA", ".", "MSE", "X", 14, 0.2, 0.2, 0.2, 1.0, "C", 4, "A")
    add("ATOM", "N", ".", "ALA", "Z", 1, 0.1, 0.1, 0.1, 1.0, "N", 9, "B", model=2)
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_entity_id",
            "label_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "pdbx_formal_charge",
            "auth_seq_id", "auth_comp_id", "auth_asym_id", "auth_atom_id", "pdbx_PDB_model_num"]
    text = ["data_TEST", "#", "loop_", "_entity.id", "_entity.type", "1 polymer", "2 water", "#", "loop_"]
    text += ["_atom_site." + c for c in cols] + rows + ["#", "loop_", "_atom_type.symbol", "C", "N", "#"]
    path.write_text("\n".join(text) + "\n")


def test_mmcif_receptor_reads_like_the_pdb(tmp_path):
    """generate_pharmacophores.py:131-132 accepts '.mmcif' receptors (Bio.PDB.MMCIFParser): the same structure written as
    PDB and as mmCIF gives the same residues, atoms, alternate-location choice, pocket graph and pocket.pdb atoms."""
    pdb, cif, sdf = tmp_path / "rec.pdb", tmp_path / "rec.mmcif", tmp_path / "lig.sdf"
    make_pdb(pdb); make_mmcif(cif); sdf.write_text(SDF)
    a, b = P.read_pdb(pdb), P.read_mmcif(cif)
    assert [(r.resname, r.chain, r.resseq, r.icode, r.hetero) for r in a] == [(r.resname, r.chain, r.resseq, r.icode, r.hetero) for r in b]
    for ra, rb in zip(a, b):
        assert [(x.name, x.element, x.altloc.strip(), x.occupancy) for x in ra.atoms] == \
               [(x.name, x.element, x.altloc.strip(), x.occupancy) for x in rb.atoms]
        assert np.allclose(np.array([x.coord for x in ra.atoms]), np.array([x.coord for x in rb.atoms]))
    assert P._cif_tokens("""ATOM 1 O "O5'" . 'A B' x""") == ["ATOM", "1", "O", "O5'", ".", "A B", "x"]
    emap, _ = P.get_prot_atom_ph_type_maps({'prot_elements': PROT_ELEMENTS, 'ph_type_map': pfa.analysis.ph_idx_to_type})
    all_pos = torch.tensor(np.array([x.coord for r in a[:2] for x in r.atoms if x.element not in ("H", "SE")]))
    out_a, out_b = tmp_path / "a", tmp_path / "b"
    out_a.mkdir(); out_b.mkdir()
    ga = P.process_ligand_and_pocket(pdb, out_a, emap, CUTOFFS, 8.0, lig_file=sdf, pp_edges=oracle_pp(all_pos))
    gb = P.process_ligand_and_pocket(cif, out_b, emap, CUTOFFS, 8.0, lig_file=sdf, pp_edges=oracle_pp(all_pos))
    assert torch.equal(ga.prot_x, gb.prot_x) and torch.equal(ga.prot_h, gb.prot_h) and torch.equal(ga.pharm_x0, gb.pharm_x0)
    la = [l for l in (out_a / "pocket.pdb").read_text().splitlines() if l.startswith("ATOM")]
    lb = [l for l in (out_b / "pocket.pdb").read_text().splitlines() if l.startswith("ATOM")]
    assert len(la) == len(lb) == 9
    for x, y in zip(la, lb):          # same name / residue / chain / number / coordinates / element columns
        assert x[12:16].strip() == y[12:16].strip() and x[16:27] == y[16:27] and x[30:54] == y[30:54], (x, y)
        assert x[76:78].strip().upper() == y[76:78].strip().upper(), (x, y)
    assert P.read_pdb(out_b / "pocket.pdb")[0].atoms[0].name == "N"           # the written records parse back
    gr = P.process_ligand_and_pocket(cif, None, emap, CUTOFFS, 8.0, residue_list=["A:1"], pp_edges=oracle_pp(all_pos[:5]))
    assert gr.num_nodes('prot') == 5
    with pytest.raises(ValueError, match="unsupported receptor file type"):
        P.process_ligand_and_pocket(tmp_path / "rec.xyz", None, emap, CUTOFFS, 8.0, lig_file=sdf)


# ---- the reference's own dataset class as the witness (tests/golden/dataset.npz, recorded by make_golden.py from
# ---- dataset/protein_pharm_dataset.py:19-179,268-276 run on a tiny processed directory) ---------------------------------
def _rebuild_processed_dir(z, root):
    names = sorted({k.split("_")[1] + "_" + k.split("_")[2] for k in z if k.startswith("in_")})
    for sname in names:
        d = root / sname
        d.mkdir(parents=True)
        arrs = {k[len(f"in_{sname}_"):]: np.asarray(z[k]) for k in z if k.startswith(f"in_{sname}_")}
        np.savez(d / 'prot_pharm_tensors.npz', **arrs)
        with gzip.open(d / 'prot_file_names.pkl.gz', 'wb') as f:
            pickle.dump([f"{sname}_{i}.pdb" for i in range(len(arrs['prot_idx']))], f)
    return names


def test_dataset_reproduces_the_reference_dataset_class(tmp_path):
    """`ProteinPharmacophoreDataset.__getitem__` (global index fix-up over the split files, one-hot features, the random
    pharmacophore subsampling and its draws, the pp edges of build_initial_complex_graph) and `collate_fn` against what the
    REFERENCE's class returned for the same files and the same `random` seeds."""
    import random
    from helpers import load
    z = load("dataset.npz")
    root = tmp_path / "processed"
    _rebuild_processed_dir(z, root)
    kw = dict(name='train', split_idxs=[0, 2], raw_data_dir=str(tmp_path), processed_data_dir=str(root), graph_cutoffs=CUTOFFS,
              prot_elements=PROT_ELEMENTS, ph_type_map=pfa.analysis.ph_idx_to_type, pp_edges_fn=oracle_pp)
    ref_names = str(z["file_names"]).split("\n")
    n = int(z["n_graphs"])
    assert len(ref_names) == n == 7
    for tag, extra in (("plain", {}), ("sub", dict(subsample_pharms=True, subsample_min=int(z["subsample_min"]), subsample_max=int(z["subsample_max"])))):
        ds = D.ProteinPharmacophoreDataset(**kw, **extra)
        assert len(ds) == n and sorted(ds.prot_file_names) == sorted(ref_names)
        for i, fname in enumerate(ref_names):                 # the reference walks the split directories in file-system order
            j = ds.prot_file_names.index(fname)
            random.seed(1000 + i)
            g = ds[j]
            for short, x, h in (("prot", g.prot_x, g.prot_h), ("pharm", g.pharm_x0, g.pharm_h0), ("ph", g.prot_ph_x, g.prot_ph_h)):
                assert torch.equal(x, z[f"{tag}_{i}_{short}_x"]), (tag, i, short)
                assert torch.equal(h, z[f"{tag}_{i}_{short}_h"]), (tag, i, short)
            assert torch.equal(g.pp_src, z[f"{tag}_{i}_pp_src"].long()) and torch.equal(g.pp_dst, z[f"{tag}_{i}_pp_dst"].long())
        if tag == "sub":                                      # the subsampling really drew: some pockets lost centers
            assert any(z[f"sub_{i}_pharm_x"].shape[0] < z[f"plain_{i}_pharm_x"].shape[0] for i in range(n))
    ds = D.ProteinPharmacophoreDataset(**kw)
    pick = [ds.prot_file_names.index(ref_names[i]) for i in z["collate_pick"].tolist()]
    gb = D.collate_fn([ds[j] for j in pick])
    assert torch.equal(torch.diff(gb.prot_ptr), z["collate_prot_counts"].long())
    assert torch.equal(torch.diff(gb.pharm_ptr), z["collate_pharm_counts"].long())
    assert torch.equal(torch.diff(gb.prot_ph_ptr), z["collate_ph_counts"].long())
    assert torch.equal(gb.prot_x, z["collate_prot_x"]) and torch.equal(gb.pharm_x0, z["collate_pharm_x"]) and torch.equal(gb.prot_ph_x, z["collate_ph_x"])
    assert torch.equal(gb.pp_src, z["collate_pp_src"].long()) and torch.equal(gb.pp_dst, z["collate_pp_dst"].long())


def test_from_dgl_adapter_on_a_heterograph_shaped_like_the_references():
    """graph.from_dgl on DGL-style heterographs (tests/golden/ref_shim.HeteroGraph, the stand-in the goldens were recorded
    with): graphs assembled the way build_initial_complex_graph assembles them (dataset/protein_pharm_dataset.py:226-264:
    node types prot / pharm / prot_ph, edge types pp / pf / ff / fp, x_0 / h_0 node data) from the reference's recorded
    outputs, single and batched with the stand-in's dgl.batch; when the reference is on this machine its own function builds
    the graph."""
    import sys
    from helpers import GOLDEN, load
    sys.path.insert(0, GOLDEN)
    import ref_shim
    z = load("dataset.npz")

    def hetero(i):
        no = ([], [])
        data = {('prot', 'pp', 'prot'): (z[f"plain_{i}_pp_src"].long(), z[f"plain_{i}_pp_dst"].long()), ('prot', 'pf', 'pharm'): no,
                ('pharm', 'ff', 'pharm'): no, ('pharm', 'fp', 'prot'): no}
        nn = {'prot': z[f"plain_{i}_prot_x"].shape[0], 'pharm': z[f"plain_{i}_pharm_x"].shape[0], 'prot_ph': z[f"plain_{i}_ph_x"].shape[0]}
        g = ref_shim.heterograph(data, num_nodes_dict=nn)
        for nt, short in (("prot", "prot"), ("pharm", "pharm"), ("prot_ph", "ph")):
            g.nodes[nt].data['x_0'] = z[f"plain_{i}_{short}_x"]
            g.nodes[nt].data['h_0'] = z[f"plain_{i}_{short}_h"]
        return g
    pg = pfa.graph.from_dgl(hetero(2))
    assert pg.batch_size == 1 and torch.equal(pg.prot_x, z["plain_2_prot_x"]) and torch.equal(pg.pharm_h0, z["plain_2_pharm_h"])
    assert torch.equal(pg.prot_ph_x, z["plain_2_ph_x"]) and torch.equal(pg.pp_src, z["plain_2_pp_src"].long())
    assert pg.prot_ptr.tolist() == [0, z["plain_2_prot_x"].shape[0]] and pg.pharm_ptr.tolist() == [0, z["plain_2_pharm_x"].shape[0]]
    pick = z["collate_pick"].tolist()
    gb = pfa.graph.from_dgl(ref_shim.batch([hetero(i) for i in pick]))
    assert gb.batch_size == len(pick)
    assert torch.equal(torch.diff(gb.prot_ptr), z["collate_prot_counts"].long()) and torch.equal(gb.prot_x, z["collate_prot_x"])
    assert torch.equal(gb.pp_src, z["collate_pp_src"].long()) and torch.equal(gb.pp_dst, z["collate_pp_dst"].long())
    # the batched adapter output is what this repository's own collate gives for the same pockets
    own = pfa.batch([pfa.graph.from_dgl(hetero(i)) for i in pick])
    for f in ("prot_x", "prot_h", "prot_ptr", "pharm_ptr", "pp_src", "pp_dst", "pharm_x0", "pharm_h0", "prot_ph_x", "prot_ph_ptr"):
        assert torch.equal(getattr(gb, f), getattr(own, f)), f
    if os.path.isdir("/root/reference/pharmacoforge"):        # build container only: the reference's own graph builder
        ref_shim.install("/root/reference")
        from pharmacoforge.dataset.protein_pharm_dataset import build_initial_complex_graph as ref_build
        g = ref_build(z["plain_1_prot_x"], z["plain_1_prot_h"], {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9},
                      pharm_atom_positions=z["plain_1_pharm_x"], pharm_atom_features=z["plain_1_pharm_h"],
                      prot_ph_pos=z["plain_1_ph_x"], prot_ph_feat=z["plain_1_ph_h"])
        pg = pfa.graph.from_dgl(g)
        assert torch.equal(pg.pp_src, z["plain_1_pp_src"].long()) and torch.equal(pg.pp_dst, z["plain_1_pp_dst"].long())
        assert torch.equal(pg.prot_x, z["plain_1_prot_x"]) and pg.num_nodes('pharm') == z["plain_1_pharm_x"].shape[0]
