"""Shared helpers for the parity tests (fixture loading; oracle is imported here as the checker)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG_KEYS = ("T", "J", "num_bones", "C_rot", "depth_rot", "heads_rot", "C_seg", "depth_seg", "heads_seg", "n_hyp")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {k: z[k] for k in z.files}
    if "cfg" in out:
        out["cfg"] = {k: int(v) for k, v in zip(CFG_KEYS, out["cfg"])}
        out["cfg"]["rot_dim"] = int(out["rot_dim"]) if "rot_dim" in out else 6
    return out


def fixture_state(fx, prefix="w::"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith(prefix)}


def droppath_mask_names(cfg, drop_path_rate):
    """Order in which the reference draws DropPath masks in one forward (mix_ste.py:70,128-173,352-358):
    rotations module then segments module; within a MixSTE: STE[i].attn, STE[i].mlp, TTE[i].attn, TTE[i].mlp
    for i = 0..depth-1, skipping blocks whose rate linspace(0, rate, depth)[i] == 0."""
    names = []
    for pre, depth in (("rotations_module.", cfg["depth_rot"]), ("segments_module.", cfg["depth_seg"])):
        dpr = torch.linspace(0, drop_path_rate, depth).tolist()
        for i in range(depth):
            if dpr[i] > 0:
                for kind in ("STEblocks", "TTEblocks"):
                    names += [f"{pre}{kind}.{i}.attn", f"{pre}{kind}.{i}.mlp"]
    return names


def fixture_masks(fx):
    n = int(fx["n_masks"])
    names = droppath_mask_names(fx["cfg"], float(fx["drop_path_rate"]))
    assert len(names) == n, (names, n)
    return {nm: torch.from_numpy(fx[f"mask::{i:03d}"].copy()) for i, nm in enumerate(names)}


def raw_dataset_dicts(fx):
    """Rebuild the raw dictionaries of the four files from the flat fixture entries."""
    h3, h2, tr, te = {}, {}, {}, {}
    for k in fx.files:
        if k.startswith("raw.h36m3|"):
            _, s, a = k.split("|")
            h3.setdefault(s, {})[a] = fx[k]
        elif k.startswith("raw.h36m2|"):
            _, s, a, c = k.split("|")
            h2.setdefault(s, {}).setdefault(a, {})[int(c)] = fx[k]
        elif k.startswith("raw.hptrain|"):
            _, seq, cam, what = k.split("|")
            tr.setdefault(seq, [{}])[0].setdefault(cam, {})["data_3d" if what == "3d" else "data_2d"] = fx[k]
        elif k.startswith("raw.hptest|"):
            _, seq, what = k.split("|")
            te.setdefault(seq, {})[{"3d": "data_3d", "2d": "data_2d", "valid": "valid"}[what]] = fx[k]
    h2 = {s: {a: [d[a][c] for c in sorted(d[a])] for a in d} for s, d in h2.items()}
    return h3, h2, tr, te


def h36m_calibration():
    import json
    with open(os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "manipose_amd", "data", "h36m_cameras.json")) as f:
        return json.load(f)


def write_raw_dataset_files(folder, fx):
    """The fixture's raw arrays as files in the reference's on-disk formats (npz whose single entry is a pickled dictionary:
    h36m_lifting.py:620, utils.py:13-14, dataset_3dhp.py:153,185)."""
    h3, h2, tr, te = raw_dataset_dicts(fx)
    np.savez_compressed(os.path.join(folder, "data_3d_h36m.npz"), positions_3d=h3)
    np.savez_compressed(os.path.join(folder, "data_2d_h36m_gt.npz"), positions_2d=h2, metadata={"num_joints": 17})
    np.savez_compressed(os.path.join(folder, "data_train_3dhp.npz"), data=tr)
    np.savez_compressed(os.path.join(folder, "data_test_3dhp.npz"), data=te)
