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
