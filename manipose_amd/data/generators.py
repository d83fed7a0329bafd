"""GPU-resident counterpart of the reference's ``PoseSequenceGenerator`` (hpe/mh_so3_hpe/data/generators.py:44-219) with its
``PoseFlip`` transform (hpe/mh_so3_hpe/augmentations/transforms.py:7-28): the pose sequences are uploaded once, back to back, and
every batch of windows is cut out of them by one HIP kernel (``mp_gather_windows``) - no CPU workers, no host copies per step.

Same constructor arguments, the same index tables (``_map_index_to_pose`` / ``_map_index_to_frame``), the same consumption of the
torch CPU RNG per item (random start: one ``torch.randint``; flip: one ``torch.rand``), so a seeded run draws the windows the
reference's loader draws with ``num_workers=0``.  The occlusion patterns (``miss_type`` "random", "random_left_arm_right_leg",
"structured_joint", "structured_frame", "noisy", "all"; generators.py:160-216) are drawn on the host from numpy's global RNG with
the reference's calls in the reference's order - a (T, J) table per window - and applied by the same kernel.

Not replicated: the reference flips IN PLACE a view of its own float32 dataset array (``torch.from_numpy(...).float()`` does not copy
float32 data), i.e. it mirrors its stored sequences a little more every epoch; here the stored sequences are never modified.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib


class PoseSequenceGenerator:
    possible_miss_types_rates = {"no_miss": 0.2, "random": 0.2, "random_left_arm_right_leg": 0.4, "structured_joint": 0.4,
                                 "structured_frame": 0.2}            # generators.py:48-55

    def __init__(self, poses_3d: Sequence, poses_2d: Sequence, cameras=None, seq_len: int = 8, random_start: bool = False,
                 drop_last: bool = True, miss_type: str = "no_miss", miss_rate: float = 0.2, noise_sigma: float = 5,
                 transform=None, device: Optional[torch.device] = None):
        assert poses_3d is not None
        assert len(poses_3d) == len(poses_2d)
        if miss_type != "all" and miss_type not in self.possible_miss_types_rates and miss_type != "noisy":
            raise ValueError(f"Unexpected miss_type: {miss_type}")
        if transform is not None and not hasattr(transform, "p") and not hasattr(transform, "probability"):
            raise NotImplementedError("manipose_amd: the only transform fused into the window kernel is PoseFlip")
        self._seq_len, self._random_start, self.drop_last = int(seq_len), bool(random_start), bool(drop_last)
        self.miss_type, self.miss_rate, self.noise_sigma, self.transform = miss_type, miss_rate, noise_sigma, transform
        self._cameras = cameras
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("manipose_amd: the window generator keeps the sequences in HBM and runs a HIP kernel; no CPU fallback")
        lens = [int(p.shape[0]) for p in poses_3d]
        for a, b in zip(poses_3d, poses_2d):
            assert a.shape[0] == b.shape[0] and a.shape[1] == b.shape[1] and a.shape[2] == 3 and b.shape[2] == 2
        self._lens = lens
        self._J = int(poses_3d[0].shape[1])
        # index tables exactly as generators.py:87-104
        self._map_index_to_pose, self._map_index_to_frame = [], []
        for i, n in enumerate(lens):
            size = n // self._seq_len
            if not drop_last and n % self._seq_len > 0:
                size += 1
            self._map_index_to_pose += [i] * size
            self._map_index_to_frame += [k * self._seq_len for k in range(size)]
        self._ds_len = len(self._map_index_to_pose)
        off = np.zeros(len(lens) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)

        def resident(seqs):      # device tensors (the ingest kernels' outputs) are concatenated in HBM; host arrays are uploaded once
            if all(isinstance(s, torch.Tensor) for s in seqs):
                return torch.cat([s.to(self.device, torch.float32) for s in seqs], dim=0).contiguous()
            host = [s.cpu().numpy() if isinstance(s, torch.Tensor) else np.asarray(s) for s in seqs]
            return torch.from_numpy(np.concatenate([h.astype(np.float32, copy=False) for h in host], axis=0)).to(self.device).contiguous()
        self._p3 = resident(poses_3d)
        self._p2 = resident(poses_2d)
        self._off = torch.from_numpy(off).to(self.device)
        sk = getattr(transform, "skeleton", None)
        mirror = list(range(self._J))
        if sk is not None:
            for l, r in zip(sk.joints_left, sk.joints_right):
                mirror[l], mirror[r] = r, l
        self._mirror = (C.c_int32 * self._J)(*mirror)
        self._p_flip = float(getattr(transform, "p", getattr(transform, "probability", 0.0))) if transform is not None else 0.0

    def __len__(self) -> int:
        return self._ds_len

    def _draw_mask(self):
        """generators.py:160-216 for one window: (mask (T, J) or None, noise (T, J, 2) or None), numpy global RNG, reference call order."""
        import math
        shape = (self._seq_len, self._J)
        if self.miss_type == "all":
            miss_type = np.random.choice(list(self.possible_miss_types_rates.keys()))
            miss_rate = self.possible_miss_types_rates[miss_type]
        else:
            miss_type, miss_rate = self.miss_type, self.miss_rate
        if miss_type == "no_miss":
            return None, None
        if miss_type == "random":
            mask = np.zeros(shape)
            u = np.random.uniform(0.0, 1.0, size=shape)
            mask[u > miss_rate] = 1.0
            return mask, None
        if miss_type == "random_left_arm_right_leg":
            mask = np.ones(shape)
            rand = np.random.choice(self._seq_len, size=math.floor(miss_rate * self._seq_len), replace=False).tolist()
            for i in [1, 2, 3, 11, 12, 13]:
                mask[rand, i] = 0.0
            return mask, None
        if miss_type == "structured_joint":
            mask = np.ones(shape)
            occl_len = int(self._seq_len * miss_rate)
            rand = np.random.choice(self._seq_len - occl_len, size=1, replace=False)
            mask[rand[0]: rand[0] + occl_len, [1, 2, 3]] = 0.0
            return mask, None
        if miss_type == "structured_frame":
            mask = np.ones(shape)
            occl_len = int(self._seq_len * miss_rate)
            rand = np.random.choice(self._seq_len - occl_len, size=1, replace=False)
            mask[rand[0]: rand[0] + occl_len] = 0.0
            return mask, None
        if miss_type == "noisy":
            return None, np.random.normal(0, self.noise_sigma, size=(self._seq_len, self._J, 2))
        raise ValueError(f"Unexpected miss_type: {self.miss_type}")

    def draw_occlusions(self, n: int):
        """Occlusion tables of ``n`` consecutive windows: (mask (n, T, J) float32 or None, noise (n, T, J, 2) float32 or None)."""
        if self.miss_type == "no_miss":
            return None, None
        masks, noises, any_mask, any_noise = [], [], False, False
        for _ in range(n):
            m, z = self._draw_mask()
            any_mask |= m is not None
            any_noise |= z is not None
            masks.append(m); noises.append(z)
        ones = np.ones((self._seq_len, self._J))
        zeros = np.zeros((self._seq_len, self._J, 2))
        mask = torch.from_numpy(np.stack([ones if m is None else m for m in masks]).astype(np.float32)) if any_mask else None
        noise = torch.from_numpy(np.stack([zeros if z is None else z for z in noises]).astype(np.float32)) if any_noise else None
        return mask, noise

    def draw(self, indices: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Host part of ``__getitem__`` for a batch: (sequence, start, flip) per index, consuming the torch CPU RNG item by item in the
        reference's order (generators.py:121-131 then transforms.py:22)."""
        seq = np.empty(len(indices), dtype=np.int32)
        start = np.empty(len(indices), dtype=np.int32)
        flip = np.zeros(len(indices), dtype=np.uint8)
        for n, index in enumerate(indices):
            s = self._map_index_to_pose[index]
            seq[n] = s
            if self._random_start:
                start[n] = torch.randint(low=0, high=self._lens[s] - self._seq_len, size=(1,)).item()
            else:
                start[n] = self._map_index_to_frame[index]
            if self.drop_last and start[n] + self._seq_len > self._lens[s]:
                raise IndexError("window past the end of its sequence (drop_last=True)")
            if self.transform is not None and torch.rand(1).item() <= self._p_flip:
                flip[n] = 1
        return torch.from_numpy(seq), torch.from_numpy(start), torch.from_numpy(flip)

    def gather(self, seq: torch.Tensor, start: torch.Tensor, flip: Optional[torch.Tensor], mask: Optional[torch.Tensor] = None,
               noise: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(X (B,T,J,2), y (B,T,J,3)) on the device for explicit (sequence, start, flip) triples (+ optional occlusion tables):
        one kernel launch."""
        B, T, J = int(seq.numel()), self._seq_len, self._J
        mask_d = mask.to(self.device, torch.float32).contiguous() if mask is not None else None
        noise_d = noise.to(self.device, torch.float32).contiguous() if noise is not None else None
        seq_d = seq.to(self.device, torch.int32).contiguous()
        start_d = start.to(self.device, torch.int32).contiguous()
        flip_d = flip.to(self.device, torch.uint8).contiguous() if flip is not None else None
        X = torch.empty(B, T, J, 2, device=self.device)
        y = torch.empty(B, T, J, 3, device=self.device)
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.mp_gather_windows(_lib.ptr(self._p2), _lib.ptr(self._p3), C.c_void_p(self._off.data_ptr()), len(self._lens),
                                             C.c_void_p(seq_d.data_ptr()), C.c_void_p(start_d.data_ptr()),
                                             C.c_void_p(flip_d.data_ptr()) if flip_d is not None else None, self._mirror, _lib.ptr(mask_d),
                                             _lib.ptr(noise_d), B, T, J,
                                             _lib.ptr(X), _lib.ptr(y), _lib.stream_ptr()), "mp_gather_windows")
        return X, y

    def batch(self, indices: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        """The windows the reference's ``__getitem__`` returns for ``indices`` (stacked), as device tensors.  (torch draws of all items
        first, then the numpy occlusion draws: the two RNG streams are independent, so each is consumed in the reference's order.)"""
        seq, start, flip = self.draw(indices)
        mask, noise = self.draw_occlusions(len(indices))
        return self.gather(seq, start, flip, mask, noise)

    def __getitem__(self, index: int) -> Tuple[torch.Tensor, torch.Tensor]:
        X, y = self.batch([index])
        return X[0], y[0]


__all__ = ["PoseSequenceGenerator"]
