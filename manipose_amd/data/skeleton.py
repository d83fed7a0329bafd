"""Kinematic-tree metadata consumed by the decoder and the flip augmentation.

Own restatement of the reference's ``Skeleton`` interface (hpe/mh_so3_hpe/data/skeleton.py:7-172: ``parents``,
``has_children``, ``children``, ``bones``, ``bones_left/right``, ``joints_left/right``, ``num_joints``,
``num_bones``, ``t_pose_operators``) plus the 17-joint H36M tree (hpe/mh_so3_hpe/data/dataset_3dhp.py:132-138)
and its T-pose operators (hpe/mh_so3_hpe/data/h36m_lifting.py:40-57).  The HIP decoder has this tree compiled
in (manipose_amd/csrc/fk_decode.hip); ``assert_h36m`` guards against a different skeleton being passed.
"""
from __future__ import annotations

import numpy as np
import torch

H36M_PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15)
H36M_JOINTS_LEFT = (4, 5, 6, 11, 12, 13)
H36M_JOINTS_RIGHT = (1, 2, 3, 14, 15, 16)
_OPS = {1: (1, 0, 0), 2: (0, -1, 0), 3: (0, -1, 0), 4: (-1, 0, 0), 5: (0, -1, 0), 6: (0, -1, 0), 7: (0, 1, 0),
        8: (0, 1, 0), 9: (0, 1, 0), 10: (0, 1, 0), 11: (-1, 0, 0), 12: (-1, 0, 0), 13: (-1, 0, 0), 14: (1, 0, 0),
        15: (1, 0, 0), 16: (1, 0, 0)}
T_POSE_OPERATORS = {j: torch.tensor(v, dtype=torch.float) for j, v in _OPS.items()}


class Skeleton:
    def __init__(self, parents, joints_left, joints_right, t_pose_operators, joints_group=None, joints_names=None):
        if len(joints_left) != len(joints_right):
            raise AssertionError("joints_left and joints_right must have the same length")
        self.t_pose_operators = t_pose_operators
        self._parents = np.array(parents)
        self._joints_left = list(joints_left)
        self._joints_right = list(joints_right)
        self._joints_group = joints_group
        self._joints_names = list(joints_names) if joints_names is not None else [""] * len(self._parents)
        if len(self._joints_names) != len(self._parents):
            raise AssertionError("joint_names should be an iterable with as many elements as joints.")
        self._refresh()

    def _refresh(self):
        n = len(self._parents)
        self._has_children = np.zeros(n, dtype=bool)
        self._children = [[] for _ in range(n)]
        for j, p in enumerate(self._parents):
            if p != -1:
                self._has_children[p] = True
                self._children[p].append(j)
        self._bones = tuple((j, int(p)) for j, p in enumerate(self._parents) if p >= 0)
        self._bones_names = tuple(f"{self._joints_names[j]}->{self._joints_names[i]}" for i, j in self._bones)
        index_of = {b: i for i, b in enumerate(self._bones)}
        parent_of = dict(self._bones)
        self._bones_left = tuple(index_of[(j, parent_of[j])] for j in self._joints_left if j >= 0)
        self._bones_right = tuple(index_of[(j, parent_of[j])] for j in self._joints_right if j >= 0)

    num_joints = property(lambda s: len(s._parents))
    num_bones = property(lambda s: int((s._parents >= 0).sum()))
    parents = property(lambda s: s._parents)
    has_children = property(lambda s: s._has_children)
    children = property(lambda s: s._children)
    joints_left = property(lambda s: s._joints_left)
    joints_right = property(lambda s: s._joints_right)
    joints_group = property(lambda s: s._joints_group)
    joints_names = property(lambda s: s._joints_names)
    bones = property(lambda s: s._bones)
    bones_left = property(lambda s: s._bones_left)
    bones_right = property(lambda s: s._bones_right)
    bones_names = property(lambda s: s._bones_names)


def h36m_skeleton() -> Skeleton:
    return Skeleton(list(H36M_PARENTS), list(H36M_JOINTS_LEFT), list(H36M_JOINTS_RIGHT), T_POSE_OPERATORS)


def assert_h36m(skeleton) -> None:
    """The fused HIP decoder is compiled for the 17-joint H36M tree; refuse anything else loudly."""
    ok = (tuple(int(p) for p in skeleton.parents) == H36M_PARENTS and all(
        tuple(float(x) for x in skeleton.t_pose_operators[j]) == tuple(float(x) for x in _OPS[j]) for j in _OPS))
    if not ok:
        raise AssertionError("manipose_amd: the HIP forward-kinematics decoder supports the 17-joint H36M skeleton "
                             "(parents / T-pose operators of hpe/mh_so3_hpe/data/h36m_lifting.py) only")
