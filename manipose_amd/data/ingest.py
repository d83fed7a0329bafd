"""Dataset ingest on the device: the reference's loaders for its two on-disk formats, with the per-sequence arithmetic done by HIP
kernels (``mp_ingest_pose3d`` / ``mp_ingest_pose2d``) straight into HBM, where ``PoseSequenceGenerator`` keeps the sequences.

* Human3.6M - ``Human36mDataset`` (hpe/mh_so3_hpe/data/h36m_lifting.py:587-660: ``data_3d_h36m.npz`` = {"positions_3d": {subject:
  {action: (N, 32, 3) world coordinates in metres}}}), ``read_3d_data`` (data/utils.py:29-58: world -> each of the four cameras, root
  relative), ``create_2d_data`` (data/utils.py:9-26: ``data_2d_h36m_<detector>.npz`` = {"positions_2d": {subject: {action: [4 x
  (N, 17, 2) pixels]}}} -> normalised screen coordinates) and ``fetch`` (data/utils.py:61-127: flat lists over subject x action x
  camera, action filter, temporal stride).
* MPI-INF-3DHP - ``Dataset3DHP`` (data/dataset_3dhp.py:107-229: ``data_train_3dhp.npz`` = {"data": {"S<i> Seq<j>": [{camera:
  {"data_3d" (N, 17, 3) mm, "data_2d" (N, 17, 2) px}}]}}, ``data_test_3dhp.npz`` = {"data": {"TS<i>": {"data_3d", "data_2d",
  "valid" (N,)}}}).

The raw arrays cross PCIe once; each Human3.6M take is uploaded once and read by the four camera launches.  Sequences come back
as device tensors.  The camera calibration (``h36m_cameras.json``) is the public Human3.6M calibration - dataset facts, not code.
No CPU fallback: without a ROCm device these loaders raise.
"""
from __future__ import annotations

import copy
import ctypes as C
import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .skeleton import Skeleton, T_POSE_OPERATORS, h36m_skeleton

TRAIN_SUBJECTS = ["S1", "S5", "S6", "S7", "S8"]                  # h36m_lifting.py:33-34
TEST_SUBJECTS = ["S9", "S11"]
H36M_ACTIONS = ["directions", "discussion", "eating", "greeting", "phoning", "photo", "posing", "purchases", "sitting", "sittingdown",
                "smoking", "waiting", "walkdog", "walking", "walktogether"]                     # h36m_lifting.py:663-679
H36M_JOINT_NAMES_17 = ["Hip", "RHip", "RKnee", "RFoot", "LHip", "LKnee", "LFoot", "Spine", "Thorax", "Neck/Nose", "Head", "LShoulder",
                       "LElbow", "LWrist", "RShoulder", "RElbow", "RWrist"]                    # h36m_lifting.py:13-30 after :651-653
_DROPPED_OF_32 = (4, 5, 9, 10, 11, 16, 20, 21, 22, 23, 24, 28, 29, 30, 31)                     # h36m_lifting.py:651-653
H36M_KEPT_JOINTS_17 = tuple(j for j in range(32) if j not in _DROPPED_OF_32)
MAP_H36M_TO_MPI_JOINTS = (14, 8, 9, 10, 11, 12, 13, 15, 1, 16, 0, 5, 6, 7, 2, 3, 4)          # dataset_3dhp.py:55-73
MAP_MPI_TO_H36M_JOINTS = tuple(int(i) for i in np.argsort(MAP_H36M_TO_MPI_JOINTS))             # dataset_3dhp.py:35-53


def _device(device=None) -> torch.device:
    dev = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    if dev.type != "cuda":
        raise RuntimeError("manipose_amd: dataset ingest runs HIP kernels and keeps the sequences in HBM; no CPU fallback")
    return dev


def _upload(a, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def _i32(values):
    return (C.c_int32 * len(values))(*[int(v) for v in values])


def _f32(values):
    return (C.c_float * len(values))(*[float(v) for v in values]) if values is not None else None


def ingest_pose3d(raw: torch.Tensor, joint_map: Sequence[int], orientation=None, translation=None, root_raw: int = -1,
                  root_out: int = -1, divisor: float = 1.0, frames: Optional[torch.Tensor] = None) -> torch.Tensor:
    """raw (N, Jraw, 3) device float32 -> (n, len(joint_map), 3); see ``mp_ingest_pose3d`` in include/manipose_hip.h."""
    assert raw.is_cuda and raw.dtype == torch.float32 and raw.is_contiguous() and raw.dim() == 3 and raw.shape[2] == 3
    n = int(frames.numel()) if frames is not None else int(raw.shape[0])
    out = torch.empty(n, len(joint_map), 3, dtype=torch.float32, device=raw.device)
    with torch.cuda.device(raw.device):
        _lib.check(_lib.load().mp_ingest_pose3d(_lib.ptr(raw), int(raw.shape[1]), C.c_void_p(frames.data_ptr()) if frames is not None else None,
                                                n, _i32(joint_map), len(joint_map), _f32(orientation), _f32(translation), root_raw,
                                                root_out, float(divisor), _lib.ptr(out), _lib.stream_ptr()), "mp_ingest_pose3d")
    return out


def ingest_pose2d(raw: torch.Tensor, joint_map: Sequence[int], w: int, h: int, frames: Optional[torch.Tensor] = None) -> torch.Tensor:
    """raw (N, Jraw, C >= 2) device float32 pixel keypoints -> (n, len(joint_map), 2) normalised screen coordinates."""
    assert raw.is_cuda and raw.dtype == torch.float32 and raw.is_contiguous() and raw.dim() == 3 and raw.shape[2] >= 2
    n = int(frames.numel()) if frames is not None else int(raw.shape[0])
    out = torch.empty(n, len(joint_map), 2, dtype=torch.float32, device=raw.device)
    with torch.cuda.device(raw.device):
        _lib.check(_lib.load().mp_ingest_pose2d(_lib.ptr(raw), int(raw.shape[1]), int(raw.shape[2]),
                                                C.c_void_p(frames.data_ptr()) if frames is not None else None, n, _i32(joint_map),
                                                len(joint_map), float(w), float(h), _lib.ptr(out), _lib.stream_ptr()), "mp_ingest_pose2d")
    return out


def normalize_screen_coordinates(X, w, h):
    """data/camera.py:9-14 on host arrays (used for the handful of calibration constants; pose arrays go through the kernel)."""
    assert X.shape[-1] == 2
    return X / w * 2 - [1, h / w]


def h36m_cameras() -> Dict[str, List[dict]]:
    """Per subject, the four cameras as Human36mDataset.__init__ prepares them (h36m_lifting.py:591-618): centre and focal length in
    normalised screen units, translation in metres, ``intrinsic`` = (focal 2, centre 2, radial 3, tangential 2)."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "h36m_cameras.json")) as f:
        calib = json.load(f)
    out = {}
    for subject, cams in calib["extrinsic"].items():
        out[subject] = []
        for i, ext in enumerate(cams):
            cam = dict(calib["intrinsic"][i])
            cam.update(ext)
            for k, v in cam.items():
                if k not in ("id", "res_w", "res_h"):
                    cam[k] = np.array(v, dtype="float32")
            cam["center"] = normalize_screen_coordinates(cam["center"], w=cam["res_w"], h=cam["res_h"]).astype("float32")
            cam["focal_length"] = cam["focal_length"] / cam["res_w"] * 2.0
            if "translation" in cam:
                cam["translation"] = cam["translation"] / 1000
            cam["intrinsic"] = np.concatenate((cam["focal_length"], cam["center"], cam["radial_distortion"], cam["tangential_distortion"]))
            out[subject].append(cam)
    return out


class Human36mDataset:
    """Same surface as the reference class (``dataset[subject][action]`` -> {"positions", "cameras"[, "positions_3d"]}, ``subjects``,
    ``cameras``, ``skeleton``, ``fps``, ``define_actions``); ``positions`` are device tensors holding the 17 kept joints."""

    def __init__(self, path, remove_static_joints: bool = True, n_joints: int = 17, device=None):
        if n_joints != 17 or not remove_static_joints:
            raise NotImplementedError("manipose_amd: the HIP decoder is compiled for the 17-joint tree (data.joints=17)")
        self.device = _device(device)
        self._fps = 50
        self._n_joints = n_joints
        self._skeleton = Skeleton(h36m_skeleton().parents.tolist(), h36m_skeleton().joints_left, h36m_skeleton().joints_right,
                                  T_POSE_OPERATORS, joints_group=[[2, 3], [5, 6], [1, 4], [0, 7], [8, 9, 10], [15, 16], [12, 13], [11, 14]],
                                  joints_names=H36M_JOINT_NAMES_17)
        self._cameras = h36m_cameras()
        data = np.load(path, allow_pickle=True)["positions_3d"].item()
        self._data = {}
        for subject, actions in data.items():
            self._data[subject] = {}
            for action_name, positions in actions.items():
                raw = _upload(positions, self.device)
                kept = ingest_pose3d(raw, H36M_KEPT_JOINTS_17) if raw.shape[1] == 32 else raw       # joint selection only
                self._data[subject][action_name] = {"positions": kept, "cameras": self._cameras[subject]}

    def __getitem__(self, key):
        return self._data[key]

    subjects = property(lambda s: s._data.keys())
    fps = property(lambda s: s._fps)
    skeleton = property(lambda s: s._skeleton)
    cameras = property(lambda s: s._cameras)

    def define_actions(self, action=None):
        if action is None:
            return list(H36M_ACTIONS)
        if action not in H36M_ACTIONS:
            raise ValueError("Undefined action: {}".format(action))
        return [action]


def read_3d_data(dataset: Human36mDataset, subjects_filter=None, action_filter=None) -> Human36mDataset:
    """data/utils.py:29-58: adds ``positions_3d`` (one root-relative camera-frame sequence per camera) to every take."""
    identity = list(range(17))
    for subject in dataset.subjects:
        if subjects_filter is not None and subject not in subjects_filter:
            continue
        for action, anim in dataset[subject].items():
            if action_filter is not None and action not in action_filter:
                continue
            anim["positions_3d"] = [ingest_pose3d(anim["positions"], identity, cam["orientation"], cam["translation"], root_out=0)
                                    for cam in anim["cameras"]]
    return dataset


def create_2d_data(data_path, dataset: Human36mDataset) -> dict:
    """data/utils.py:9-26: {subject: {action: [per-camera (N, 17, 2) device tensors in normalised screen coordinates]}}."""
    keypoints = np.load(data_path, allow_pickle=True)["positions_2d"].item()
    out = {}
    for subject in keypoints.keys():
        out[subject] = {}
        for action in keypoints[subject]:
            seqs = []
            for cam_idx, kps in enumerate(keypoints[subject][action]):
                cam = dataset.cameras[subject][cam_idx]
                raw = _upload(kps, dataset.device)
                seqs.append(ingest_pose2d(raw, list(range(raw.shape[1])), cam["res_w"], cam["res_h"]))
            out[subject][action] = seqs
    return out


def fetch(subjects, dataset: Human36mDataset, keypoints: dict, action_filter=None, stride: int = 1, parse_3d_poses: bool = True):
    """data/utils.py:61-127 -> (poses_3d, poses_2d, actions, camera_params), one entry per subject x action x camera.  ``actions`` and
    ``camera_params`` hold one value per sequence (the reference repeats it per frame)."""
    out_3d, out_2d, out_actions, out_cams = [], [], [], []
    for subject in subjects:
        for action in keypoints[subject].keys():
            if action_filter is not None and not any(action.lower().split(" ")[0] == a for a in action_filter):
                continue
            cams = dataset.cameras[subject]
            poses_2d = keypoints[subject][action]
            for i in range(len(poses_2d)):
                out_2d.append(poses_2d[i])
                out_actions.append(action.split(" ")[0])
                out_cams.append(np.concatenate([cams[i]["intrinsic"], cams[i]["orientation"], cams[i]["translation"], np.array([i])]))
            if parse_3d_poses and "positions_3d" in dataset[subject][action]:
                poses_3d = dataset[subject][action]["positions_3d"]
                assert len(poses_3d) == len(poses_2d), "Camera count mismatch"
                out_3d.extend(poses_3d)
    if len(out_3d) == 0:
        out_3d = None
    if stride > 1:
        out_2d = [p[::stride].contiguous() for p in out_2d]
        if out_3d is not None:
            out_3d = [p[::stride].contiguous() for p in out_3d]
    return out_3d, out_2d, out_actions, out_cams


class Dataset3DHP:
    """data/dataset_3dhp.py:107-229: ``.poses`` / ``.poses_2d`` (lists of device tensors, H36M joint order, metres / normalised screen
    coordinates), ``.skeleton`` and the configuration fields the reference keeps."""

    def __init__(self, config, root_path, train: bool = True, MAE: bool = False, device=None):
        self.device = _device(device)
        self.data_type = config.data.dataset
        self.train = train
        self.keypoints_name = config.data.keypoints
        self.root_path = root_path
        self.data_augmentation = config.train.flip_aug
        self.reverse_augmentation = False
        self.batch_size = config.train.batch_size if train else config.train.batch_size_test
        self.action_filter = None if config.data.actions == "*" else config.data.actions.split(",")
        self.seq_len = config.data.seq_len
        self.test_aug = config.train.tta
        self.MAE = MAE
        sk = h36m_skeleton()
        self.skeleton = Skeleton(sk.parents.tolist(), sk.joints_left, sk.joints_right, T_POSE_OPERATORS, joints_names=H36M_JOINT_NAMES_17)
        self.poses, self.poses_2d = self.prepare_data(self.root_path, train=train)

    def prepare_data(self, path, train: bool = True):
        m = list(MAP_H36M_TO_MPI_JOINTS)
        out_3d, out_2d = [], []
        name = "data_train_3dhp.npz" if train else "data_test_3dhp.npz"
        data = np.load(os.path.join(path, name), allow_pickle=True)["data"].item()
        for seq in data.keys():
            takes = [data[seq][0][cam] for cam in data[seq][0].keys()] if train else [data[seq]]
            for anim in takes:
                frames = None
                if not train:
                    valid = np.flatnonzero(np.asarray(anim["valid"]).astype(bool)).astype(np.int32)
                    frames = torch.from_numpy(valid).to(self.device)
                w, h = (1920, 1080) if (not train and seq in ("TS5", "TS6")) else (2048, 2048)
                out_3d.append(ingest_pose3d(_upload(anim["data_3d"], self.device), m, root_raw=14, divisor=1000.0, frames=frames))
                out_2d.append(ingest_pose2d(_upload(anim["data_2d"], self.device), m, w, h, frames=frames))
        return out_3d, out_2d


__all__ = ["Human36mDataset", "Dataset3DHP", "read_3d_data", "create_2d_data", "fetch", "ingest_pose3d", "ingest_pose2d", "h36m_cameras",
           "normalize_screen_coordinates", "TRAIN_SUBJECTS", "TEST_SUBJECTS", "H36M_ACTIONS", "H36M_KEPT_JOINTS_17",
           "MAP_H36M_TO_MPI_JOINTS", "MAP_MPI_TO_H36M_JOINTS"]
