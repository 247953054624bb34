"""TUM RGB-D loader with the reference's constructor and item layout (datasets/tum.py:16-600), ingesting
on the device: PNGs are decoded on the host (PIL), staged in pinned memory, copied asynchronously and
turned into the float32 images of the SLAM path by gs_ingest_color_u8_f32 / gs_ingest_depth_u16_f32
(resize + depth scaling + optional colour normalisation); pose preprocessing runs through
gs_relative_pose_f32.  Items are returned ON THE DEVICE, ready for RGBDImages.

Differences from the reference, all deliberate: `device` argument (default "cuda"); decoding needs PIL
instead of imageio; use DataLoader(num_workers=0) -- the ingest stage is a device kernel, not a worker
process."""
import os
from typing import Optional, Union

import numpy as np
import torch

from . import datautils, tumutils
from ._base import SequenceDataset, window_ids

__all__ = ["TUM"]

_LAYOUT = ("TUM folder should look something like:\n\n| ├── basedir\n| │   ├── rgbd_dataset_freiburgX_NAME\n"
           "| │   │   ├── depth/\n| │   │   ├── rgb/\n| │   │   ├── accelerometer.txt\n| │   │   └── depth.txt\n"
           "| │   │   └── groundtruth.txt\n| │   │   └── rgb.txt\n| │   ├── ...")


class TUM(SequenceDataset):
    r"""Sequences of `seqlen` frames (every `dilation + 1`-th frame, starting every `stride` frames) from
    extracted TUM RGB-D sequences under `basedir`.  `__getitem__` returns, in this order and as enabled
    by the `return_*` flags: colours (L, H, W, 3), depths (L, H, W, 1) in metres, intrinsics (1, 4, 4),
    poses (L, 4, 4) relative to the first frame, frame-to-frame transforms (L, 4, 4), frame names,
    time stamps -- channels first when `channels_first`."""

    scaling_factor = 5000.0   # depth PNG units per metre

    def __init__(self, basedir: str, sequences: Union[tuple, str, None] = None, seqlen: int = 4,
                 dilation: Optional[int] = None, stride: Optional[int] = None, start: Optional[int] = None,
                 end: Optional[int] = None, height: int = 480, width: int = 640, channels_first: bool = False,
                 normalize_color: bool = False, *, return_depth: bool = True, return_intrinsics: bool = True,
                 return_pose: bool = True, return_transform: bool = True, return_names: bool = True,
                 return_timestamps: bool = True, device: Union[torch.device, str] = "cuda"):
        super().__init__()
        basedir = os.path.normpath(basedir)
        self.return_depth, self.return_intrinsics = return_depth, return_intrinsics
        self.return_pose, self.return_transform = return_pose, return_transform
        self.return_names, self.return_timestamps = return_names, return_timestamps
        self.load_poses = return_pose or return_transform
        self._init_common(seqlen, dilation, stride, start, end, height, width, channels_first, normalize_color,
                          device, quoted=True)

        if isinstance(sequences, str):
            if not os.path.isfile(sequences):
                raise ValueError("incorrect filename: {} doesn't exist".format(sequences))
            with open(sequences, "r") as f:
                sequences = tuple(f.read().split("\n"))
        elif not (sequences is None or isinstance(sequences, tuple)):
            raise TypeError('"sequences" should either be path to .txt file or tuple of sequence names or None, '
                            " but was of type {0} instead".format(type(sequences)))
        if isinstance(sequences, tuple) and len(sequences) == 0:
            raise ValueError('"sequences" must have atleast one element. Got len(sequences)=0')

        seq_dirs = []
        for item in os.listdir(basedir):
            if not os.path.isdir(os.path.join(basedir, item)):
                continue
            parts = item.split("_")
            if len(parts) < 4 or parts[0] != "rgbd" or parts[1] != "dataset" or parts[2][:-1] != "freiburg":
                raise ValueError('Incorrect folder names in "basedir" ({0}). Folder names of extracted .tgz files '
                                 'from TUM should follow the following naming convention: '
                                 '"rgbd_dataset_freiburgX_NAME". Got "{1}".'.format(basedir, item))
            if sequences is None or item in sequences:
                seq_dirs.append(os.path.join(basedir, item))
        if not seq_dirs:
            raise ValueError('Incorrect folder structure in basedir ("{0}"). '.format(basedir) + _LAYOUT)
        if sequences is not None and len(seq_dirs) != len(sequences):
            raise ValueError('"sequences" contains sequences not available in basedir:\n"sequences" contains: '
                             + ", ".join(sequences) + '\n"basedir" contains: '
                             + ", ".join(map(os.path.basename, seq_dirs)) + "\n" + _LAYOUT)

        self.colorfiles, self.depthfiles, self.poses, self.framenames, self.timestamps = [], [], [], [], []
        for seq_dir in seq_dirs:
            files = {}
            for key, fname, label in (("rgb", "rgb.txt", '"rgb.txt" file'), ("depth", "depth.txt", '"depth.txt" file'),
                                      ("pose", "groundtruth.txt", 'poses file ("groundtruth.txt")')):
                if key == "pose" and not self.load_poses:
                    files[key] = None
                    continue
                files[key] = os.path.join(seq_dir, fname)
                if not os.path.isfile(files[key]):
                    raise ValueError("Missing {0} in {1}. ".format(label, files[key]) + _LAYOUT)
            name = os.path.basename(seq_dir)
            assoc, stamps = self._findAssociations(files["rgb"], files["depth"], files["pose"])
            for a in assoc:
                if a[0][:3] != "rgb" or a[1][:5] != "depth":
                    raise ValueError("Incorrect reading from TUM associations")
            colors = [os.path.normpath(os.path.join(seq_dir, a[0])) for a in assoc]
            depths = [os.path.normpath(os.path.join(seq_dir, a[1])) for a in assoc]
            names = [name.strip("/\\") + "/" + a[0][3:-4] for a in assoc]
            for ids in window_ids(len(assoc), self.seqlen, self.dilation, self.stride):
                self.colorfiles.append([colors[i] for i in ids])
                self.depthfiles.append([depths[i] for i in ids])
                self.framenames.append(", ".join(names[i] for i in ids))
                self.timestamps.append([stamps[i] for i in ids])
                if self.load_poses:
                    self.poses.append([assoc[i][2] for i in ids])
        self.num_sequences = len(self.colorfiles)
        self._set_intrinsics([[525.0, 0, 319.5, 0], [0, 525.0, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])

    def _homogenPoses(self, poses_point_quaternion):
        return [datautils.pointquaternion_to_homogeneous(p) for p in poses_point_quaternion]

    def __getitem__(self, idx: int):
        out = self._images_and_poses(idx, self._homogenPoses(self.poses[idx]) if self.load_poses else None)
        if self.return_names:
            out.append(self.framenames[idx])
        if self.return_timestamps:
            out.append("\n".join("rgb {} depth {} pose {}".format(*t) for t in self.timestamps[idx]))
        return tuple(out)

    def _findAssociations(self, rgb_text_file: str, depth_text_file: str, poses_text_file: Optional[str] = None,
                          max_difference: float = 0.02):
        r"""(rgb path, depth path[, (7,) point-quaternion]) per matched frame and the matched stamps."""
        rgb = tumutils.read_file_list(rgb_text_file, self.start, self.end)
        depth = tumutils.read_file_list(depth_text_file)
        pairs = tumutils.associate(rgb, depth, 0, float(max_difference))
        if poses_text_file is None:
            return ([(rgb[a][0], depth[b][0]) for a, b in pairs], [(a, b, None) for a, b in pairs])
        traj = tumutils.read_trajectory(poses_text_file, matrix=False)
        rgb_of_depth = {b: a for a, b in pairs}
        triples = [(rgb_of_depth[b], b, p) for b, p in tumutils.associate(rgb_of_depth, traj, 0, float(max_difference))]
        return ([(rgb[a][0], depth[b][0], np.array(traj[p], dtype=np.float32)) for a, b, p in triples], list(triples))
