"""ICL-NUIM loader with the reference's constructor and item layout (datasets/icl.py:16-572); the
per-pixel stage and the pose algebra run on the device (see _base.py).  Trajectory folders
`living_room_trajX_frei_png/` hold `rgb/`, `depth/`, `associations.txt` (one line per frame:
`id depth/N.png id rgb/N.png`) and `livingRoomXn.gt.sim` (3x4 camera-to-world matrices, one per frame,
4 lines apart)."""
import os
import warnings
from typing import Optional, Union

import numpy as np
import torch

from ._base import SequenceDataset, window_ids

__all__ = ["ICL"]


def _layout():
    s = "ICL folder should look something like:\n\n| ├── basedir\n"
    for i in range(4):
        s += ("| │   ├── living_room_traj{0}_frei_png\n| │   │   ├── depth/\n| │   │   ├── rgb/\n"
              "| │   │   ├── associations.txt\n| │   │   └── livingRoom{0}n.gt.sim\n").format(i)
    return s


def _is_traj_name(name):
    return name[:16] == "living_room_traj" and name[-9:] == "_frei_png"


class ICL(SequenceDataset):
    r"""Items, as enabled by the `return_*` flags: colours (L, H, W, 3), depths (L, H, W, 1) in metres,
    intrinsics (1, 4, 4) (note fy < 0), poses relative to the first frame (L, 4, 4), frame-to-frame
    transforms (L, 4, 4), frame names."""

    scaling_factor = 5000.0

    def __init__(self, basedir: str, trajectories: Union[tuple, str, None] = None, seqlen: int = 4,
                 dilation: Optional[int] = None, stride: Optional[int] = None, start: Optional[int] = None,
                 end: Optional[int] = None, height: int = 480, width: int = 640, channels_first: bool = False,
                 normalize_color: bool = False, *, return_depth: bool = True, return_intrinsics: bool = True,
                 return_pose: bool = True, return_transform: bool = True, return_names: bool = True,
                 device: Union[torch.device, str] = "cuda"):
        super().__init__()
        basedir = os.path.normpath(basedir)
        self.return_depth, self.return_intrinsics = return_depth, return_intrinsics
        self.return_pose, self.return_transform, self.return_names = return_pose, return_transform, return_names
        self.load_poses = return_pose or return_transform
        self._init_common(seqlen, dilation, stride, start, end, height, width, channels_first, normalize_color,
                          device, quoted=False)

        wanted = [f for f in os.listdir(basedir) if os.path.isdir(os.path.join(basedir, f)) and _is_traj_name(f)]
        if not wanted:
            raise ValueError('basedir ({0}) should contain trajectory folders with the following naming convention: '
                             '"living_room_trajX_frei_png". Found 0 folders with this naming convention.'.format(basedir))
        if isinstance(trajectories, str):
            if not os.path.isfile(trajectories):
                raise ValueError("incorrect filename: {} doesn't exist".format(trajectories))
            with open(trajectories, "r") as f:
                trajectories = tuple(f.read().split("\n"))
            wanted = list(trajectories)
        elif not (trajectories is None or isinstance(trajectories, tuple)):
            raise TypeError('"trajectories" should either be path to .txt file or tuple of trajectory names or None, '
                            " but was of type {0} instead".format(type(trajectories)))
        if isinstance(trajectories, tuple):
            if len(trajectories) == 0:
                raise ValueError('"trajectories" must have atleast one element. Got len(trajectories)=0')
            for t in trajectories:
                if not _is_traj_name(t):
                    raise ValueError('"trajectories" should only contain trajectory folder names of the following '
                                     'convention: "living_room_trajX_frei_png". It contained: {0}.'.format(t))
            wanted = list(trajectories)
        traj_dirs = [os.path.join(basedir, item) for item in os.listdir(basedir)
                     if os.path.isdir(os.path.join(basedir, item)) and item in wanted]
        if not traj_dirs:
            raise ValueError('Incorrect folder structure in basedir ("{0}"). '.format(basedir) + _layout())
        if trajectories is not None and len(traj_dirs) != len(trajectories):
            raise ValueError('"trajectories" contains trajectories not available in basedir:\ntrajectories contains: '
                             + ", ".join(trajectories) + "\nbasedir contains: "
                             + ", ".join(map(os.path.basename, traj_dirs)) + "\n" + _layout())

        self.colorfiles, self.depthfiles, self.posemetas, self.framenames = [], [], [], []
        for traj_dir in traj_dirs:
            name = os.path.basename(traj_dir)
            assoc_file = os.path.join(traj_dir, "associations.txt")
            if not os.path.isfile(assoc_file):
                raise ValueError('Missing associations file ("associations.txt") in {0}. '.format(traj_dir) + _layout())
            poses_file, n_pose_lines = None, 0
            if self.load_poses:
                num = traj_dir[traj_dir.index("living_room_traj") + 16:].split("_")[0]
                poses_file = os.path.join(traj_dir, "livingRoom{0}n.gt.sim".format(num))
                if not os.path.isfile(poses_file):
                    raise ValueError('Missing ground truth poses file ("{0}") in {1}. '.format(poses_file, basedir)
                                     + _layout())
                with open(poses_file, "r") as f:
                    n_pose_lines = sum(1 for _ in f)
            with open(assoc_file, "r") as f:
                lines = f.readlines()
            last = len(lines) if self.end is None else self.end
            if last > len(lines):
                warnings.warn("end was larger than number of frames in trajectory: {0} > {1} (trajectory: {2})".format(
                    last, len(lines), name))
            if name == "living_room_traj0_frei_png":   # its pose file is one pose short (as the reference notes)
                lines = lines[:-1]
            lines = lines[self.start:last]
            colors, depths, pose_lines, names = [], [], [], []
            for k, line in enumerate(lines):
                f = line.strip().split()
                if f[3][:3] != "rgb" or f[1][:5] != "depth":
                    raise ValueError("incorrect reading from ICL associations")
                colors.append(os.path.normpath(os.path.join(traj_dir, f[3])))
                depths.append(os.path.normpath(os.path.join(traj_dir, f[1])))
                if self.load_poses:
                    if k * 4 > n_pose_lines:
                        raise ValueError('{0}th pose should start from line {1} of file "{2}", but said file has only '
                                         "{3} lines.".format(k, k * 4, os.path.join(*poses_file.split(os.sep)[-2:]),
                                                             n_pose_lines))
                    pose_lines.append(k * 4)
                names.append(os.path.join(name, f[1][6:].split(".")[0]))
            for ids in window_ids(len(colors), self.seqlen, self.dilation, self.stride):
                self.colorfiles.append([colors[i] for i in ids])
                self.depthfiles.append([depths[i] for i in ids])
                self.framenames.append(", ".join(names[i] for i in ids))
                if self.load_poses:
                    self.posemetas.append({"file": poses_file, "line_nums": [pose_lines[i] for i in ids]})
        self.num_sequences = len(self.colorfiles)
        self._set_intrinsics([[481.20, 0, 319.5, 0], [0, -480.0, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])

    def _loadPoses(self, pose_path, start_lines):
        """4x4 float32 poses whose 3 matrix rows start at the given lines of the .gt.sim file"""
        with open(pose_path, "r") as f:
            lines = f.readlines()
        poses = []
        for first in start_lines:
            rows = [lines[first + r].strip().split() for r in range(3)]
            if any(len(r) != 4 for r in rows):
                raise ValueError("Faulty poses file: Expected line {0} of the poses file {1} to contain pose matrix "
                                 'values, but it didn\'t. You can download "Global_RT_Trajectory_GT" from here:\n'
                                 "https://www.doc.ic.ac.uk/~ahanda/VaFRIC/iclnuim.html".format(first, pose_path))
            poses.append(np.array(rows + [[0.0, 0.0, 0.0, 1.0]], dtype=np.float32))
        return poses

    def __getitem__(self, idx: int):
        poses = None
        if self.load_poses:
            meta = self.posemetas[idx]
            poses = self._loadPoses(meta["file"], meta["line_nums"])
        out = self._images_and_poses(idx, poses)
        if self.return_names:
            out.append(self.framenames[idx])
        return tuple(out)
