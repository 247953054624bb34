"""ScanNet loader with the reference's constructor and item layout (datasets/scannet.py:18-407): sequences are
described by one metadata `.txt` per sequence in `seqmetadir` (`sceneXXXX_XX-seq_Y.txt`, one line per frame:
`color <p> depth <p> pose <p> label-filt <p> ... intrinsic_depth <p>` with paths relative to `basedir`).  The
per-pixel stage (colour INTER_LINEAR resize / normalisation, uint16 depth -> metres with scale 1000 and
INTER_NEAREST resize) and the pose algebra run on the device (_base.py: gs_ingest_*, gs_relative_pose_f32); the
`nyu40` label images (uint8, resized with INTER_NEAREST, optionally remapped to `scannet20`) are returned as
float tensors like the reference does."""
import glob
import os
from collections import OrderedDict
from typing import Union

import numpy as np
import torch

from . import datautils
from ._base import SequenceDataset, read_png

__all__ = ["Scannet", "get_color_encoding", "nyu40_to_scannet20"]

_NYU40 = [("unlabeled", (0, 0, 0)), ("wall", (174, 199, 232)), ("floor", (152, 223, 138)), ("cabinet", (31, 119, 180)),
          ("bed", (255, 187, 120)), ("chair", (188, 189, 34)), ("sofa", (140, 86, 75)), ("table", (255, 152, 150)),
          ("door", (214, 39, 40)), ("window", (197, 176, 213)), ("bookshelf", (148, 103, 189)),
          ("picture", (196, 156, 148)), ("counter", (23, 190, 207)), ("blinds", (178, 76, 76)),
          ("desk", (247, 182, 210)), ("shelves", (66, 188, 102)), ("curtain", (219, 219, 141)),
          ("dresser", (140, 57, 197)), ("pillow", (202, 185, 52)), ("mirror", (51, 176, 203)),
          ("floormat", (200, 54, 131)), ("clothes", (92, 193, 61)), ("ceiling", (78, 71, 183)),
          ("books", (172, 114, 82)), ("refrigerator", (255, 127, 14)), ("television", (91, 163, 138)),
          ("paper", (153, 98, 156)), ("towel", (140, 153, 101)), ("showercurtain", (158, 218, 229)),
          ("box", (100, 125, 154)), ("whiteboard", (178, 127, 135)), ("person", (120, 185, 128)),
          ("nightstand", (146, 111, 194)), ("toilet", (44, 160, 44)), ("sink", (112, 128, 144)),
          ("lamp", (96, 207, 209)), ("bathtub", (227, 119, 194)), ("bag", (213, 92, 176)),
          ("otherstructure", (94, 106, 211)), ("otherfurniture", (82, 84, 163)), ("otherprop", (100, 85, 144))]
# nyu40 ids kept by the scannet20 palette, in scannet20 order (ids 1..12 map to themselves); everything else -> 0
_SCANNET20_IDS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]


def get_color_encoding(seg_classes):
    """label name -> colour of the `"nyu40"` or `"scannet20"` palette (reference: datasets/scannet.py:410-480)"""
    if seg_classes.lower() == "nyu40":
        return OrderedDict(_NYU40)
    if seg_classes.lower() == "scannet20":
        return OrderedDict(_NYU40[i] for i in _SCANNET20_IDS)


def nyu40_to_scannet20(label):
    """remaps a `nyu40` label image to the contiguous `scannet20` ids, in place (datasets/scannet.py:483-531)"""
    lut = np.zeros(256, dtype=label.dtype)
    lut[:41] = 0
    for tar, src in enumerate(_SCANNET20_IDS):
        lut[src] = tar
    lut[41:] = np.arange(41, 256).astype(label.dtype)   # ids beyond the nyu40 range are left alone
    label[...] = lut[label]
    return label


class Scannet(SequenceDataset):
    r"""Items, as enabled by the `return_*` flags: colours (L, H, W, 3), depths (L, H, W, 1) in metres, intrinsics
    (1, 4, 4), poses relative to the first frame (L, 4, 4), frame-to-frame transforms (L, 4, 4), the sequence name,
    labels (L, H, W, 1)."""

    scaling_factor = 1000.0
    native_size = (480, 640)   # the intrinsics are scaled by height / 480 and width / 640 (datasets/scannet.py:89-90)

    def __init__(self, basedir: str, seqmetadir: str, scenes: Union[tuple, str, None], start: int = 0, end: int = -1,
                 height: int = 480, width: int = 640, seg_classes: str = "scannet20", channels_first: bool = False,
                 normalize_color: bool = False, *, return_depth: bool = True, return_intrinsics: bool = True,
                 return_pose: bool = True, return_transform: bool = True, return_names: bool = True,
                 return_labels: bool = True, device: Union[torch.device, str] = "cuda"):
        super().__init__()
        basedir = os.path.normpath(basedir)
        self.device = torch.device(device)
        self.height, self.width = height, width
        self.height_downsample_ratio, self.width_downsample_ratio = float(height) / 480, float(width) / 640
        self.seg_classes, self.channels_first, self.normalize_color = seg_classes, channels_first, normalize_color
        self.return_depth, self.return_intrinsics = return_depth, return_intrinsics
        self.return_pose, self.return_transform = return_pose, return_transform
        self.return_names, self.return_labels = return_names, return_labels
        self.load_poses = return_pose or return_transform
        self.color_encoding = get_color_encoding(seg_classes)
        self._stage = {}
        self.start, self.end = start, end
        full_sequence = end == -1
        if start < 0:
            raise ValueError("Start frame cannot be less than 0.")
        if not (end == -1 or end > start):
            raise ValueError("End frame ({}) should be equal to -1 or greater than start ({})".format(end, start))
        self.seqlen = self.end - self.start
        if isinstance(scenes, str):
            if not os.path.isfile(scenes):
                raise ValueError("incorrect filename: {} doesn't exist".format(scenes))
            with open(scenes, "r") as f:
                scenes = tuple(f.read().split("\n"))
        elif not (scenes is None or isinstance(scenes, tuple)):
            raise TypeError("scenes should either be path to split.txt or tuple of scenes or None, but was of type %r "
                            "instead" % type(scenes))
        self.colorfiles, self.depthfiles, self.posefiles = [], [], []
        self.labelfiles, self.intrinsicsfiles, self.seqnames = [], [], []
        for seqmetapath in sorted(glob.glob(os.path.join(seqmetadir, "*.txt"))):   # natsorted for plain names
            scene_name = os.path.basename(seqmetapath).split("-")[0]
            if scenes is not None and scene_name not in scenes:
                continue
            with open(seqmetapath, "r") as f:
                lines = f.readlines()
            if full_sequence:
                self.end = len(lines)
                self.seqlen = self.end - self.start
            if self.seqlen > len(lines):
                raise ValueError("sequence length can't be larger than dataset sequence length but it was: %r > %r"
                                 % (self.seqlen, len(lines)))
            lines = lines[self.start:self.end]
            cols = {"color": [], "depth": [], "pose": [], "label-filt": [], "intrinsic_depth": []}
            for line in lines:
                f = line.strip().split()
                for pos, key in ((0, "color"), (2, "depth"), (4, "pose"), (6, "label-filt"), (14, "intrinsic_depth")):
                    if f[pos] != key:
                        raise ValueError("incorrect reading from scannet metadata")
                    cols[key].append(os.path.join(basedir, f[pos + 1]))
            self.colorfiles.append(cols["color"])
            self.depthfiles.append(cols["depth"])
            self.posefiles.append(cols["pose"])
            self.labelfiles.append(cols["label-filt"])
            self.intrinsicsfiles.append(cols["intrinsic_depth"][0])
            self.seqnames.append(os.path.basename(seqmetapath).split(".")[0])
        self.num_sequences = len(self.colorfiles)

    def _preprocess_intrinsics(self, intrinsics):
        return datautils.scale_intrinsics(intrinsics, self.height_downsample_ratio, self.width_downsample_ratio)[None]

    def _preprocess_label(self, label: np.ndarray):
        """INTER_NEAREST to (height, width) -- OpenCV's resizeNN picks source index min(floor(dst * scale), n - 1) --
        then the optional nyu40 -> scannet20 remap; (H, W, 1)."""
        H0, W0 = label.shape[:2]
        if (H0, W0) != (self.height, self.width):
            ys = np.minimum(np.floor(np.arange(self.height) * (H0 / self.height)).astype(np.int64), H0 - 1)
            xs = np.minimum(np.floor(np.arange(self.width) * (W0 / self.width)).astype(np.int64), W0 - 1)
            label = label[ys][:, xs]
        label = np.array(label, dtype=np.uint8)
        if self.seg_classes.lower() == "scannet20":
            label = nyu40_to_scannet20(label)
        return label[..., None]

    def __getitem__(self, idx: int):
        poses_np = None
        if self.load_poses:
            poses_np = [np.loadtxt(p).astype(np.float32) for p in self.posefiles[idx]]
        intr = self.intrinsics_for(idx) if self.return_intrinsics else None
        out = [torch.stack([self._preprocess_color(read_png(p)) for p in self.colorfiles[idx]], 0)]
        if self.return_depth:
            out.append(torch.stack([self._preprocess_depth(read_png(p)) for p in self.depthfiles[idx]], 0))
        if self.return_intrinsics:
            out.append(intr)
        if self.load_poses:
            poses = torch.from_numpy(np.stack(poses_np)).float().to(self.device)
            if self.return_pose:
                out.append(self._preprocess_poses(poses))
            if self.return_transform:
                out.append(self._frame_to_frame(poses))
        if self.return_names:
            out.append(self.seqnames[idx])
        if self.return_labels:
            labels = [torch.from_numpy(self._preprocess_label(read_png(p))) for p in self.labelfiles[idx]]
            out.append(torch.stack(labels, 0).float().to(self.device))
        return tuple(out)

    def intrinsics_for(self, idx):
        K = np.loadtxt(self.intrinsicsfiles[idx]).astype(float)
        return torch.from_numpy(self._preprocess_intrinsics(K)).float().to(self.device)
