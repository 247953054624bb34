"""What the RGB-D sequence loaders share: argument checks of the sequence sampler (seqlen / dilation /
stride / start / end), the host -> device staging of decoded frames and the device ingest stage
(gs_ingest_color_u8_f32, gs_ingest_depth_u16_f32, gs_relative_pose_f32).  Subclasses discover files
and parse poses in their dataset's format."""
import numpy as np
import torch

from . import datautils


def read_png(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def window_ids(n_frames, seqlen, dilation, stride):
    """Frame indices of every extracted sequence: `seqlen` frames `dilation + 1` apart, a new sequence every
    `stride` frames, while the last frame still exists."""
    offsets = np.arange(seqlen) * (dilation + 1)
    out = []
    for first in range(0, n_frames, stride):
        if first + offsets[-1] >= n_frames:
            break
        out.append(first + offsets)
    return out


class SequenceDataset(torch.utils.data.Dataset):
    scaling_factor = 5000.0      # depth PNG units per metre
    native_size = (480, 640)

    def _init_common(self, seqlen, dilation, stride, start, end, height, width, channels_first, normalize_color,
                     device, quoted):
        q = (lambda n: '"%s"' % n) if quoted else (lambda n: n)
        self.device = torch.device(device)
        self.height, self.width = height, width
        self.height_downsample_ratio = float(height) / self.native_size[0]
        self.width_downsample_ratio = float(width) / self.native_size[1]
        self.channels_first, self.normalize_color = channels_first, normalize_color
        if not isinstance(seqlen, int):
            raise TypeError("{0} must be int. Got {1}.".format(q("seqlen"), type(seqlen)))
        if not (isinstance(stride, int) or stride is None):
            raise TypeError("{0} must be int or None. Got {1}.".format(q("stride"), type(stride)))
        if not (isinstance(dilation, int) or dilation is None):
            raise TypeError("dilation must be int or None. Got {0}.".format(type(dilation)))
        dilation = 0 if dilation is None else dilation
        stride = seqlen * (dilation + 1) if stride is None else stride
        self.seqlen, self.stride, self.dilation = seqlen, stride, dilation
        if seqlen < 0:
            raise ValueError("{0} must be positive. Got {1}.".format(q("seqlen"), seqlen))
        if dilation < 0:
            raise ValueError('"dilation" must be positive. Got {0}.'.format(dilation))
        if stride < 0:
            raise ValueError("{0} must be positive. Got {1}.".format(q("stride"), stride))
        for name, v in (("start", start), ("end", end)):
            if not (isinstance(v, int) or v is None):
                raise TypeError("{0} must be int or None. Got {1}.".format(q(name), type(v)))
        start = 0 if start is None else start
        self.start, self.end = start, end
        if start < 0:
            raise ValueError(('"start" must be None or positive. Got {0}.' if quoted else
                              "start must be positive. Got {0}.").format(stride))
        if not (end is None or end > start):
            raise ValueError("{0} ({1}) must be None or greater than start ({2})".format(q("end"), end, start))
        self._stage = {}

    def _set_intrinsics(self, K):
        self.intrinsics = datautils.scale_intrinsics(torch.tensor(K).float(), self.height_downsample_ratio,
                                                     self.width_downsample_ratio).unsqueeze(0).to(self.device)

    def __len__(self):
        return self.num_sequences

    # ------------------------------------------------------------------ host -> device staging
    def _to_device(self, key, arr):
        """pinned staging buffer (reused per shape / dtype) + asynchronous copy on the current stream"""
        t = torch.from_numpy(np.array(arr, order="C"))   # decoded images are read-only views: copy
        if self.device.type != "cuda":
            return t.to(self.device)
        slot = self._stage.get((key, t.shape, t.dtype))
        if slot is None:
            slot = self._stage[(key, t.shape, t.dtype)] = [torch.empty(t.shape, dtype=t.dtype).pin_memory(),
                                                          torch.cuda.Event()]
        pinned, done = slot
        done.synchronize()          # the previous copy out of this buffer has finished
        pinned.copy_(t)
        dev = pinned.to(self.device, non_blocking=True)
        done.record()
        return dev

    def _preprocess_color(self, color: np.ndarray):
        from .. import ops
        if color.ndim == 2:
            color = np.repeat(color[..., None], 3, -1)
        raw = self._to_device("color", color[..., :3].astype(np.uint8, copy=False))
        out = ops.ingest_color(raw, self.height, self.width, self.normalize_color)
        return out.permute(2, 0, 1).contiguous() if self.channels_first else out

    def _preprocess_depth(self, depth: np.ndarray):
        from .. import ops
        raw = self._to_device("depth", depth.astype(np.uint16, copy=False))
        out = ops.ingest_depth(raw, self.height, self.width, self.scaling_factor)
        return out.unsqueeze(0) if self.channels_first else out.unsqueeze(-1)

    def _preprocess_poses(self, poses: torch.Tensor):
        """poses relative to the first frame of the sequence"""
        from .. import ops
        return ops.relative_pose(poses[:1].expand_as(poses).contiguous(), poses)

    def _frame_to_frame(self, poses: torch.Tensor):
        """identity, then inv(pose[i-1]) . pose[i]"""
        from .. import ops
        eye = torch.eye(4, dtype=torch.float32, device=self.device)[None]
        if len(poses) < 2:
            return eye
        return torch.cat([eye, ops.relative_pose(poses[:-1].contiguous(), poses[1:].contiguous())], 0)

    def _images_and_poses(self, idx, poses_np):
        """the leading items every loader returns: colours[, depths][, intrinsics][, poses][, transforms]"""
        out = [torch.stack([self._preprocess_color(read_png(p)) for p in self.colorfiles[idx]], 0)]
        if self.return_depth:
            out.append(torch.stack([self._preprocess_depth(read_png(p)) for p in self.depthfiles[idx]], 0))
        if self.return_intrinsics:
            out.append(self.intrinsics)
        if self.load_poses:
            poses = torch.from_numpy(np.stack(poses_np)).float().to(self.device)
            if self.return_pose:
                out.append(self._preprocess_poses(poses))
            if self.return_transform:
                out.append(self._frame_to_frame(poses))
        return out
