"""RGB-D frame container with the interface of the reference's `gradslam.RGBDImages`
(structures/rgbdimages.py:13-915).  The lazily evaluated maps (`vertex_map`, `normal_map`,
`global_vertex_map`, `global_normal_map`, `valid_depth_mask`) are produced by the HIP kernels
gs_frame_maps_f32 / gs_global_maps_f32 instead of the reference's einsum / cross / norm chains
(rgbdimages.py:643-762); there is no PyTorch fallback for them.

Layout: the kernels work on channels-last (H, W, C) frames; a channels-first container is
converted at the kernel boundary, exactly like the reference converts before the SLAM path
(structures/utils.py:39)."""
from typing import Optional, Union

import torch

__all__ = ["RGBDImages"]


class RGBDImages(object):
    r"""Batch of RGB-D sequences: rgb (B, L, H, W, 3), depth (B, L, H, W, 1), intrinsics (B, 1, 4, 4)
    and optional poses (B, L, 4, 4) (channels-first variants supported through `channels_first`)."""

    _INTERNAL_TENSORS = ["_rgb_image", "_depth_image", "_intrinsics", "_poses", "_pixel_pos", "_vertex_map",
                         "_normal_map", "_global_vertex_map", "_global_normal_map"]

    def __init__(
        self,
        rgb_image: torch.Tensor,
        depth_image: torch.Tensor,
        intrinsics: torch.Tensor,
        poses: Optional[torch.Tensor] = None,
        channels_first: bool = False,
        device: Union[torch.device, str, None] = None,
        *,
        pixel_pos: Optional[torch.Tensor] = None,
    ):
        # argument checks, table driven (same exception types and messages as the reference ctor)
        required = (("rgb_image", rgb_image), ("depth_image", depth_image), ("intrinsics", intrinsics))
        optional = (("poses", poses), ("pixel_pos", pixel_pos))
        for name, val in required:
            if not torch.is_tensor(val):
                raise TypeError("Expected {} to be of type tensor; got {}".format(name, type(val)))
        if not (poses is None or torch.is_tensor(poses)):
            raise TypeError("Expected poses to be of type tensor or None; got {}".format(type(poses)))
        if not isinstance(channels_first, bool):
            raise TypeError("Expected channels_first to be of type bool; got {}".format(type(channels_first)))
        if not (pixel_pos is None or torch.is_tensor(pixel_pos)):
            raise TypeError("Expected pixel_pos to be of type tensor or None; got {}".format(type(pixel_pos)))
        self._channels_first = channels_first
        for name, val, nd in (("rgb_image", rgb_image, 5), ("depth_image", depth_image, 5),
                              ("intrinsics", intrinsics, 4), ("poses", poses, 4)):
            if val is not None and val.ndim != nd:
                raise ValueError("{} should have ndim={}, but had ndim={}".format(name, nd, val.ndim))

        full = tuple(rgb_image.shape)
        c = self.cdim
        self._rgb_image_shape = rgb_image.shape
        self._depth_shape = self._depth_image_shape = full[:c] + (1,) + full[c + 1:]
        self._intrinsics_shape = (full[0], 1, 4, 4)
        self._poses_shape = full[:2] + (4, 4)
        self._pixel_pos_shape = (full[:c] + (3,) + full[c + 1:]) if channels_first else (full[:c] + full[c + 1:] + (3,))
        if full[c] != 3:
            raise ValueError("Expected rgb_image to have 3 channels on dimension {0}. Got {1} instead".format(c, full[c]))
        for name, val, want in (("depth_image", depth_image, self._depth_shape),
                                ("intrinsics", intrinsics, self._intrinsics_shape), ("poses", poses, self._poses_shape),
                                ("pixel_pos", pixel_pos, self._pixel_pos_shape)):
            if val is not None and tuple(val.shape) != tuple(want):
                raise ValueError("Expected {0} to have shape {1}. Got {2} instead".format(name, want, val.shape))
        devices = {t.device for _, t in required + optional if t is not None}
        if len(devices) != 1:
            raise ValueError("All inputs must be on same device, but got more than 1 device: {}".format(devices))

        self._rgb_image = rgb_image if device is None else rgb_image.to(device)
        self.device = self._rgb_image.device
        self._depth_image, self._intrinsics = depth_image.to(self.device), intrinsics.to(self.device)
        self._poses = None if poses is None else poses.to(self.device)
        self._pixel_pos = None if pixel_pos is None else pixel_pos.to(self.device)
        if self._pixel_pos is not None:
            # the back-projection kernel generates (u, v, 1) itself: a caller-supplied grid must be that grid.  The
            # shape was checked above; with channels_first the grid is (B, L, 3, H, W): compared channels-last
            hh, ww = (rgb_image.shape[3], rgb_image.shape[4]) if channels_first else (rgb_image.shape[2], rgb_image.shape[3])
            v, u = torch.meshgrid(torch.arange(hh, dtype=torch.float32, device=self.device),
                                  torch.arange(ww, dtype=torch.float32, device=self.device), indexing="ij")
            std = torch.stack([u, v, torch.ones_like(u)], -1)
            pp = self._pixel_pos.float()
            if channels_first:
                pp = pp.permute(0, 1, 3, 4, 2)
            if not bool((pp == std).all()):
                raise NotImplementedError("gradslam_amd back-projects on the regular pixel grid (u, v, 1); a custom "
                                          "pixel_pos is not supported")

        self._vertex_map = None
        self._global_vertex_map = None
        self._normal_map = None
        self._global_normal_map = None
        self._valid_depth_mask = None
        self._alpha_cache = None  # (sigma, alpha map) computed by the same kernel as the vertex map
        self._sigma_hint = None   # sigma a later consumer (PointFusion) will ask the alpha map for

        self._B, self._L = self._rgb_image.shape[:2]
        self.h = self._rgb_image.shape[3] if self._channels_first else self._rgb_image.shape[2]
        self.w = self._rgb_image.shape[4] if self._channels_first else self._rgb_image.shape[3]
        self.shape = (self._B, self._L, self.h, self.w)

    # ------------------------------------------------------------------ indexing
    def __getitem__(self, index):
        if isinstance(index, tuple) or isinstance(index, int):
            slices = ()
            if isinstance(index, int):
                slices += (slice(index, index + 1),) + (slice(None, None),)
            elif len(index) > 2:
                raise IndexError("Only batch and sequences can be indexed")
            else:
                for x in index:
                    slices += (slice(x, x + 1),) if isinstance(x, int) else (x,)
                if len(slices) == 1:
                    slices += (slice(None, None),)
            new_rgb = self._rgb_image[slices[0], slices[1]]
            if new_rgb.shape[0] == 0:
                raise IndexError("Incorrect indexing at dimension 0, make sure range is within 0 and {0}".format(self._B))
            if new_rgb.shape[1] == 0:
                raise IndexError("Incorrect indexing at dimension 1, make sure range is within 0 and {0}".format(self._L))
            # a view of a validated object needs no second validation (frames[:, s] sits in every SLAM loop): copy
            # the fields, slice the tensors, fix the shape-dependent ones
            other = object.__new__(RGBDImages)
            other.__dict__.update(self.__dict__)
            other._rgb_image = new_rgb
            other._depth_image = self._depth_image[slices[0], slices[1]]
            whole = isinstance(slices[0], slice) and slices[0] == slice(None, None)
            other._intrinsics = self._intrinsics if whole else self._intrinsics[slices[0], :]
            for k in self._INTERNAL_TENSORS:
                if k in ("_rgb_image", "_depth_image", "_intrinsics"):
                    continue
                v = getattr(self, k)
                if torch.is_tensor(v):
                    setattr(other, k, v[slices[0], slices[1]])
            ac = self._alpha_cache
            other._alpha_cache = None if ac is None else (ac[0], ac[1][slices[0], slices[1]])
            other._valid_depth_mask = None
            full = tuple(new_rgb.shape)
            c = other.cdim
            other._rgb_image_shape = new_rgb.shape
            other._depth_shape = other._depth_image_shape = full[:c] + (1,) + full[c + 1:]
            other._intrinsics_shape = (full[0], 1, 4, 4)
            other._poses_shape = full[:2] + (4, 4)
            other._pixel_pos_shape = ((full[:c] + (3,) + full[c + 1:]) if other._channels_first
                                      else (full[:c] + full[c + 1:] + (3,)))   # (as the constructor)
            other._B, other._L = full[0], full[1]
            other.shape = (other._B, other._L, other.h, other.w)
            return other
        raise IndexError(index)

    def __len__(self):
        return self._B

    # ------------------------------------------------------------------ plain properties
    @property
    def channels_first(self):
        return self._channels_first

    @property
    def cdim(self):
        return 2 if self.channels_first else 4

    rgb_image = property(lambda self: self._rgb_image)
    depth_image = property(lambda self: self._depth_image)
    intrinsics = property(lambda self: self._intrinsics)
    poses = property(lambda self: self._poses)

    @property
    def pixel_pos(self):
        """(B, L, H, W, 3) = [column u, row v, 1] per pixel once the vertex map has been computed, None before
        (reference: structures/rgbdimages.py:308-317, :647-661).  The HIP kernel generates the grid in registers; this
        tensor exists for API parity only."""
        if self._pixel_pos is None and self._vertex_map is not None:
            B, L, H, W = self.shape
            v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=self.device),
                                  torch.arange(W, dtype=torch.float32, device=self.device), indexing="ij")
            grid = torch.stack([u, v, torch.ones_like(u)], -1)
            self._pixel_pos = grid.view(1, 1, H, W, 3).repeat(B, L, 1, 1, 1)
        return self._pixel_pos

    @property
    def has_poses(self):
        return self._poses is not None

    @property
    def valid_depth_mask(self):
        if self._valid_depth_mask is None:
            self._valid_depth_mask = self._depth_image > 0
        return self._valid_depth_mask

    # ------------------------------------------------------------------ HIP-backed lazy maps
    def _cl(self, t):
        """channels-last contiguous float32 view/copy of a (B, L, ...) image tensor."""
        if self.channels_first:
            t = t.permute(0, 1, 3, 4, 2)
        return t.contiguous().float()

    def _from_cl(self, t):
        return t.permute(0, 1, 4, 2, 3).contiguous() if self.channels_first else t

    def _wants_grad(self):
        return torch.is_grad_enabled() and (self._depth_image.requires_grad or self._intrinsics.requires_grad)

    def _compute_local_maps(self, sigma=None):
        """vertex + normal (+ alpha when sigma is given) for every (b, l) frame: one kernel each.
        When depth requires grad the maps stay on the autograd tape (hand-written HIP backward)."""
        from .. import ops
        B, L, H, W = self.shape
        K = self._intrinsics.contiguous().float()
        if sigma is None:
            sigma = getattr(self, "_sigma_hint", None)
        if self._wants_grad():
            depth = self._cl(self._depth_image)
            sg = 0.6 if sigma is None else float(sigma)
            rows = [[ops.FrameMapsFunction.apply(depth[b, s, ..., 0], K[b, 0], sg) for s in range(L)] for b in range(B)]
            vm = torch.stack([torch.stack([r[0] for r in row]) for row in rows])
            nm = torch.stack([torch.stack([r[1] for r in row]) for row in rows])
            self._vertex_map, self._normal_map = self._from_cl(vm), self._from_cl(nm)
            self._alpha_cache = (sg, torch.stack([torch.stack([r[2] for r in row]) for row in rows]).unsqueeze(-1))
            return
        # every (b, l) frame in ONE launch (gs_frame_maps_batch_f32)
        vm = torch.empty((B, L, H, W, 3), dtype=torch.float32, device=self.device)
        nm = torch.empty_like(vm)
        am = torch.empty((B, L, H, W, 1), dtype=torch.float32, device=self.device) if sigma is not None else None
        # the depth stack is read in place (channels-last (B, L, H, W, 1): its last dimension is dropped by a view; a
        # frame slice of a longer stack keeps its strides, nothing is copied)
        depth = self._depth_image[:, :, 0] if self.channels_first else self._depth_image[..., 0]
        ops.frame_maps_batch(depth, K.view(B, 4, 4), sigma,
                             out=(vm, nm, None if am is None else am[..., 0]))
        self._vertex_map, self._normal_map = self._from_cl(vm), self._from_cl(nm)
        if am is not None:
            self._alpha_cache = (float(sigma), am)

    def _alpha_map(self, sigma):
        """(B, L, H, W, 1) sample confidence exp(-|v|^2 / 2 sigma^2) of the local vertex map
        (slam/fusionutils.py:657), channels-last."""
        sigma = float(sigma)
        if self._alpha_cache is None or self._alpha_cache[0] != sigma or self._vertex_map is None:
            self._compute_local_maps(sigma)
        return self._alpha_cache[1]

    @property
    def vertex_map(self):
        if self._vertex_map is None:
            self._compute_local_maps()
        return self._vertex_map

    @property
    def normal_map(self):
        if self._normal_map is None:
            self._compute_local_maps()
        return self._normal_map

    def _compute_global_maps(self):
        from .. import ops
        B, L, H, W = self.shape
        vm, nm = self._cl(self.vertex_map), self._cl(self.normal_map)
        if self._poses is None:
            self._global_vertex_map, self._global_normal_map = self.vertex_map.clone(), self.normal_map.clone()
            return
        depth = self._cl(self._depth_image)
        poses = self._poses.contiguous().float()
        if torch.is_grad_enabled() and (vm.requires_grad or nm.requires_grad or poses.requires_grad):
            rows = [[ops.GlobalMapsFunction.apply(vm[b, s], nm[b, s], depth[b, s, ..., 0], poses[b, s]) for s in range(L)]
                    for b in range(B)]
            gv = torch.stack([torch.stack([r[0] for r in row]) for row in rows])
            gn = torch.stack([torch.stack([r[1] for r in row]) for row in rows])
        else:
            gv, gn = torch.empty_like(vm), torch.empty_like(nm)
            for b in range(B):
                for s in range(L):
                    ops.global_maps(vm[b, s], nm[b, s], depth[b, s, ..., 0], poses[b, s], out=(gv[b, s], gn[b, s]))
        self._global_vertex_map, self._global_normal_map = self._from_cl(gv), self._from_cl(gn)

    @property
    def global_vertex_map(self):
        if self._global_vertex_map is None:
            self._compute_global_maps()
        return self._global_vertex_map

    @property
    def global_normal_map(self):
        if self._global_normal_map is None:
            self._compute_global_maps()
        return self._global_normal_map

    # ------------------------------------------------------------------ setters (cache rules of
    # the reference: rgbdimages.py:399-463)
    @staticmethod
    def _assert_shape(value, shape):
        if not torch.is_tensor(value):
            raise TypeError("value must be torch.Tensor. Got {}".format(type(value)))
        if tuple(value.shape) != tuple(shape):
            raise ValueError("value must have shape {0}. Got {1} instead".format(tuple(shape), tuple(value.shape)))

    @rgb_image.setter
    def rgb_image(self, value):
        if value is not None:
            self._assert_shape(value, self._rgb_image_shape)
        self._rgb_image = value

    def _drop_local(self):
        self._vertex_map = self._normal_map = self._alpha_cache = None
        self._global_vertex_map = self._global_normal_map = None

    @depth_image.setter
    def depth_image(self, value):
        if value is not None:
            self._assert_shape(value, self._depth_image_shape)
        self._depth_image = value
        self._valid_depth_mask = None
        self._drop_local()

    @intrinsics.setter
    def intrinsics(self, value):
        if value is not None:
            self._assert_shape(value, self._intrinsics_shape)
        self._intrinsics = value
        self._drop_local()

    @poses.setter
    def poses(self, value):
        if value is not None:
            self._assert_shape(value, self._poses_shape)
        self._poses = value
        self._global_vertex_map = None
        self._global_normal_map = None

    # ------------------------------------------------------------------ copies / moves / layout
    def _mapped(self, fn):
        """A new container whose every tensor (inputs and cached maps alike) is fn(tensor); layout flag and shape
        bookkeeping carried over.  detach / clone / to are this with a different fn (structures/rgbdimages.py:465-525
        of the reference give the semantics: every cached map follows its inputs)."""
        other = object.__new__(type(self))
        other.__dict__.update(self.__dict__)
        for name, value in self.__dict__.items():
            if torch.is_tensor(value):
                other.__dict__[name] = fn(value)
        if self._alpha_cache is not None:   # (sigma, alpha map) rides with the vertex map it was computed with
            other._alpha_cache = (self._alpha_cache[0], fn(self._alpha_cache[1]))
        return other

    def detach(self):
        return self._mapped(torch.Tensor.detach)   # (shares storage with this object, off the autograd tape)

    def clone(self):
        return self._mapped(torch.clone)

    def to(self, device, copy: bool = False):
        device = torch.empty(0).to(device).device   # (resolves "cuda" to the current device index)
        if self.device == device and not copy:
            return self
        other = self._mapped(lambda t: t.to(device, copy=True))
        other.device = device
        return other

    def cpu(self):
        return self.to(torch.device("cpu"))

    def cuda(self):
        return self.to(torch.device("cuda"))

    def to_channels_last(self, copy: bool = False):
        if not (copy or self.channels_first):
            return self
        return self.clone().to_channels_last_()

    def to_channels_first(self, copy: bool = False):
        if not copy and self.channels_first:
            return self
        return self.clone().to_channels_first_()

    def _permute_all(self, ordering):
        for k in ("_rgb_image", "_depth_image", "_vertex_map", "_global_vertex_map", "_normal_map",
                  "_global_normal_map"):
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.permute(*ordering).contiguous())
        self._valid_depth_mask = None
        self._rgb_image_shape = tuple(self._rgb_image.shape)
        self._depth_image_shape = tuple(self._depth_image.shape)

    def to_channels_last_(self):
        if not self.channels_first:
            return self
        self._permute_all((0, 1, 3, 4, 2))
        self._channels_first = False
        return self

    def to_channels_first_(self):
        if self.channels_first:
            return self
        self._permute_all((0, 1, 4, 2, 3))
        self._channels_first = True
        return self
