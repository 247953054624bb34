# Test-infrastructure shim: the reference imports cv2 in structutils (unused on the hot path) and in the
# dataset loaders, where it is only asked to resize.  OpenCV is not installed in this image; resizing to
# the SAME size (what the golden generator asks for) is a copy, anything else is refused.
INTER_NEAREST, INTER_LINEAR = 0, 1


def resize(src, dsize, interpolation=INTER_LINEAR):
    w, h = dsize
    if (h, w) != tuple(src.shape[:2]):
        raise NotImplementedError("cv2 shim: only same-size resize (a copy) is available")
    return src.copy()
