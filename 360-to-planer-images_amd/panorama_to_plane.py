"""Legacy entry point kept drop-in: app/legacy/panorama_to_plane.py ("L") of the reference.

    interpolate_color(U, V, img, method='bilinear')   L:159-180  cv2.remap(img, U, V, interp, BORDER_REFLECT)
    panorama_to_plane(pano_array, U, V)               L:182-194  -> interpolate_color(U, V, pano_array)

Both run on the GPU through p2p_remap_maps_u8 (bit-exact restatement of cv2.remap's fixed-point
INTER_LINEAR).  The reference only ever calls the bilinear method (L:194); 'nearest' and
'bicubic' exist in its table (L:172-176) but are unreachable from its own callers and are not
implemented here -- they raise instead of silently substituting another filter.
"""
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _native  # type: ignore
else:
    from . import _native

_DEVICE = int(os.environ.get("P2P_DEVICE", "0"))


def interpolate_color(U, V, img, method="bilinear"):
    if method != "bilinear":
        raise NotImplementedError(
            "only method='bilinear' (cv2.INTER_LINEAR) is implemented on the GPU path; got %r" % (method,)
        )
    return _native.remap_maps(img, U, V, border=_native.BORDER_REFLECT, device=_DEVICE)


def panorama_to_plane(pano_array, U, V):
    return interpolate_color(U, V, pano_array)
