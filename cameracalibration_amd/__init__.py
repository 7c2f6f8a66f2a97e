"""MI355X-native dense stereo disparity hot path (libmdisp.so) with a host-side mirror of the
reference's function-level API.  See include/mdisp.h for the C ABI."""
from . import _lib  # noqa: F401

__all__ = ["api", "_lib", "build"]
