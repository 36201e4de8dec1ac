"""object_detection_amd -- MI355X-native box processing (anchors, IoU matching + target
encoding, decode + per-class NMS) behind the Python signatures of bioothod/object_detection's
`anchors` and `bndbox` modules.  HIP kernels via a ctypes C-ABI (include/odbox.h)."""
from . import anchors, bndbox, cocoeval, config, detection, dist, metric  # noqa: F401
from .anchors import AnchorGenerator, anchor_targets_bbox, generate_anchors  # noqa: F401
from .bndbox import decode_and_nms, nms  # noqa: F401
from .config import AnchorsConfig  # noqa: F401
