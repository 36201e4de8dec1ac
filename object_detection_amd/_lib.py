"""ctypes binding of libodbox.so (include/odbox.h).  There is no CPU fallback: if the HIP
library is missing or no GPU is present every compute entry raises."""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libodbox.so')
# A/B tuning only (tools/ab_*.sh): another build of the same library, e.g. made with ODBOX_EXTRA_HIPCC_FLAGS
LIB_PATH = os.environ.get('ODBOX_LIBRARY', LIB_PATH)

_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/odbox.h one to one
SIGNATURES = {
    'odbox_version': (_i, []),
    'odbox_last_error': (ctypes.c_char_p, []),
    'odbox_set_output_policy': (_i, [_i]),
    'odbox_match_encode_plan': (_i, [_i, _i, _i, _i, _i, _i]),
    'odbox_match_encode_last_plan': (_i, []),
    'odbox_anchor_tile': (_i, [_vp, _i, _i, _i, _f, _vp, _vp]),
    'odbox_match_encode': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'odbox_match_loss_workspace': (_sz, [_i, _i]),
    'odbox_match_loss': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _f, _f, _f, _vp, _vp, _vp,
                              _vp, _sz, _vp]),
    'odbox_match_loss_noalloc': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _f, _f, _f, _vp, _vp, _vp,
                                      _vp]),
    'odbox_match_loss_backward': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _f, _f, _f, _vp, _vp,
                                       _vp, _vp, _vp, _vp]),
    'odbox_match_loss_f16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _f, _f, _f, _vp, _vp, _vp,
                                  _vp, _sz, _vp]),
    'odbox_match_loss_backward_f16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _f, _f, _f, _vp,
                                           _vp, _vp, _vp, _vp, _vp]),
    'odbox_match_loss_bf16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _f, _f, _f, _vp, _vp, _vp,
                                   _vp, _sz, _vp]),
    'odbox_match_loss_backward_bf16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _f, _f, _f, _vp,
                                            _vp, _vp, _vp, _vp, _vp]),
    'odbox_focal_loss_workspace': (_sz, [_sz]),
    'odbox_focal_loss': (_i, [_vp, _vp, _sz, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    'odbox_set_nms_tier': (_i, [_i]),
    'odbox_set_nms_fused_tail': (_i, [_i]),
    'odbox_focal_loss_rows': (_i, [_vp, _vp, _sz, _i, _f, _f, _i, _vp, _vp, _vp]),
    'odbox_bbox_overlap': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'odbox_bbox_transform': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    'odbox_regress_boxes': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    'odbox_clip_boxes': (_i, [_vp, _sz, _f, _f, _vp, _vp]),
    'odbox_scale_boxes': (_i, [_vp, _sz, _f, _f, _i, _vp, _vp]),
    'odbox_decode_nms_workspace': (_sz, [_i, _i, _i, _i, _i]),
    'odbox_decode_nms': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _f, _i, _i,
                              _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'odbox_decode_nms_mixed': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _f, _i, _f, _f, _i, _i,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'odbox_coco_records': (_i, [_vp, _sz, _vp, _vp, _vp, _vp]),
    'odbox_coco_iou': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    'odbox_coco_match_workspace': (_sz, [_i, _i]),
    'odbox_coco_match': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i,
                              _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'odbox_pack_records': (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    'odbox_unpack_records': (_i, [_vp, _sz, _vp, _vp, _vp, _vp]),
    'odbox_restore_boxes': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'odbox_gt_prepare_xywh': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _f, _i, _i, _vp, _vp, _vp]),
    'odbox_gt_prepare_poly': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _f, _i, _i, _vp, _vp, _vp]),
    'odbox_legacy_iou': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'odbox_legacy_match': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, ctypes.c_double, _vp, _vp, _vp, _vp, _vp]),
}


class OdboxError(RuntimeError):
    pass


def load():
    """Load libodbox.so and declare every prototype.  Needs no GPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OdboxError(
                'libodbox.so is not built (%s); run `python -m object_detection_amd.build`. '
                'There is no CPU fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise OdboxError('no ROCm GPU visible: odbox kernels are HIP-only, there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def check(rc, what):
    if rc != 0:
        msg = load().odbox_last_error()
        raise OdboxError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_ws_lock = threading.Lock()
_ws_cache = {}


def workspace(kind, dev, need):
    """Scratch bytes for one library call: one buffer per (kind, device, stream, host thread), grown
    on demand and reused by later calls.  Two host threads never share a buffer (calls that overlap
    must not share a workspace, include/odbox.h), and calls of one thread on one stream are ordered
    by the stream.  `release_workspaces()` drops them all."""
    key = (kind, dev, torch.cuda.current_stream().cuda_stream, threading.get_ident())
    with _ws_lock:
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < need:
            ws = _ws_cache[key] = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev)
        return ws


def release_workspaces():
    with _ws_lock:
        _ws_cache.clear()


def as_dev(x, dtype, device):
    """torch tensor / numpy array / nested list -> contiguous 16-byte-aligned device tensor."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    x = x.to(device=device, dtype=dtype, non_blocking=True).contiguous()
    if x.data_ptr() % 16:
        x = x.clone()
    return x
