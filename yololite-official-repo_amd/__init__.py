"""yololite-official-repo_amd -- MI355X-native (gfx950) YoloLite inference hot path.

backbone -> FPN -> heads -> anchor-free decode -> class-wise NMS as hand-written HIP kernels
behind a C ABI (include/yololite_hip.h, libyololite_hip.so); this package is the Python host
side mirroring the reference's call surface.  Importable as `yololite_amd` (see yololite_amd.py
at the repository root: the directory name contains a hyphen).
"""
from ._lib import YoloLiteHipError, load as load_library  # noqa: F401
from .model import (HipContext, YOLOLiteHIP, build_model_from_meta,  # noqa: F401
                    load_model_names_imgsize_from_ckpt)
from .postprocess import (_decode_batch_to_coco_dets, decode_anchorfree_like_train,  # noqa: F401
                          decode_preds_anchorfree, infer_main_postprocess, nms, predict_coco_dets, predict_main)
from .preprocess import letterbox_geometry, preprocess_batch  # noqa: F401
from .program import BACKBONES, Program, build_program  # noqa: F401
from .evalops import build_curves_from_coco, create_confusion_matrix  # noqa: F401
from .tracker import KalmanSortTracker, TrackerBank  # noqa: F401
from .serving import ServingPipeline  # noqa: F401

__all__ = ["YoloLiteHipError", "load_library", "HipContext", "YOLOLiteHIP", "build_model_from_meta",
           "load_model_names_imgsize_from_ckpt", "decode_preds_anchorfree", "_decode_batch_to_coco_dets",
           "decode_anchorfree_like_train", "infer_main_postprocess", "nms", "predict_main", "predict_coco_dets", "build_program", "Program", "BACKBONES",
           "preprocess_batch", "letterbox_geometry", "build_curves_from_coco", "create_confusion_matrix",
           "KalmanSortTracker", "TrackerBank", "ServingPipeline"]
