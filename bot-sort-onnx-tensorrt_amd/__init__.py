"""MI355X-native BoT-SORT tracking hot path (gfx950 HIP kernels behind a C ABI).

The directory name carries hyphens (it mirrors the upstream project name), so import it with
`importlib.import_module("bot-sort-onnx-tensorrt_amd")` or through the alias package `botsort_amd`.

Public surface = the reference's (demo_bottrack_onnx_tflite.py): KalmanFilter, TrackState, BaseTrack,
STrack, BoTSORT, linear_assignment, iou_distance, bbox_ious, bbox_iou, joint_stracks, sub_stracks,
remove_duplicate_stracks -- plus BoTSORT.update_arrays / DeviceTracker for pre-extracted detections.
"""
from . import _lib
from ._lib import BtsError, Context, LIB_PATH
from .kalman_filter import KalmanFilter
from .matching import (bbox_iou, bbox_ious, cosine_similarity, fuse_cost_stage1, fuse_cost_stage3, iou_distance,
                       lapjv, linear_assignment, associate_parts, find_most_relevant_object, bbox_iou_by_box)
from .tracker import (Box, Body, Head, Face, Hand, attach_parts, BaseTrack, BoTSORT, DeviceTracker, tracker_memory_bytes, PinnedMot, STrack, TrackState, joint_stracks, remove_duplicate_stracks,
                      sub_stracks)
from . import multistream, synth

__all__ = [
    "BtsError", "Context", "LIB_PATH", "KalmanFilter", "TrackState", "BaseTrack", "STrack", "BoTSORT", "DeviceTracker", "tracker_memory_bytes",
    "linear_assignment", "lapjv", "iou_distance", "bbox_ious", "bbox_iou", "cosine_similarity", "fuse_cost_stage1",
    "fuse_cost_stage3", "joint_stracks", "sub_stracks", "remove_duplicate_stracks", "synth", "multistream",
    "PinnedMot", "Box", "Body", "Head", "Face", "Hand", "attach_parts", "associate_parts", "find_most_relevant_object", "bbox_iou_by_box",
]
