from .checkpoint import YOLOV3Checkpointer
from .load_darknet import load_darknet_weights_to_dict, save_darknet_weights

__all__ = ["YOLOV3Checkpointer", "load_darknet_weights_to_dict", "save_darknet_weights"]
