from .yolo import YOLOV3, YoloV3Head

__all__ = ["YOLOV3", "YoloV3Head"]
