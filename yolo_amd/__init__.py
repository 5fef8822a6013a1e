"""yolo_amd: the YOLOv3 inference hot path of shiboyang/YOLO on MI355X (gfx950)."""
