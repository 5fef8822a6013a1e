"""CPU oracle for the YOLOv3 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``yolo_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.

PARITY UNPINNED: the reference (shiboyang/YOLO) cannot be imported in the build
container (detectron2 / torchvision / fvcore / yacs / cv2 are absent) and its own
tests hold no assertion, golden tensor or known-answer vector for this path.  The
functions here restate the reference's algorithm line by line from its source text
(every function cites the file:line it follows) and the published behaviour of the
detectron2 / torchvision routines it calls; the conv / batch-norm / activation
arithmetic itself is ``torch.nn.functional`` on CPU, the same routines the reference
executes.
"""
