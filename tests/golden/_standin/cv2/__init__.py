"""Test-only placeholder for OpenCV, which is not installed in this image.  The reference's utils.py does
`import cv2` at import time (utils.py:7) but none of the functions tests/golden/make_golden_pp.py runs
(compute_SCC_and_Clusters, splitting, remove_edges_single_direction, pruning) touches it.  Deliberately empty:
any attribute access fails.  Never imported by the product package."""
