"""Test-only placeholder for PyTorch Geometric, which is not installed in this image.  The reference's
inference.py imports `Data`, `Batch` and `to_networkx` at import time (inference.py:12,17); the one function
tests/golden/make_golden_pp.py runs from that file (`post_processing`, inference.py:70-169) uses none of them.
Never imported by the product package."""
