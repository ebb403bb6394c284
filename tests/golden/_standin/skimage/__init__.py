"""Test-only placeholder for scikit-image, which is not installed in this image.  The reference's libs/dataset.py does
`from skimage.io import imread` at import time (libs/dataset.py:11); the one method tests/golden/make_golden_graph.py
runs from that file (`AIC_dataset_inference_precomputed_features.__getitem__`, libs/dataset.py:283-312) never reads an
image.  Never imported by the product package."""
