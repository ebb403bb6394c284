"""Test-only placeholder for torchvision, which is not installed in this image.  The reference's libs/transforms.py
imports seven transform classes by name at import time (libs/transforms.py:7-9); nothing the golden generators run
constructs one.  Never imported by the product package."""
