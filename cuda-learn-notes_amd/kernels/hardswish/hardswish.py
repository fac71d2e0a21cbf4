"""hardswish bench -- same rows/tags as reference kernels/hardswish/hardswish.py. No GPU: only the torch rows run, on CPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _activation import main  # noqa: E402

if __name__ == "__main__":
    main("hardswish")
