"""recommendation_amd — MI355X-native hot path of the CTR model zoo.

Python host code over libctrhip.so (hand-written HIP kernels behind a C ABI).  The model
classes that mirror the reference's import shape live under recommendation_amd/zoo
(`from DeepFM import DeepFM; DeepFM.DeepFM(args, cate_num, cont_num, cate_list)`); call
`recommendation_amd.enable_zoo_imports()` or put that directory on PYTHONPATH.
"""
import os
import sys

from .session import Session  # noqa: F401

ZOO_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'zoo')


def enable_zoo_imports():
    """Make `from DeepFM import DeepFM` (the reference's import shape) resolve to this build."""
    if ZOO_DIR not in sys.path:
        sys.path.insert(0, ZOO_DIR)
