"""ctypes binding of libmggan_hip.so (the C ABI declared in include/mggan_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, a RuntimeError is raised.
"""
from .lib import lib, load, LIB_PATH, HipError  # noqa: F401
