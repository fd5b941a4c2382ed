#!/usr/bin/env python
"""bench.py against another build of the library: UNETK_AB_LIB=build_ab/libunetk_X.so python tools/ab_bench.py [bench.py flags]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unet_amd import _lib
if os.environ.get("UNETK_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["UNETK_AB_LIB"])
import bench
bench.main()
