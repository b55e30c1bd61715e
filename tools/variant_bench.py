#!/usr/bin/env python3
"""Tuning helper: run bench.py against a variant build of libbpgpu.so.
  python tools/variant_bench.py build/libbpgpu_x.so [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bulletproofs_amd import ffi  # noqa: E402

ffi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
