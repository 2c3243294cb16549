"""Runs a couple of fixed shapes a few times (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench_gemm as mb  # noqa: E402

if __name__ == "__main__":
    mb.conv(2, 64, 64, 320, 320)
    mb.gemm(8192, 320, 1280)
    mb.gemm(4096, 4096, 4096)
