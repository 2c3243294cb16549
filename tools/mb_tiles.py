"""Block-tile study on the conv / GEMM shapes that have enough tiles for a bigger block (run once per COMAT_FORCE_TILE
value: 64, 12864, 64128, 128)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench_gemm as mb  # noqa: E402

if __name__ == "__main__":
    print("tile:", os.environ.get("COMAT_FORCE_TILE", "64"))
    mb.conv(1, 512, 512, 128, 128)
    mb.conv(1, 256, 256, 256, 256)
    mb.conv(1, 128, 128, 512, 512)
    mb.conv(2, 64, 64, 320, 320)
    mb.conv(2, 32, 32, 640, 640)
    mb.gemm(8192, 2560, 320)
    mb.gemm(8192, 320, 1280)
    mb.gemm(8192, 320, 320)
