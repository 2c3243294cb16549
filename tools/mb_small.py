import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench_gemm as mb
mb.gemm(16, 768, 3072)
mb.gemm(16, 3072, 768)
mb.gemm(16, 768, 768)
mb.gemm(16, 30524, 768)
mb.gemm(16, 768, 30524)
mb.gemm(577, 768, 1024)
