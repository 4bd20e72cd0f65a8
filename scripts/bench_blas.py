"""bench.py with the BLAS backend torch prefers switched (argv[1] = cublas | cublaslt), rest of argv passed on."""
import runpy
import sys

import torch

torch.backends.cuda.preferred_blas_library(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(__file__.replace("scripts/bench_blas.py", "bench.py"), run_name="__main__")
