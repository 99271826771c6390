"""How many problems does sqp_select_kernel let run per round?  Reads run_buf is not exposed; instead: per-round
active counts (SCO_SQP_TRACE_ROUNDS) and ADMM iteration totals of two consecutive solves with different caps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from sco_py_amd import workloads as af, _lib, batch as sb
import torch
print("torch sees", torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
