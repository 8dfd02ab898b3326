"""Prints the resident blocks per CU the runtime grants the split-bf16 tile kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommendation_amd.ops import lib
torch.zeros(1, device='cuda')
print('blocks per CU', lib().ctr_gemm_split_blocks_per_cu())
