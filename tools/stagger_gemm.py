import os, sys, torch
sys.path.insert(0, ".")
from tools.ablate_gemm import run
