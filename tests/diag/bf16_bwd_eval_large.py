"""Diagnostic: tests/test_gpu_bf16.py::test_bf16_storage_backward at a shape / mode outside its parameter list.
usage: python tests/diag/bf16_bwd_eval_large.py [train|eval] [n h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import test_gpu_bf16 as t
mode = sys.argv[1] if len(sys.argv) > 1 else "eval"
shape = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 256, 320)
try:
    t.test_bf16_storage_backward(shape, mode, sys.argv[5] if len(sys.argv) > 5 else "bf16")
except AssertionError as e:
    print("assert:", str(e)[:300])
