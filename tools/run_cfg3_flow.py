"""cfg3 end to end through the radix tree (bench.py extras.cfg3_flow), both trees, on its own."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mini-sglang_amd"))
from harness import workloads as w
for native in (True, False, True):
    print(json.dumps(w.cfg3_flow("cuda:0", native_tree=native)), flush=True)
