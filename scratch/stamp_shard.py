"""Unprofiled timeline of the small-shard step under graph replay (device clock stamps between the plan's segments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exp_shard import build

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng, xc, xv = build(B)
eng.enable_stamps()
for _ in range(12):
    eng.step(xc, xv)
print("plan:", eng.plan_summary())
for rep in range(2):
    for _ in range(3):
        eng.step(xc, xv)
    for n, t in eng.read_stamps():
        print(f"{t:9.2f} us  {n}")
    print()
