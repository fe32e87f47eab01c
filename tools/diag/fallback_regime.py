"""The headline shape with emissions that contradict the targets (bench.py's fallback_regime leg), standalone."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(json.dumps(bench.fallback_regime_numbers(torch.device("cuda", 0))))
