"""bench.py's emission_regimes legs on their own (trained regime / label noise / sharp unrelated at the headline shape)."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import bench
out = bench.emission_regime_numbers(torch.device("cuda", 0))
print(json.dumps(out, indent=1))
