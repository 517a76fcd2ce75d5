"""Does a conditioned model with more spread targets give the triplet loss partly active hinges (so that its own gradient is
compared with the oracle's)?   python tools/probe/cond_spread.py [common_weight]"""
import os, sys, warnings
warnings.filterwarnings("ignore")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from parity_c2_report import conditioned_report
cw = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
margin = float(sys.argv[2]) if len(sys.argv) > 2 else None
out = conditioned_report(steps=300, common_weight=cw, margin=margin)
print("target_cosine", out["target_cosine"], "hinge_active", out["hinge_active"], "hinge compared:", "hinge" in out)
for key in ("smooth", "hinge"):
    if key in out:
        print(key, {k: (round(v["ratio"], 3), round(v["cos"], 3), round(v["ratio16"], 3), round(v["cos16"], 3)) for k, v in out[key]["gstats"].items()})
