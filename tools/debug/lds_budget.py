"""Dev tool (GPU box): prints the LDS budget of the stepping kernel (KS_DEBUG line) for every shape and for the 14-shape mixed context."""
import os, sys
sys.path.insert(0, '.')
os.environ["KS_DEBUG"] = "1"
import torch
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim
names = scenarios.README_SHAPES if hasattr(scenarios, "README_SHAPES") else ["CubeS", "CubeM", "CubeB", "CylinderS", "CylinderM", "CylinderB", "Cube45S", "Cube45B", "Cone1S", "Cone1B", "Cone2S", "Cone2B", "Vase1S", "Vase1B", "Vase2S", "Vase2B"]
for nm in list(names) + ["mbox", "bcyl"]:
    try:
        s = KinovaSim(64, nm); s.close()
    except Exception as e:
        print(nm, "ERR", e)
s = KinovaSim(512, list(names)[:14]); s.close()
