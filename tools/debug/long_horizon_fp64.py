import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from tests.studies import long_horizon as lh
res64 = lh.shapes_batches(4, 210, precision=64)
tot9=tot4=tot=0
for sh,res in res64.items():
    r=res["rel"][199]; fb=lh.first_bad(res["rel"])
    print(f"  {sh:10s} substep 200: within 1e-9 {int((r<=1e-9).sum()):2d}/12  within 1e-4 {int((r<=1e-4).sum()):2d}/12   median {np.median(r):.1e}  max {r.max():.1e}   first beyond 1e-4: {sorted(int(x) for x in fb[fb>=0])} status {sorted(set(res['status'].tolist()))}", flush=True)
    tot9+=int((r<=1e-9).sum()); tot4+=int((r<=1e-4).sum()); tot+=len(r)
print(f"fp64 kernels: {tot9}/{tot} within 1e-9, {tot4}/{tot} within 1e-4 at substep 200")
