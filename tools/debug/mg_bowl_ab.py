import sys, time, os
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from kinovagrasping_amd import scenarios
from kinovagrasping_amd.sim import KinovaSim
n=4096
for sh,pose in (("BowlS","top"),("BowlB","normal"),("BowlB","rotated"),("RBowlS","top"),("BottleS","rotated"),("CubeS","normal")):
    sim=KinovaSim(n, [sh,"BottleS"] if sh=="CubeS" else sh, horizon=30, auto_reset=True)
    rng=np.random.RandomState(3)
    q=np.zeros((16,n)); q[12]=1; hq=np.zeros((4,n))
    for e in range(n):
        cmd = scenarios.start_coord_table(sh, pose)[rng.randint(0, 4000)] if scenarios.has_start_table(sh, pose) else scenarios.fallback_start(sh, pose, rng)
        q[9:12,e]=scenarios.reset_body_position(sh,cmd); q[0:3,e]=scenarios.hand_slide_offsets(pose,sh,"pose"); hq[:,e]=scenarios.hand_quat_for(pose)
    sim.reset(torch.as_tensor(q), torch.as_tensor(hq), object_id=np.zeros(n,dtype=np.int32) if sh=="CubeS" else None)
    a=torch.zeros(4,n,device="cuda"); a[1:]=0.5
    for _ in range(5): sim.step(a)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(30): sim.step(a)
    torch.cuda.synchronize(); dt=time.time()-t0
    st=sim.get_state()
    print(f"  {sh:8s} {pose:8s} {dt/30*1e3:6.3f} ms, status {sorted(set(st['status'].cpu().numpy().tolist()))}, contacts mean {float(st['ncon'].float().mean()):.1f} max {int(st['ncon'].max())}", flush=True)
    sim.close()
