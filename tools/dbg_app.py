import sys, subprocess, numpy as np
sys.path.insert(0,'.')
from leansdr_amd import synth_dvbs
iq, ts = synth_dvbs.capture_u8(n_packets=1000, seed=3)
for bf in (4,):
    p = subprocess.run(['leansdr_amd/host/apps/leandvb_amd','--u8','-f','2400e3','--sr','2000e3','--cr','1/2','--buf-factor',str(bf),'-d'], input=iq.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print('bf',bf,'rc',p.returncode,'out bytes',len(p.stdout))
    e=p.stderr.decode(); print(e[:600]); print('....'); print(e[-2200:])
