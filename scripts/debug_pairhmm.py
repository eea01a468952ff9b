import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import pairhmm_oracle_lib as pol
from mgl_amd import pairhmm
h=pairhmm.MicrosoftPairHmm(0); h.load(); h.initialize(None)
rng=np.random.default_rng(1)
for R,H in [(4,4),(10,30),(16,30),(17,30),(20,30),(33,40),(40,8),(40,9),(40,15),(40,16),(40,17),(40,24),(40,25),(40,100)]:
    hap=bytes(rng.choice(list(b"ACGT"),size=H).astype(np.uint8)); s=int(rng.integers(0,max(1,H-R)))
    read=((hap[s:s+R]+hap)*3)[:R]
    q=bytes([30]*R); g=bytes([40]*R); c=bytes([10]*R)
    out=np.zeros(1); h.computeLikelihoods([pairhmm.ReadDataHolder(read,q,g,g,c)],[pairhmm.HaplotypeDataHolder(hap)],out)
    w,_=pol.log10_likelihood(hap,read,q,g,g,c)
    print(R,H,out[0],w,"OK" if abs(out[0]-w)<1e-5 else "BAD")
