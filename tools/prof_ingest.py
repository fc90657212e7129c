import sys, time, os, cProfile, pstats
sys.path.insert(0,'.'); sys.path.insert(0,'./tests')
import numpy as np, bamio
from nanocaller_amd.synth_device import _read_layout
L=3_000_000
rng=np.random.default_rng(1)
ref=rng.integers(0,4,L+1).astype(np.uint8)
s,e=_read_layout(rng,L,30.0,"ont")
lut=np.frombuffer(b"AGTCNNNN",np.uint8)
recs=[]
for r in range(len(s)):
    c=ref[s[r]:e[r]].copy()
    m=rng.random(len(c))
    c[m<0.04]=(c[m<0.04]+1)%4
    c[(m>=0.04)&(m<0.08)]=4
    recs.append(dict(name="r%d"%r,flag=16 if r&1 else 0,pos0=int(s[r])-1,cigar=[("M",int(e[r]-s[r]))],seq=lut[c].tobytes().decode(),tags={}))
bam,fa="/tmp/p.bam","/tmp/p.fa"
bamio.write_bam(bam,"c1",L,recs,level=1); bamio.write_fasta(fa,"c1",lut[ref[1:]].tobytes().decode())
print("written",os.path.getsize(bam)/1e6)
from nanocaller_amd import generate_SNP_pileups as gsp
from nanocaller_amd.wire import build_wire_from_world
def run():
    gsp.release_contig()
    w=gsp._resolve(bam,"c1",fa)
    return build_wire_from_world(w, pin=False)
run()
for _ in range(2):
    t=time.perf_counter(); run(); print("ingest %.1f ms"%((time.perf_counter()-t)*1e3))
pr=cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
