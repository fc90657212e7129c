import csv, glob, os, sys
d=sys.argv[1]
kt=sorted(glob.glob(os.path.join(d,"**","*kernel_trace.csv"),recursive=True))[0]
mt=sorted(glob.glob(os.path.join(d,"**","*memory_copy_trace.csv"),recursive=True))[0]
K=list(csv.DictReader(open(kt))); M=list(csv.DictReader(open(mt)))
K.sort(key=lambda r:int(r["Start_Timestamp"]))
ex=[i for i,r in enumerate(K) if "k_wire_expand" in r["Kernel_Name"]]
a,b=ex[-20],ex[-18]
t0=int(K[a]["Start_Timestamp"]); t1=int(K[b]["Start_Timestamp"])
ev=[]
for r in K[a:b+1]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:28]
    if any(x in n for x in ("expand","k5_trunk","k_sites","k_snp_heads","featurize","k_copy4")):
        ev.append((int(r["Start_Timestamp"])-t0, int(r["End_Timestamp"])-int(r["Start_Timestamp"]), "K "+n))
print(M[0].keys())
for r in M:
    s=int(r["Start_Timestamp"])
    if t0-2_000_000<=s<=t1:
        ev.append((s-t0, int(r["End_Timestamp"])-s, "M %s %s bytes" % (r.get("Direction",""), r.get("Bytes", r.get("Size","")))))
for e in sorted(ev): print("%+10.1f us %9.1f us  %s" % (e[0]/1e3, e[1]/1e3, e[2]))
