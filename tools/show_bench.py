import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.3f G el/s  ms/step %.4f  frac %.4f"%(d["value"]/1e9, d["ms_per_step"], d["roofline"]["frac"]))
print("check", d.get("check"))
ex=d.get("extras",d)
for k,v in ex.items():
    if any(s in k for s in ("fri_commit","merkle","stark_prove","proofs_per_s","callsite")) and not isinstance(v,dict):
        print(" ",k,v)
print("cpu_baseline", {k:v for k,v in d["cpu_baseline"].items() if k in ("value","unit","cores","kind","digest_matches_gpu")})
