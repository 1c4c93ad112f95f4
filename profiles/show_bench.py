import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d["roofline"]
print(d['value'], r["kernel"], r["frac"], r["all_gemm_kernels"])
for k,v in r["other_kernels"].items():
    if not k.startswith(("gemm","conv3x3")): print(k, v)
