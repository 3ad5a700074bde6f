import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l)
        print("value %.3e ms/step %.4f conts/s %.3e"%(d["value"],d["ms_per_step"],d["continuations_per_s"]))
        for k,v in d["kernels"].items(): print("  %-14s avg %.4f ms  x%d  %.0f GB/s"%(k,v["avg_ms"],v["launches"],v["achieved_GBps"]))
        print(" ", d["roofline"])
