import signal
signal.signal(signal.SIGPIPE, signal.SIG_DFL)
import sys, json
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
r=d["roofline"]
print(round(d["value"]), round(d["ms_per_step"],2), round(r["achieved"]), round(r["frac"],4), r.get("clock_ghz"), r.get("frac_at_clock"), round(r["gemm_ms_per_step"],2))
for k,v in d["gemm_sites"].items(): print("  ", k, round(v["tflops"]), round(v["us"],1), v["launches_per_step"])
