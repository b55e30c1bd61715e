import sys,re
v=[float(m.group(1)) for l in sys.stdin for m in [re.search(r"lr\s+([0-9.]+) ms",l)] if m]
v=v[1:]
print("rounds",len(v),"mean lr %.3f min %.3f max %.3f"%(sum(v)/len(v),min(v),max(v)))
