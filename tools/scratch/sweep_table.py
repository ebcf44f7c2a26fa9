import sys, json, collections
t = collections.OrderedDict()
for l in sys.stdin:
    if not l.startswith("{"):
        if "amdgpu.ids" not in l: print(l.strip()[:200])
        continue
    d = json.loads(l)
    if "error" in d:
        print(d); continue
    t.setdefault((d["N"], d["bitrate"]), []).append("r%s/w%s:%.3f%s%s" % (d["rounds"], d["W"], d["ms"], "" if d["same"] else "!!", "" if d["fused"] else "(2k)"))
for k, v in t.items(): print(k, " ".join(v))
