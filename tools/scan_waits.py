import re,sys,subprocess,os
src=os.path.abspath(sys.argv[1])
out="/tmp/scan_"+os.path.basename(src)+".s"
subprocess.run(["hipcc","-O3","-std=c++17","--offload-arch=gfx950","-I","/root/repo/include","-S","--cuda-device-only",src,"-o",out],stderr=subprocess.DEVNULL,cwd="/tmp")
name=None; lines=open(out).read().split("\n")
res={}
for i,l in enumerate(lines):
    m=re.match(r"^(_ZN\S+):",l)
    if m: name=m.group(1); res[name]=[0,0]; continue
    if name is None: continue
    t=l.strip()
    if t.startswith("global_load_dword") or t.startswith("global_load_ubyte") or t.startswith("global_load_ushort"):
        res[name][0]+=1
        nxt=[x.strip() for x in lines[i+1:i+4]]
        if any(x.startswith("s_waitcnt vmcnt(0)") for x in nxt): res[name][1]+=1
for k,(a,b) in res.items():
    if b: print(f"{b:3d} of {a:3d} global loads wait at once: {k[:150]}")
