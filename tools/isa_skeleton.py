import re,subprocess,sys
path=sys.argv[1]; want=sys.argv[2].replace(" ","")
cur=None; lines=[]; names={}
with open(path) as f:
    for ln in f:
        m=re.match(r'^(_Z\w+):',ln)
        if m:
            n=m.group(1)
            if n not in names:
                names[n]=subprocess.run(["c++filt",n],capture_output=True,text=True).stdout.strip()
            cur=names[n].replace(" ","")
            on = cur.startswith("void"+want+"(") or cur.startswith(want+"(")
            continue
        if cur and on:
            lines.append(ln.rstrip())
            if ln.strip().startswith("s_endpgm"): break
cnt=0
for l in lines:
    t=l.strip()
    if re.match(r'^(global_load|global_store|scratch_|s_waitcnt|s_cbranch|s_barrier|ds_read|ds_write|\.LBB)',t):
        print("%5d valu | %s"%(cnt,t[:100])); cnt=0
    elif t.startswith('v_'): cnt+=1
