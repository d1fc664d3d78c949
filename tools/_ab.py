import os, subprocess, sys
# usage: python tools/_ab.py ENVVAR v1 v2 ... -- extra layer_bench args
name=sys.argv[1]; i=sys.argv.index('--'); vals=sys.argv[2:i]; extra=sys.argv[i+1:]
for rnd in range(2):
    for v in vals:
        env=dict(os.environ); env[name]=v
        out=subprocess.run([sys.executable,'tools/layer_bench.py']+extra,env=env,capture_output=True,text=True).stdout
        lines=[l for l in out.splitlines() if l.startswith(('total','conv'))]
        print(name,'=',v,' | '.join(l.split('H=')[0].strip() for l in lines), flush=True)
