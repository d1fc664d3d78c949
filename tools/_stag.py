import os, subprocess, sys
# interleaved A/B in separate processes but same box; 2 rounds each
cfgs=[('0','0'),('1','2'),('1','4'),('2','2'),('2','4'),('1','1')]
for rnd in range(2):
    for m,n in cfgs:
        env=dict(os.environ, PNNP_STAGGER=m, PNNP_STAGGER_N=n)
        out=subprocess.run([sys.executable,'tools/layer_bench.py','--reps','5','--only','fwd','--layers','conv1_2,conv2_2,conv3_2,conv4_2,conv5_2,conv6_1,conv8_1,conv9_1'],env=env,capture_output=True,text=True).stdout
        tot=[l for l in out.splitlines() if l.startswith('total')]
        print('mode',m,'n',n,tot[0] if tot else out[-200:], flush=True)
