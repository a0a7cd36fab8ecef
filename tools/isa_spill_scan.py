#!/usr/bin/env python3
"""Development tool: per kernel of a device assembly listing (hipcc -S --cuda-device-only), the number of vector
instructions, of v_writelane/v_readlane (SGPRs parked in VGPR lanes) and of scratch instructions.  Usage: isa_spill_scan.py file.s"""
import sys,re,collections
def scan(path):
    lines=open(path).read().split('\n')
    cur=None;res={}
    for l in lines:
        if l.startswith('_ZN3crf') and ': ; @' in l:
            cur=l.split(':')[0]; res[cur]=collections.Counter(); continue
        if cur is None: continue
        t=l.strip().split()
        if not t: continue
        if t[0]=='.end_amdhsa_kernel' or t[0].startswith('.section'): cur=None; continue
        if t[0].startswith(('v_','s_','ds_','buffer_','global_','scratch_')):
            res[cur][t[0]]+=1
    for k,c in res.items():
        valu=sum(v for o,v in c.items() if o.startswith('v_'))
        lanes=c['v_writelane_b32']+c['v_readlane_b32']
        scr=sum(v for o,v in c.items() if o.startswith('scratch_'))
        if lanes>40 or scr>0:
            print(f"{k[:95]:95s} valu={valu:6d} lane_spill={lanes:5d} scratch={scr}")
scan(sys.argv[1])
