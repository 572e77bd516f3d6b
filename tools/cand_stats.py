#!/usr/bin/env python3
"""Candidate statistics of the screening bounds on the REFERENCE's own z_e / codebook (build container only: imports
/root/reference).  For each operand format and bound: fraction of rows that keep >= 2 / >= 3 candidates.
Results: profiles/r02_vq_knockout.txt section 0."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import torch_port as tp
torch.manual_seed(0)
sys.path.insert(0,'/root/reference')
import os
os.environ['PYTHONDONTWRITEBYTECODE']='1'
from models.vqvae import VQVAE
m = VQVAE(128,32,2,512,64,0.25).eval()
x = torch.randn(256,3,32,32)
sd = m.state_dict()
with torch.no_grad():
    ze = tp.encode(sd, x, 2)
z = ze.permute(0,2,3,1).reshape(-1,64).double().numpy()
E = sd['vector_quantization.embedding.weight'].double().numpy()
N,K = z.shape[0], E.shape[0]
print('rows',N,'|z| mean',np.linalg.norm(z,axis=1).mean(),'|e| max',np.linalg.norm(E,axis=1).max())
def rnd(a, kind):
    t = torch.from_numpy(a).float()
    if kind=='bf16': return t.bfloat16().double().numpy()
    if kind=='fp16':
        s = 2.0**(14-np.floor(np.log2(np.abs(a).max())))
        return (t*s).half().double().numpy()/s
S = z@E.T - 0.5*(E*E).sum(1)[None]
M = S.max(1)
srt = np.sort(S,axis=1)
gap = srt[:,-1]-srt[:,-2]
print('gap mean',gap.mean(),'median',np.median(gap))
g = 2*65*2.0**-24
for kind,u in (('bf16',2.0**-8),('fp16',2.0**-11)):
    zh = rnd(z,kind); Eh = rnd(E,kind)
    Sh = zh@Eh.T - 0.5*(E*E).sum(1)[None]
    Mh = Sh.max(1)
    zn = np.linalg.norm(z,axis=1); znh=np.linalg.norm(zh,axis=1)
    dz = np.linalg.norm(z-zh,axis=1); 
    en = np.linalg.norm(E,axis=1); enh=np.linalg.norm(Eh,axis=1); de = np.linalg.norm(E-Eh,axis=1)
    variants = {
      'worst both': (2*u+u*u)*zn*en.max(),
      'actual e, worst z': u*zn*enh.max() + (1+u)*zn*de.max(),
      'actual both (max over codes)': dz*enh.max() + (znh+dz)*de.max(),
    }
    for name,err in variants.items():
        delta = 2*(err + g*zn*en.max())
        c = (Sh >= (Mh-delta)[:,None]).sum(1)
        # verify containment of true argmax
        ta = S.argmax(1); ok = Sh[np.arange(N),ta] >= Mh-delta
        print(f'{kind:5s} {name:30s} delta/gapmean {delta.mean()/gap.mean():.3f} rows>=2 {np.mean(c>=2):.4f} rows>=3 {np.mean(c>=3):.4f} meancand {c.mean():.3f} contain {ok.all()}')
    # per-code bound
    errk = dz[:,None]*enh[None] + (znh+dz)[:,None]*de[None] + g*zn[:,None]*en[None]
    kb = Sh.argmax(1)
    dl = errk + errk[np.arange(N),kb][:,None]
    c = (Sh >= Mh[:,None]-dl).sum(1)
    print(f'{kind:5s} per-code actual both: rows>=2 {np.mean(c>=2):.4f} rows>=3 {np.mean(c>=3):.4f} meancand {c.mean():.3f}')
