"""A/B: times the attention forward (and any other entry) of several library builds on the same box.
usage: python tools/ab_bench.py lib1.so lib2.so ...   (each in a fresh subprocess; headline shapes)"""
import os, subprocess, sys, json
CODE = r'''
import sys, os, torch, json
sys.path.insert(0, os.environ["ROOT"])
from diverse_channel_vit_amd import hip
hip.load()
B,N,H=64,1569,6; D=384; M=B*N
torch.manual_seed(0)
qkv=torch.randn(B,N,3*D,device="cuda").to(torch.bfloat16); o=torch.empty(B,N,D,dtype=torch.bfloat16,device="cuda"); lse=torch.empty(B,H,N,device="cuda")
dO=torch.randn(B,N,D,device="cuda").to(torch.bfloat16); dqkv=torch.empty_like(qkv); delta=torch.empty(2,B,H,N,device="cuda")
A=torch.randn(M,D,device="cuda").to(torch.bfloat16); W=torch.randn(3*D,D,device="cuda").to(torch.bfloat16)*0.05; bias=torch.zeros(3*D,device="cuda"); out=torch.empty(M,3*D,dtype=torch.bfloat16,device="cuda")
W1=torch.randn(4*D,D,device="cuda").to(torch.bfloat16)*0.05; b1=torch.zeros(4*D,device="cuda"); z=torch.empty(M,4*D,dtype=torch.bfloat16,device="cuda"); hh=torch.empty_like(z)
dW=torch.zeros(3*D,D,device="cuda"); db=torch.zeros(3*D,device="cuda")
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
r={}
r["attn_fwd_us"]=t(lambda: hip.attn_fwd(qkv,o,lse,B,N,H,64,0.125))
r["attn_bwd_us"]=t(lambda: hip.attn_bwd(qkv,o,dO,lse,delta,dqkv,B,N,H,64,0.125))
r["gemm_qkv_us"]=t(lambda: hip.gemm_nt(A,W,hip.EPI_BIAS_BF16,out,bias=bias))
r["gemm_fc1_us"]=t(lambda: hip.gemm_nt(A,W1,hip.EPI_BIAS_GELU_BF16,z,bias=b1,out2=hh))
r["gemm_tn_qkv_us"]=t(lambda: hip.gemm_tn_acc(out,A,dW,db))
dWp=torch.zeros(D,D,device="cuda"); dW1=torch.zeros(4*D,D,device="cuda"); dW2=torch.zeros(D,4*D,device="cuda"); db1=torch.zeros(4*D,device="cuda"); dbp=torch.zeros(D,device="cuda")
r["gemm_tn_proj_us"]=t(lambda: hip.gemm_tn_acc(A,A,dWp,dbp))
r["gemm_tn_fc1_us"]=t(lambda: hip.gemm_tn_acc(z,A,dW1,db1))
r["gemm_tn_fc2_us"]=t(lambda: hip.gemm_tn_acc(A,z,dW2,dbp))
Wp=torch.randn(D,D,device="cuda").to(torch.bfloat16)*0.05; bp=torch.zeros(D,device="cuda"); xr=torch.randn(M,D,device="cuda"); xo=torch.empty_like(xr)
r["gemm_proj_resid_us"]=t(lambda: hip.gemm_nt(A,Wp,hip.EPI_BIAS_RESID_F32,xo,bias=bp,aux=xr))
W2=torch.randn(D,4*D,device="cuda").to(torch.bfloat16)*0.05
r["gemm_fc2_resid_us"]=t(lambda: hip.gemm_nt(hh,W2,hip.EPI_BIAS_RESID_F32,xo,bias=bp,aux=xr))
du=torch.empty(M,D,dtype=torch.bfloat16,device="cuda")
r["gemm_du2_plain_us"]=t(lambda: hip.gemm_nt(z,W2,hip.EPI_PLAIN_BF16,du))
r["gemm_dO_plain_us"]=t(lambda: hip.gemm_nt(A,Wp,hip.EPI_PLAIN_BF16,du))
r["gemm_dz_gelubwd_us"]=t(lambda: hip.gemm_nt(A,W1,hip.EPI_GELU_BWD_BF16,hh,aux=z))
xs=[torch.randn(M,D,device="cuda") for _ in range(3)]; g=torch.ones(D,device="cuda"); bb=torch.zeros(D,device="cuda"); u=torch.empty(M,D,dtype=torch.bfloat16,device="cuda"); mean=torch.empty(M,device="cuda"); rstd=torch.empty(M,device="cuda")
dus=[torch.randn(M,D,device="cuda").to(torch.bfloat16) for _ in range(3)]; dxi=torch.randn(M,D,device="cuda"); dxo=torch.empty(M,D,device="cuda"); dxb=torch.empty(M,D,dtype=torch.bfloat16,device="cuda"); dg=torch.zeros(D,device="cuda"); db2=torch.zeros(D,device="cuda")
it=[0]
def lnf():
    it[0]+=1; hip.ln_fwd(xs[it[0]%3],g,bb,u,mean,rstd,M,D,1e-6)
def lnb():
    it[0]+=1; hip.ln_bwd(dus[it[0]%3],xs[it[0]%3],mean,rstd,g,dxi,dxo,dxb,dg,db2,M,D)
r["ln_fwd_us"]=t(lnf); r["ln_bwd_us"]=t(lnb)
print(json.dumps(r))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, DCV_LIB=os.path.join(root, lib), ROOT=root)
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(rnd, lib, line[-1] if line else out.stderr[-400:], flush=True)
