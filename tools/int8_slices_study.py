"""Is there an arithmetic cheaper than six bf16 piece products per fp32 product?  The one candidate the chip offers is the
int8 matrix instruction at twice the bf16 rate (v_mfma_i32_16x16x64_i8): operands cut into s signed 7-bit slices under a block exponent per row (A) / per column (B)
(the Ozaki scheme), slice products with i + j < keep kept, exact int32 accumulation.  This script measures the ACCURACY of that scheme against the bf16x6 products and fp32
on operands shaped like a hidden layer (K = 256; post-ReLU activations with a per-point scale, nn.Linear-initialised weights) and on heavy-tailed rows (gradient-like), error
relative to sum |a||b| (the scale fp32 rounding errors live on).  CPU only, numpy, seconds.  Result (recorded in DESIGN section 7): activations need 10 int8 products = 5.0
bf16-product equivalents for fp32-class accuracy, heavy-tailed rows 15 = 7.5 equivalents -- at best a sixth of the forward matrix time, before the row maxima and the integer
slicing (more VALU work than the bf16 split), and a loss for dX / dW.  Not a lever."""
# numerical check: int8-slice (Ozaki-style, row/column block exponents) product vs the bf16x6 product vs fp32, against fp64
import numpy as np
rng=np.random.default_rng(0)
M,K,N=512,256,256
def bf16(x):
    u=np.asarray(x,np.float32).view(np.uint32).astype(np.uint64)
    r=((u+0x7fff+((u>>16)&1))>>16<<16).astype(np.uint32)
    return r.view(np.float32)
def split3(x):
    h=bf16(x); r=(x-h).astype(np.float32); m=bf16(r); l=bf16((r-m).astype(np.float32)); return h,m,l
def x6(A,B):
    ah,am,al=split3(A); bh,bm,bl=split3(B)
    f=lambda a,b:(a.astype(np.float64)@b.astype(np.float64))
    return f(al,bh)+f(ah,bl)+f(am,bm)+f(am,bh)+f(ah,bm)+f(ah,bh)   # (fp32 accumulation not modelled: products only)
def slices(X,axis,s):
    mx=np.abs(X).max(axis=axis,keepdims=True); e=np.ceil(np.log2(np.maximum(mx,1e-300)))
    sc=2.0**(-e)           # |X*sc| <= 1
    r=X.astype(np.float64)*sc; out=[]; w=[]
    q=np.round(r*64); out.append(q); w.append(1/64.); r=r*64-q           # |q| <= 64, |r| <= 1/2
    for i in range(1,s):
        q=np.round(r*128); out.append(q); w.append(w[-1]/128.); r=r*128-q    # |q| <= 64
    return out,sc,w
def ozaki(A,B,s,keep):
    sa,sca,wa=slices(A,1,s); sb,scb,wb=slices(B,0,s)
    C=np.zeros((A.shape[0],B.shape[1])); n=0
    for i in range(s):
        for j in range(s):
            if i+j<keep:
                C+=(sa[i]@sb[j])*(wa[i]*wb[j]); n+=1
    return C/(sca*scb), n
# activations: post-ReLU of a gaussian pre-activation with per-point scale variation; weights: uniform(-1/16,1/16) like nn.Linear(256)
A=np.maximum(rng.standard_normal((M,K))*np.exp(rng.standard_normal((M,1))),0).astype(np.float32)
B=(rng.uniform(-1,1,(K,N))/16).astype(np.float32)
ref=A.astype(np.float64)@B.astype(np.float64)
den=(np.abs(A).astype(np.float64)@np.abs(B).astype(np.float64))      # sum |a||b| : the scale fp32 rounding errors live on
def err(C): 
    e=np.abs(C-ref)/den; return e.max(), np.sqrt((e**2).mean())
f32=(A@B).astype(np.float64)
print('fp32 numpy matmul        max %.2e rms %.2e (x 2^-24 = %.2e)'%(*err(f32),2**-24))
print('bf16x6 (products exact)  max %.2e rms %.2e'%err(x6(A,B)))
for s,keep in ((3,3),(4,4),(4,5),(5,5)):
    C,n=ozaki(A,B,s,keep)
    print('int8 slices s=%d, i+j<%d: %2d products = %.1f bf16-product equivalents   max %.2e rms %.2e'%(s,keep,n,n/2,*err(C)))
# gradients-like operand with a wide dynamic range inside a row (dY): heavy tails
A2=(rng.standard_normal((M,K))*np.exp(2*rng.standard_normal((M,K)))).astype(np.float32)
ref2=A2.astype(np.float64)@B.astype(np.float64); den2=np.abs(A2).astype(np.float64)@np.abs(B).astype(np.float64)
def err2(C): e=np.abs(C-ref2)/den2; return e.max(), np.sqrt((e**2).mean())
print('-- heavy-tailed rows (gradient-like)')
print('bf16x6                   max %.2e rms %.2e'%err2(x6(A2,B)))
for s,keep in ((4,4),(4,5),(5,5)):
    C,n=ozaki(A2,B,s,keep)
    print('int8 slices s=%d, i+j<%d: %2d products = %.1f equivalents   max %.2e rms %.2e'%(s,keep,n,n/2,*err2(C)))
