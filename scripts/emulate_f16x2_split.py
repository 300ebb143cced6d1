"""CPU emulation (numpy) of the f16x2 arithmetic of csrc/gemm_nt_h2w.hip on a weight gradient whose dy / x channels span
several decades: per-tensor scales (what the kernels do) vs per-row scales vs plain fp32, row- and column-wise error against
fp64.  Predicts the bound documented in DESIGN.md §2 and asserted on the GPU by tests/test_f16x2_gpu.py::test_per_channel_spread_*."""
import numpy as np
def scale_from_amax(a):
    e=np.floor(np.log2(a)); return 2.0**(14-e)
def split(x,s):
    xs=(x*s).astype(np.float32)
    hi=xs.astype(np.float16); lo=(xs-hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64),lo.astype(np.float64)
rng=np.random.default_rng(0)
M,C,K=64,64,4096
for spread in (0,-2,-4,-6,-8):
    chs=np.logspace(0,spread,M)
    dy=(rng.standard_normal((M,K))*chs[:,None]).astype(np.float32)
    x=(rng.standard_normal((C,K))*np.logspace(0,spread,C)[:,None]).astype(np.float32)
    ref=dy.astype(np.float64)@x.astype(np.float64).T
    sa=scale_from_amax(np.abs(dy).max()); sx=scale_from_amax(np.abs(x).max())
    ah,al=split(dy,sa); xh,xl=split(x,sx)
    got=(ah@xh.T+ah@xl.T+al@xh.T)/(sa*sx)
    f32=(dy@x.T).astype(np.float64)   # numpy fp32 accumulate (pairwise)
    # per-row scales
    sar=np.array([scale_from_amax(np.abs(r).max()) for r in dy]); sxr=np.array([scale_from_amax(np.abs(r).max()) for r in x])
    ah2,al2=split(dy,sar[:,None]); xh2,xl2=split(x,sxr[:,None])
    got2=(ah2@xh2.T+ah2@xl2.T+al2@xh2.T)/(sar[:,None]*sxr[None,:])
    rowerr=lambda g: (np.linalg.norm(g-ref,axis=1)/np.linalg.norm(ref,axis=1))
    elerr=lambda g: np.abs(g-ref)/np.abs(ref).max(axis=1,keepdims=True)
    print(f"spread 1e{spread}: tensor-scale relL2 {np.linalg.norm(got-ref)/np.linalg.norm(ref):.2e} worst row {rowerr(got).max():.2e} | row-scale worst row {rowerr(got2).max():.2e} | fp32 worst row {rowerr(f32).max():.2e}")
    # column-wise too
    colerr=lambda g: (np.linalg.norm(g-ref,axis=0)/np.linalg.norm(ref,axis=0))
    print(f"      worst column: tensor {colerr(got).max():.2e} rowscale {colerr(got2).max():.2e}")
