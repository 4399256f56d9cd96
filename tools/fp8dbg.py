import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import sae_oracle as O
O.MATMUL_MODE="fp32"; O.NORM_MODE="float64"
from freud_amd.engine import SaeEngine
d,n,M=1280,2560,600
g = torch.Generator().manual_seed(d+n+M)
W = torch.randn(d, n, generator=g) / d ** 0.5
b = 0.01 * torch.randn(n, generator=g)
x = ((torch.relu(torch.randn(M, 64, generator=g)) * 0.1) @ torch.randn(64, d, generator=g)).to(torch.bfloat16)
x.view(-1)[torch.randint(0, x.numel(), (50,), generator=g)] = -1.0
eng = SaeEngine(variant="l1", d_model=d, n_dict=n, max_rows=M, optimizer="adam", recon_alpha=1e4, precision="fp8")
eng.set_params({"decoder.weight": W.numpy(), "encoder_bias": b.numpy()})
eng.forward_backward(x.cuda())
Wn = O.normalize_columns(W)
f8 = O.l1_forward(x.float(), Wn, b, 1e4, True, "fp8")
c8 = eng.debug_read(9, M*n).reshape(M,n); ref8=f8["c8"].numpy()
c = eng.debug_read(0, M*n).reshape(M,n); cref=f8["c"].numpy()
diff = c8!=ref8
print("frac", diff.mean(), "s_c", f8["s_c"], "s_x", f8["s_x"])
ulp = 2.0 ** -7 * (np.abs(ref8) + 0.06 * f8["s_c"])
bound = 0.15*np.abs(ref8)+2.0**-8+1.2*ulp
viol = np.abs(c8-ref8) > bound
print("violations", viol.sum())
idx = np.argwhere(viol)[:20]
for r,cidx in idx:
    print(r,cidx,"c8",c8[r,cidx],"ref8",ref8[r,cidx],"c(bf16)",c[r,cidx],"cref",cref[r,cidx],"b",b[cidx].item(), "c*sc", cref[r,cidx]*f8["s_c"])
