import json, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from freud_amd.engine import SaeEngine
from oracle import sae_oracle as O
KEYS = ["encoder.weight", "encoder.bias", "W_dec", "b_dec"]
def rel(a,b):
    a,b=np.asarray(a,np.float64),np.asarray(b,np.float64); return np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30)
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for name in ["topk_adam_linear_d16","topk_adam_linear_d64"]:
    z=np.load(f"{root}/tests/golden/{name}.npz"); meta=json.loads(str(z["meta"]))
    d,n,k,B,T=meta["d"],meta["n"],meta["k"],meta["B"],meta["T"]
    eng=SaeEngine(variant="topk",d_model=d,n_dict=n,max_rows=B*T,optimizer="adam",k=k,auxk_alpha=meta["auxk_alpha"])
    eng.set_topk_options(meta["dead_feature_threshold"],T)
    eng.set_params({kk:z["init__"+kk] for kk in KEYS})
    xs=torch.tensor(z["x"]).cuda()
    eng.forward_backward(xs[0])
    m=eng.metrics(); print(name,"fvu",m[0],z["fvu"][0],"mse",m[2],z["mse"][0])
    flat=eng.debug_read(2,2*n*d+n+d); nd=n*d
    g={"encoder.weight":flat[:nd].reshape(n,d),"encoder.bias":flat[nd:nd+n],"W_dec":flat[nd+n:2*nd+n].reshape(n,d),"b_dec":flat[2*nd+n:]}
    for kk in KEYS: print("  ",kk,rel(g[kk],z["first__"+kk]))
    idx=eng.debug_read(3,B*T*k).reshape(B*T,k).astype(np.int64); ref=z["first__top_indices"].reshape(B*T,k)
    print("   idx same rows:",(np.sort(idx,1)==np.sort(ref,1)).all(1).mean())
    gw=g["encoder.weight"]; rw=z["first__encoder.weight"]
    rowerr=np.abs(gw-rw).max(1); print("   rows with err:", np.nonzero(rowerr>1e-3*np.abs(rw).max())[0][:40])
    dn=eng.debug_read(0,B*T*n).reshape(B*T,n); print("   dense nnz per row", (dn>0).sum(1)[:8])
