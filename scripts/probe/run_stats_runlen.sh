for r in 5 10 30 75 150; do echo "RUN_A=$r"; D4W_FK_RUN_A=$r python - <<'PY'
import os,sys; sys.path.insert(0,os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch, numpy as np, das4whales_amd as dw
nx,ns=20000,120000
x=torch.randn((nx,ns),device="cuda"); y=torch.empty_like(x)
p=dw.dsp.get_fk_plan(nx,ns); p.set_mask(dw.dsp.fk_filter_design((nx,ns),[0,nx,1],2.0419046878814697,200.0))
p.apply_stats(x,out=y)
acc=np.zeros(5)
for _ in range(8):
    _,m,mx,ms=p.apply_stats(x,out=y,timed=True); acc+=np.array(ms)
print([round(v,3) for v in acc/8], round(acc.sum()/8,3))
PY
done
