"""Functional probe of the explicit-matrix-as-parameter kernels (dsea_op_sddmm, dsea_op_update_vals) on a ragged matrix, both layouts."""
import torch, numpy as np, scipy.sparse as sp, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dominantsparseeigenad_amd.operators import CSROperator, TFIMOperator
dev=torch.device('cuda:0')
rng=np.random.RandomState(3)
n=1037
M=sp.random(n,n,density=0.02,random_state=rng,format='lil'); M[5,:]=0; M[:,5]=0
M=sp.csr_matrix(M); M=(M+M.T).tocsr(); M.sort_indices()
x=torch.randn(n,dtype=torch.float64,device=dev); v1=torch.randn_like(x); v2=torch.randn_like(x)
ref=torch.from_numpy(M@x.cpu().numpy())
rows=np.repeat(np.arange(n),np.diff(M.indptr))
for layout in ('sell','csr'):
  for c16 in ('auto',False):
    op=CSROperator.from_scipy(M,dev,layout=layout,col16=c16)
    y=op(x).cpu()
    print(layout,c16,getattr(op,'col16',None),'spmv rel',float((y-ref).abs().max()/ref.abs().max()))
    g=op.sddmm(v1,v2).cpu(); gr=v1.cpu()[rows]*v2.cpu()[M.indices]
    print('  sddmm equal',bool(torch.equal(g,gr)))
    gs=op.sddmm(v1,v2,symmetric=True).cpu(); grs=0.5*(v1.cpu()[rows]*v2.cpu()[M.indices]+v1.cpu()[M.indices]*v2.cpu()[rows])
    print('  sym maxdiff',float((gs-grs).abs().max()))
    # update
    newv=torch.randn(M.nnz,dtype=torch.float64,device=dev)
    op.vals.copy_(newv); 
    M2=M.copy(); M2.data=newv.cpu().numpy()
    y2=op(x).cpu(); ref2=torch.from_numpy(M2@x.cpu().numpy())
    print('  after in-place update rel',float((y2-ref2).abs().max()/ref2.abs().max()))
    op2=CSROperator(op.rowptr,op.colidx,newv.clone(),n,layout=layout,col16=c16)
    print('  update == rebuild bitwise',bool(torch.equal(op2(x),op(x))))
