import torch
torch.manual_seed(0)
for (m,k,n) in [(6102,64,48),(6102,16,128),(100000,96,9),(6102,128,48)]:
    a=torch.randn(m,k); b=torch.randn(k,n)
    ref=(a.double()@b.double())
    cpu=(a@b).double()
    gpu=(a.cuda()@b.cuda()).cpu().double()
    lin=torch.nn.functional.linear(a.cuda(), b.t().contiguous().cuda()).cpu().double()
    r=lambda x: float((x-ref).norm()/ref.norm())
    print(m,k,n,'cpu',r(cpu),'gpu mm',r(gpu),'gpu linear',r(lin))
print(torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision())
import os; print({k:v for k,v in os.environ.items() if 'BLAS' in k or 'TF32' in k or 'TUNABLE' in k})
