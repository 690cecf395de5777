import torch, sys
sys.path.insert(0, ".")
from tools.gemm_big_bench import timeit
M=44400
x=torch.randn(M,256,device="cuda").bfloat16()
o768=torch.empty(M,768,device="cuda",dtype=torch.bfloat16)
o1024=torch.empty(M,1024,device="cuda",dtype=torch.bfloat16)
def f768():
    o768.view(M,3,256).copy_(x.unsqueeze(1).expand(M,3,256))
def f1024():
    o1024.view(M,4,256).copy_(x.unsqueeze(1).expand(M,4,256))
big=torch.empty(64<<20,device="cuda",dtype=torch.bfloat16)
def fill(): big.fill_(1.0)
src=torch.empty(64<<20,device="cuda",dtype=torch.bfloat16)
def cp(): big.copy_(src)
for n,f,by in (("bcast 22.7->68MB",f768,M*256*2+M*768*2),("bcast 22.7->91MB",f1024,M*256*2+M*1024*2),("fill 128MB",fill,128<<20),("copy 128+128MB",cp,256<<20)):
    tw,tc=timeit(f,False,20),timeit(f,True,10)
    print(f"{n:20s} {tw:7.1f}/{tc:7.1f} us  {by/tw/1e3:6.0f}/{by/tc/1e3:6.0f} GB/s")
