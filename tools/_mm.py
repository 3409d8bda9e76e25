import torch
dev = torch.device("cuda:0")
for (M,N,K) in [(17877,2008,320),(4096,500,2080),(17877,2048,320)]:
    A = torch.randn(M,K,device=dev); B = torch.randn(N,K,device=dev)
    for _ in range(5): torch.mm(A,B.t())
    torch.cuda.synchronize()
A = torch.randn(17877,2048,device=dev); B = torch.randn(17877,320,device=dev)
for _ in range(5): torch.mm(A.t(),B)
torch.cuda.synchronize()
A = torch.randn(4096,500,device=dev); B = torch.randn(500,2080,device=dev)
for _ in range(5): torch.mm(A,B)
torch.cuda.synchronize()
