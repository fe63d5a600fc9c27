import torch
dev = torch.device("cuda:0")
shapes = [(65536, 2304, 768), (65536, 768, 768), (65536, 3072, 768), (65536, 768, 3072), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
