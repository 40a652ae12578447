import torch, torch.nn.functional as F, time
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
X = torch.randn(8*256*256, 128, device='cuda'); W = torch.randn(128, 128, device='cuda'); b = torch.randn(128, device='cuda')
W3 = torch.randn(9, 128, device='cuda'); b3 = torch.randn(9, device='cuda')
G = torch.randn(8*256*256, 128, device='cuda')
for lib in ("default", "cublas", "cublaslt"):
    if lib != "default": torch.backends.cuda.preferred_blas_library(lib)
    print(lib)
    print('  linear 128->128 fwd  ', t(lambda: F.linear(X, W, b)), 'ms', 2*X.shape[0]*128*128/1e9, 'GFLOP')
    print('  matmul X@W.T (nobias)', t(lambda: X @ W.t()))
    print('  addmm                ', t(lambda: torch.addmm(b, X, W.t())))
    print('  linear 128->9 fwd    ', t(lambda: F.linear(X, W3, b3)))
    print('  dX = G @ W           ', t(lambda: G @ W))
    print('  dW = G.T @ X         ', t(lambda: G.t() @ X))
    print('  X4d (8,256,256,128) linear', t(lambda: F.linear(X.view(8,256,256,128), W, b)))
