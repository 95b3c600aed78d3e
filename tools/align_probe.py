import math, sys, torch
sys.path.insert(0, '/root/repo')
from multimodal_vqvae_compression_audio_tactile_amd import ops
dev = torch.device('cuda:0'); B = 256
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for T in (2999, 3000):
    x = torch.randn(B, 384, T, device=dev)
    w7 = ops.pack_conv1d(torch.randn(384, 384, 7, device=dev) / 50); w1 = ops.pack_conv1d(torch.randn(384, 384, 1, device=dev) / 20)
    b = torch.randn(384, device=dev); al = torch.rand(384, device=dev) + 0.5
    r = torch.randn(B, 384, T, device=dev)
    t7 = timeit(lambda: ops.conv1d(x, w7, 384, 7, bias=b, dil=3, pad=9, alpha_out=al))
    t1 = timeit(lambda: ops.conv1d(x, w1, 384, 1, bias=b, residual=r, alpha_dual=al))
    wt = ops.pack_conv_transpose1d(torch.randn(384, 192, 8, device=dev) / 30, 4)
    bt = torch.randn(192, device=dev); a2 = torch.rand(192, device=dev) + 0.5
    tt = timeit(lambda: ops.conv_transpose1d(x, wt, 192, 4, 2, bias=bt, alpha_dual=a2))
    print(f"T={T}: k7 {t7:.3f} ms  k1(+res,+dual) {t1:.3f} ms  convT {tt:.3f} ms")
