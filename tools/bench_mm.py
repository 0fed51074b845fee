import torch, time
dev = torch.device('cuda:0')
def bench(M, N, K, n=20):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.mm(a, b.t())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): torch.mm(a, b.t())
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / n
    print('mm M=%d N=%d K=%d: %.3f ms  %.0f TF' % (M, N, K, t, 2.0 * M * N * K / t / 1e9))
for (M, N, K) in [(524288, 256, 1152), (131072, 512, 1152), (32768, 256, 4608), (524288, 128, 1152), (524288, 128, 2304), (8192, 8192, 8192), (32768, 1024, 1152),
                  (1152, 256, 524288), (1152, 512, 131072),
                  # the small-map layers of the generator (round 5): 16^2 1024->1024, its [gamma | beta] data gradient, 8^2, 32^2, the fused 16^2
                  (2048, 1024, 9216), (2048, 128, 18432), (512, 1024, 9216), (8192, 512, 9216), (8192, 512, 4608), (2048, 2048, 1152)]:
    bench(M, N, K)
