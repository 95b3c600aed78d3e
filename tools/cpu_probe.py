import os, time, torch, torch.nn.functional as F
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cgroup", e)
x = torch.randn(4, 384, 2999); w = torch.randn(384, 384, 7)
for n in (8, 16, 32, 64):
    torch.set_num_threads(n)
    F.conv1d(x, w, padding=3)
    t0 = time.perf_counter()
    for _ in range(5): F.conv1d(x, w, padding=3)
    dt = (time.perf_counter() - t0) / 5
    print(n, "threads: %.1f GFLOP/s" % (2 * 4 * 384 * 384 * 7 * 2999 / dt / 1e9))
