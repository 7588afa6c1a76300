"""device -> pinned host copy rate of this box: one stream, two streams at once, chunked -- what bounds the tail of stage II
   usage: python tools/micro/d2h_rate.py"""
import time, torch
dev = torch.device("cuda", 0)
GB = 1 << 30
src = torch.empty(2 * GB, dtype=torch.uint8, device=dev); src.fill_(7)
dst = torch.empty(2 * GB, dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(label, fn, nbytes, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-58s %6.1f GB/s" % (label, nbytes / best / 1e9))
def one():
    with torch.cuda.stream(s1): dst[:GB].copy_(src[:GB], non_blocking=True)
def two():
    with torch.cuda.stream(s1): dst[:GB].copy_(src[:GB], non_blocking=True)
    with torch.cuda.stream(s2): dst[GB:].copy_(src[GB:], non_blocking=True)
def chunks(k):
    def f():
        step = GB // k
        with torch.cuda.stream(s1):
            for i in range(k): dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)
    return f
def four():
    q = GB // 2
    ss = [s1, s2, torch.cuda.Stream(), torch.cuda.Stream()]
    for i, s in enumerate(ss):
        with torch.cuda.stream(s): dst[i * q:(i + 1) * q].copy_(src[i * q:(i + 1) * q], non_blocking=True)
run("1 GiB, one stream", one, GB)
run("2 x 1 GiB, two streams at once", two, 2 * GB)
run("4 x 0.5 GiB, four streams at once", four, 2 * GB)
run("1 GiB in 16 pieces, one stream", chunks(16), GB)
run("1 GiB in 256 pieces, one stream", chunks(256), GB)
