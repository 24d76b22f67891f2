import torch, time
n = 6_000_000_000
x = torch.empty(n, dtype=torch.int32, device="cuda"); y = torch.empty(n, dtype=torch.int32, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps
dt = t(lambda: x.fill_(7)); print("fill  %.2f ms  %.2f TB/s write" % (dt*1e3, n*4/dt/1e12))
dt = t(lambda: y.copy_(x)); print("copy  %.2f ms  %.2f TB/s (r+w)" % (dt*1e3, 2*n*4/dt/1e12))
dt = t(lambda: x.sum());    print("sum   %.2f ms  %.2f TB/s read" % (dt*1e3, n*4/dt/1e12))
