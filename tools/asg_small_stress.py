import sys, time, threading
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib = _lib.load(); dev = _lib.require_gpu()
rng = np.random.RandomState(123)
def inst(k):
    n = int(rng.randint(2, 257))
    fam = k % 8
    if fam == 0: M = rng.rand(n, n) * 10
    elif fam == 1:
        d = int(rng.choice([1, 2, 3, 8, 50])); x = rng.randn(n, d); y = rng.randn(n, d) + rng.rand()
        M = ((x[:, None, :] - y[None]) ** 2).sum(-1)
    elif fam == 2: M = rng.randint(0, int(rng.choice([2, 5, 50])), size=(n, n)).astype(float)
    elif fam == 3: M = np.sort(rng.rand(n))[:, None] * np.sort(rng.rand(n))[None, :] * -1.0     # Monge-like, many near ties
    elif fam == 4: M = rng.rand(n, n) ** 8 * 1e4
    elif fam == 5: M = rng.randn(n, n) * 1e3 + 1e5
    elif fam == 6: M = np.abs(np.subtract.outer(np.sort(rng.rand(n)), np.sort(rng.rand(n))))   # 1-D chains
    else: M = np.repeat(rng.rand(n, (n + 1) // 2), 2, axis=1)[:, :n]
    return np.ascontiguousarray(M, dtype=np.float32)
bad = 0; small = 0; t0 = time.time()
N = 600
for k in range(N):
    Mnp = inst(k)
    M = torch.from_numpy(Mnp).to(dev)
    perm, info = ot.assign_exact(M, return_info=True)
    p = perm.cpu().numpy().astype(np.int64)
    ref = oracle.exact_perm(Mnp)
    c, cr = oracle.assignment_cost(Mnp, p), oracle.assignment_cost(Mnp, ref)
    sc = max(1.0, abs(cr), float(np.abs(Mnp).max()))
    ok = sorted(p.tolist()) == list(range(len(p))) and c <= cr + 1e-9 * sc
    small += bool(info['stats'][7] & 0x40000000)
    if not ok:
        bad += 1; print("MISMATCH", k, Mnp.shape, c, cr, info['stats'])
print(f"{N} instances, {small} on the one-workgroup path, {bad} mismatches, {time.time()-t0:.1f} s")
# three threads, three streams, concurrently
errs = []
def worker(seed):
    r = np.random.RandomState(seed); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(60):
            n = int(r.randint(2, 257)); Mnp = (r.rand(n, n) * 5).astype(np.float32)
            M = torch.from_numpy(Mnp).to(dev, non_blocking=False)
            p = ot.assign_exact(M).cpu().numpy().astype(np.int64)
            ref = oracle.exact_perm(Mnp)
            if not np.array_equal(p, ref): errs.append((seed, k))
ths = [threading.Thread(target=worker, args=(s,)) for s in (1, 2, 3)]
[t.start() for t in ths]; [t.join() for t in ths]
print("concurrent:", "ok" if not errs else errs[:5])
