"""What happens to the log-domain Sinkhorn loop when the exponent x_ij = g_j - M_ij / reg is formed in fp32 against a
per-row reference instead of in fp64 (the change that would take three fp64 operations per pair out of the
variant-B kernel): x_ij - r_i = fp32( fp32(g_j - r_i) - fp32(M_ij) * fp32(1 / reg) ), r_i = the row's
previous LSE.  Reports, per configuration, the marginal-error floor the fp32 exponents reach, the potentials' distance
from the float64 loop, and the threshold at which the switch to the fp64 phase would have to sit."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import cfm_oracle as oracle

def lse(A, axis):
    m = A.max(axis, keepdims=True)
    return (m + np.log(np.exp(A - m).sum(axis, keepdims=True))).squeeze(axis)

def lse32(pot, ref, M32, inv_reg32, axis):
    """LSE over `axis` of pot - M/reg with fp32 exponents against the reference `ref` of the kept axis."""
    if axis == 1:      # rows keep i: pot = g_j, ref = r_i
        d = (pot[None, :] - ref[:, None]).astype(np.float32)
    else:              # columns keep j: pot = f_i, ref = r_j
        d = (pot[:, None] - ref[None, :]).astype(np.float32)
    x = (d - M32 * inv_reg32).astype(np.float32)                     # fp32 fma
    m = x.max(axis, keepdims=True)
    s = np.exp((x - m).astype(np.float32)).astype(np.float32).sum(axis, dtype=np.float32, keepdims=True)
    return ref + (m.astype(np.float64) + np.log(s.astype(np.float64))).squeeze(axis)

def run(name, reg, iters, B=None):
    x0, x1 = oracle.config_inputs(name, B=B) if B else oracle.config_inputs(name)
    M = oracle.sqeuclid_cost_f64(np.asarray(x0), np.asarray(x1)).astype(np.float32)
    M64 = M.astype(np.float64); n0, n1 = M.shape
    loga, logb = -np.log(n0), -np.log(n1)
    inv32 = np.float32(1.0 / reg)
    u = np.zeros(n0); v = np.zeros(n1); u3 = np.zeros(n0); v3 = np.zeros(n1)
    ru = np.zeros(n0); rv = np.zeros(n1)
    floor = []
    for it in range(iters):
        v = logb - lse(u[:, None] - M64 / reg, 0); u = loga - lse(v[None, :] - M64 / reg, 1)
        lv = lse32(u3, rv, M, inv32, 0); v3n = logb - lv; rv = lv
        lu = lse32(v3n, ru, M, inv32, 1); u3n = loga - lu; ru = lu
        if it % 10 == 0 or it == iters - 1:
            # POT's check: column marginal of the current plan vs b, in the fp32-exponent arithmetic and exactly
            e32 = np.sqrt(((np.exp(v3 - v3n) - 1.0) ** 2).sum()) / n1 if it else np.nan
            P = np.exp(u3n[:, None] + v3n[None, :] - M64 / reg)
            floor.append((it, e32, np.sqrt(((P.sum(0) - 1.0 / n1) ** 2).sum())))
        u3, v3 = u3n, v3n
    sc = max(np.abs(u).max(), np.abs(v).max(), 1.0)
    print(f"{name} B={n0} reg={reg}: |u32-u64| {np.abs(u3 - u).max():.2e} |v32-v64| {np.abs(v3 - v).max():.2e} (scale {sc:.0f}, "
          f"relative {max(np.abs(u3 - u).max(), np.abs(v3 - v).max()) / sc:.1e}); exponent scale max M/reg {M.max() / reg:.0f}")
    print("   iteration: measured err (as the kernel would see it) / true column-marginal error of the fp32-exponent iterate")
    print("   " + "  ".join(f"{it}: {a:.1e}/{b:.1e}" for it, a, b in floor[-6:]))
    print(f"   switch threshold today 1e-4/sqrt(B1) = {1e-4 / np.sqrt(n1):.1e}")

if __name__ == "__main__":
    run("C2", 0.05, 60, B=1024)
    run("C2", 2.0, 220, B=1024)
    run("C5", 0.1, 60, B=1024)
    run("C5", 4.0, 60, B=1024)
