import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/oracle")
    import torch
    from cfm_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    import cfm_amd.optimal_transport as ot, cfm_oracle as oracle
    _lib.load(); dev = _lib.require_gpu()
    x0, x1 = oracle.config_inputs("C2"); a, b = x0.to(dev), x1.to(dev); M = ot.cost_matrix(a, b)
    ot.sinkhorn_log_points(a, b, M, 0.05, max_iter=20, stop_thr=0.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ot.sinkhorn_log_points(a, b, M, 0.05, max_iter=200, stop_thr=0.0); e1.record(); torch.cuda.synchronize()
    print(os.path.basename(sys.argv[1]), f"{e0.elapsed_time(e1)/200*1e3:.2f} us / iteration", flush=True)
else:
    for f in sorted(os.listdir(os.path.join(ROOT, "tools/probe"))):
        if f.startswith("libcfm_skp_"):
            subprocess.run([sys.executable, __file__, os.path.join(ROOT, "tools/probe", f)])
