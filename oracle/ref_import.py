"""Import the UNMODIFIED reference package from /root/reference with the POT stand-in on the
path (build container only — /root/reference does not exist on the GPU box).  Test
infrastructure: used by tests/golden/make_golden.py and by CPU tests that cross-check the
oracle against the reference's own code when it is present."""
import os
import sys

REFERENCE_ROOT = os.environ.get("CFM_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "torchcfm"))


def import_reference():
    """Returns (conditional_flow_matching, optimal_transport) modules of the reference."""
    if not available():
        raise ImportError("reference tree not present")
    here = os.path.dirname(os.path.abspath(__file__))
    stand = os.path.join(here, "ot_standin")
    if stand not in sys.path:
        sys.path.insert(0, stand)
    import importlib.util
    import types
    # load the two hot-path modules directly: torchcfm/__init__ would also import
    # models/unet and utils (matplotlib, torchdyn) which are out of scope / not installed
    pkg = types.ModuleType("torchcfm_ref")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "torchcfm")]
    sys.modules["torchcfm_ref"] = pkg
    mods = {}
    for name in ("optimal_transport", "conditional_flow_matching"):
        spec = importlib.util.spec_from_file_location(
            f"torchcfm_ref.{name}", os.path.join(REFERENCE_ROOT, "torchcfm", f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"torchcfm_ref.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["conditional_flow_matching"], mods["optimal_transport"]


def import_runner_metrics():
    """The reference's evaluation metrics module (runner/src/models/components/distribution_distances.py
    with its siblings mmd.py / optimal_transport.py), unmodified, POT stand-in on the path."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "runner")):
        raise ImportError("reference runner tree not present")
    here = os.path.dirname(os.path.abspath(__file__))
    stand = os.path.join(here, "ot_standin")
    if stand not in sys.path:
        sys.path.insert(0, stand)
    import importlib.util
    import types
    base = os.path.join(REFERENCE_ROOT, "runner", "src", "models", "components")
    pkg = types.ModuleType("runner_ref_components")
    pkg.__path__ = [base]
    sys.modules["runner_ref_components"] = pkg
    out = {}
    for name in ("mmd", "optimal_transport", "distribution_distances"):
        spec = importlib.util.spec_from_file_location(f"runner_ref_components.{name}", os.path.join(base, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"runner_ref_components.{name}"] = m
        spec.loader.exec_module(m)
        out[name] = m
    return out["distribution_distances"], out["mmd"]


def import_runner_sinkhorn():
    """The reference's own pure-NumPy statement of POT's unbalanced Sinkhorn-Knopp loop
    (runner/src/models/components/sinkhorn_knopp_unbalanced.py), unmodified.  With reg_m_1 = reg_m_2 -> infinity
    its fixed point is the balanced entropic plan, which pins the log-domain solvers at convergence."""
    path = os.path.join(REFERENCE_ROOT, "runner", "src", "models", "components", "sinkhorn_knopp_unbalanced.py")
    if not os.path.isfile(path):
        raise ImportError("reference runner tree not present")
    import importlib.util
    spec = importlib.util.spec_from_file_location("runner_ref_sinkhorn_knopp_unbalanced", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
