"""cfm_amd — MI355X (gfx950) native minibatch-OT coupling + CFM sampling, behind the
TorchCFM API (``torchcfm.OTPlanSampler`` / ``ConditionalFlowMatcher`` family).

    import cfm_amd as torchcfm
    from cfm_amd.conditional_flow_matching import ExactOptimalTransportConditionalFlowMatcher
    from cfm_amd.optimal_transport import OTPlanSampler, wasserstein
    from cfm_amd.models import MLP
"""
from .conditional_flow_matching import *  # noqa: F401,F403
from .conditional_flow_matching import (  # noqa: F401
    ConditionalFlowMatcher,
    ExactOptimalTransportConditionalFlowMatcher,
    SchrodingerBridgeConditionalFlowMatcher,
    TargetConditionalFlowMatcher,
    VariancePreservingConditionalFlowMatcher,
    pad_t_like_x,
)
from .models import MLP  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .train import RegressionStep  # noqa: F401
from .optimal_transport import OTPlanSampler, wasserstein  # noqa: F401

__version__ = "0.1.0"
