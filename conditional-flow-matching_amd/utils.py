"""Glue + toy data — counterpart of ``torchcfm/utils.py`` (ref lines cited inline)."""
import math

import numpy as np
import torch


def eight_normal_sample(n, dim, scale=1, var=1):
    """Mixture of eight Gaussians on the unit circle (axis points first, then the diagonals), as
    torchcfm/utils.py:11-32 draws it: one MVN draw of the noise for all n points, then ONE
    multinomial draw of the component labels — the same two RNG calls in the same order, so a
    seeded run yields the reference's points; the per-point Python loop of the reference is a
    single gather here."""
    h = 1.0 / np.sqrt(2)
    axis_pts = [(1, 0), (-1, 0), (0, 1), (0, -1)]
    diag_pts = [(sx * h, sy * h) for sx in (1, -1) for sy in (1, -1)]
    centers = torch.tensor(axis_pts + diag_pts) * scale
    cov = math.sqrt(var) * torch.eye(dim)
    noise = torch.distributions.multivariate_normal.MultivariateNormal(torch.zeros(dim), cov).sample((n,))
    component = torch.multinomial(torch.ones(len(centers)), n, replacement=True)
    return centers.index_select(0, component) + noise


def generate_moons(n_samples=100, noise=1e-4):
    """torchdyn.datasets.generate_moons restated: two half circles plus
    one uniform jitter value per row from the global np.random stream."""
    n_out = n_samples // 2
    n_in = n_samples - n_out
    outer_x = np.cos(np.linspace(0, np.pi, n_out))
    outer_y = np.sin(np.linspace(0, np.pi, n_out))
    inner_x = 1 - np.cos(np.linspace(0, np.pi, n_in))
    inner_y = 1 - np.sin(np.linspace(0, np.pi, n_in)) - 0.5
    X = np.vstack([np.append(outer_x, inner_x), np.append(outer_y, inner_y)]).T
    y = np.hstack([np.zeros(n_out, dtype=np.int64), np.ones(n_in, dtype=np.int64)])
    if noise is not None:
        X += np.random.rand(n_samples, 1) * noise
    return torch.Tensor(X), torch.LongTensor(y)


def sample_moons(n):
    """ref: torchcfm/utils.py:35-37."""
    x0, _ = generate_moons(n, noise=0.2)
    return x0 * 3 - 1


def sample_8gaussians(n):
    """ref: torchcfm/utils.py:40-41."""
    return eight_normal_sample(n, 2, scale=5, var=0.1).float()


class torch_wrapper(torch.nn.Module):
    """Wraps model to torchdyn compatible format (ref: torchcfm/utils.py:44-52)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, t, x, *args, **kwargs):
        from .models import MLP
        if isinstance(self.model, MLP) and self.model.time_varying and not torch.is_grad_enabled() \
                and torch.cuda.is_available():
            return self.model.forward_hip(x, t)   # time column folded into the GEMM epilogue
        return self.model(torch.cat([x, t.repeat(x.shape[0])[:, None]], 1))


def plot_trajectories(traj, n=2000):
    """Scatter plot of a [T, B, >=2] trajectory: start points, the flow, end points
    (counterpart of torchcfm/utils.py:55-65; matplotlib is imported lazily)."""
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(6, 6))
    layers = (
        (traj[0, :n], dict(s=10, alpha=0.8, c="black", label="Prior sample z(S)")),
        (traj[:, :n].reshape(-1, traj.shape[-1]), dict(s=0.2, alpha=0.2, c="olive", label="Flow")),
        (traj[-1, :n], dict(s=4, alpha=1, c="blue", label="z(0)")),
    )
    for pts, style in layers:
        ax.scatter(pts[:, 0], pts[:, 1], **style)
    ax.legend()
    ax.set_xticks([])
    ax.set_yticks([])
    plt.show()
