"""muax_amd -- MI355X-native batched MuZero search behind the muax.MuZero.act() contract.

Only the hot path of bwfbowen/muax is rebuilt here: the
mctx.muzero_policy loop as hand-written gfx950 HIP kernels behind a C-ABI
(include/mzsearch.h), and the host-side mirror of the reference interface
(MuZero.act, the repr_fn/pred_fn/dy_fn plugin surface, the policy adapters).
"""
from . import checkpoint, episode_tracer, loss, nn, optimizers, prng, replay_buffer, utils  # noqa: F401
from .episode_tracer import NStep, PNStep  # noqa: F401
from .replay_buffer import Trajectory, TrajectoryReplayBuffer  # noqa: F401
from .loss import Transition, default_loss_fn  # noqa: F401
from .model import MuZero  # noqa: F401
from .nn import MZNetwork, MZNetworkParams, create_muzero_network  # noqa: F401
from .policy import GumbelMuZeroPolicy, MuZeroPolicy, Policy, StochasticMuZeroPolicy  # noqa: F401
from .search import MuZeroSearch, PolicyOutput, SearchConfig, SearchTree, key_words  # noqa: F401
from .sharding import allreduce_mean_flat, gather_roots, shard_roots  # noqa: F401
from .vector import VectorCollector, episode_trajectory, fit_vector, nstep_returns, test_vector  # noqa: F401
from .train import _temperature_fn, collect_batched, fit, fit_batched, rollout, rollout_batched, test  # noqa: F401

__version__ = "0.1.0"
